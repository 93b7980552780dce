"""Autograd operators over the HIP library: fused message+aggregate and segment mean.

``aggregate`` replaces the gather -> message -> scatter-add that PyG's
``MessagePassing.propagate`` runs for the reference (models/layers.py:38,110) and the
backward autograd derives for it; ``segment_mean`` replaces ``global_mean_pool``
(models/mpnn.py:54,150).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from .graph import CSR, MPGraph, build_csr


# Optional timing hook: when set to a list, every library call of the two edge passes appends
# (name, start_event, end_event) recorded on the stream the kernels run on.
EVENT_SINK = None


class _Timed:
    def __init__(self, name, device):
        self.name, self.device = name, device

    def __enter__(self):
        if EVENT_SINK is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)
            self.t0.record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, *exc):
        if EVENT_SINK is not None:
            self.t1.record(torch.cuda.current_stream(self.device))
            EVENT_SINK.append((self.name, self.t0, self.t1))
        return False


def _check_features(name, t, rows, H=None):
    _lib.require_device(t, name)
    if t.dim() != 2 or t.size(0) != rows or (H is not None and t.size(1) != H):
        raise ValueError(f"{name} must have shape [{rows}, {H if H is not None else 'H'}], got {tuple(t.shape)}")
    if t.stride(1) != 1:
        raise ValueError(f"{name} must have unit stride along features, got strides {t.stride()}")


# The backward reads the relu mask bits the forward stored (one row gather per edge) when the layout exists
# for H; MPNN_NO_MASKBITS=1 forces the recomputing backward (two row gathers per edge, no [E, words] buffer).
USE_MASKBITS = os.environ.get("MPNN_NO_MASKBITS", "0") != "1"


def _run_forward(A, B, ea_sorted, We, csr: CSR, graph: MPGraph, want_cnt: bool):
    """mpnn_aggregate_fwd.  Returns (out, counts, maskbits); ``counts`` = (byte counts [N, H], exact float counts of
    the long rows [n_split + n_big, H] or None) when a backward will follow, else None."""
    lib = _lib.load()
    N, H = A.shape
    dev, dtype = A.device, A.dtype
    out = torch.empty((N, H), dtype=dtype, device=dev)
    cnt = torch.empty((N, H), dtype=torch.uint8, device=dev) if want_cnt else None
    n_long = csr.n_split + csr.n_big
    cnt_big = torch.empty((n_long, H), dtype=torch.float32, device=dev) if (want_cnt and n_long) else None
    words = lib.mpnn_mask_words(H, _lib.dtype_code(dtype)) if (want_cnt and USE_MASKBITS and csr.num_edges) else 0
    vec = 8 if dtype == torch.bfloat16 else 4
    if words and (A.data_ptr() % 16 or B.data_ptr() % 16 or A.stride(0) % vec or B.stride(0) % vec):
        words = 0  # the library would take its scalar path, which writes no mask bits
    maskbits = torch.empty((csr.num_edges, words), dtype=torch.int32, device=dev) if words else None
    part_out = part_cnt = None
    if csr.n_parts > 0:
        part_out = graph.scratch("part_out", (csr.n_parts, H), torch.float32, dev)
        if want_cnt:
            part_cnt = graph.scratch("part_cnt", (csr.n_parts, H), torch.float32, dev)
    with _Timed("aggregate_fwd", dev):
        _lib.check(lib.mpnn_aggregate_fwd(
            _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(ea_sorted), ea_sorted.size(1), _lib.ptr(We),
            _lib.ptr(csr.col), _lib.ptr(csr.items), csr.n_items, _lib.ptr(csr.split), csr.n_split, csr.n_parts,
            _lib.ptr(out), out.stride(0), _lib.ptr(cnt), _lib.ptr(cnt_big), csr.n_big, _lib.ptr(part_out),
            _lib.ptr(part_cnt), _lib.ptr(maskbits), N, H, _lib.dtype_code(dtype), _lib.stream_handle(dev)),
            "mpnn_aggregate_fwd")
    return out, ((cnt, cnt_big) if want_cnt else None), maskbits


def count_fixup(g, counts, csr: CSR, g_a) -> None:
    """g_a[row] = g[row] * (exact count) for the rows whose counts can exceed a byte (after g_a = g * byte counts)."""
    cnt_big = counts[1]
    if cnt_big is None:
        return
    lib = _lib.load()
    _lib.check(lib.mpnn_count_fixup(_lib.ptr(g), g.stride(0), _lib.ptr(cnt_big), _lib.ptr(csr.items), _lib.ptr(csr.split),
                                    csr.n_split, csr.n_parts, csr.n_big, _lib.ptr(g_a), g_a.stride(0), g.size(1),
                                    _lib.dtype_code(g.dtype), _lib.stream_handle(g.device)), "mpnn_count_fixup")


def times_counts(g, counts, csr: CSR, out) -> None:
    """out = g * counts (elementwise; ``out`` a [N, H] view with unit column stride)."""
    torch.mul(g, counts[0], out=out)
    count_fixup(g, counts, csr, out)


def _run_backward_src(A, B, G, ea_sorted_t, We, csr_t: CSR, graph: MPGraph, gB, maskbits=None):
    """Fills gB (a [Ns, H] view, unit feature stride) and returns gWe."""
    lib = _lib.load()
    H = A.size(1)
    Fe = ea_sorted_t.size(1)
    dev, dtype = A.device, A.dtype
    gWe = torch.empty((H, Fe), dtype=dtype, device=dev)
    part = graph.scratch("part_gB", (csr_t.n_parts, H), torch.float32, dev) if csr_t.n_parts > 0 else None
    scratch = graph.scratch("gwe", (lib.mpnn_bwd_scratch_floats(H, Fe),), torch.float32, dev)
    if maskbits is not None:
        with _Timed("aggregate_bwd_src", dev):
            _lib.check(lib.mpnn_aggregate_bwd_src_masked(
                _lib.ptr(G), G.stride(0), _lib.ptr(maskbits), _lib.ptr(graph.tpos), _lib.ptr(ea_sorted_t), Fe,
                _lib.ptr(csr_t.col), _lib.ptr(csr_t.items), csr_t.n_items, _lib.ptr(csr_t.split), csr_t.n_split,
                csr_t.n_parts, _lib.ptr(gB), gB.stride(0), _lib.ptr(gWe), _lib.ptr(part), _lib.ptr(scratch),
                csr_t.num_rows, H, _lib.dtype_code(dtype), _lib.stream_handle(dev)),
                "mpnn_aggregate_bwd_src_masked")
        return gWe
    with _Timed("aggregate_bwd_src", dev):
        _lib.check(lib.mpnn_aggregate_bwd_src(
            _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(G), G.stride(0), _lib.ptr(ea_sorted_t), Fe,
            _lib.ptr(We), _lib.ptr(csr_t.col), _lib.ptr(csr_t.items), csr_t.n_items, _lib.ptr(csr_t.split),
            csr_t.n_split, csr_t.n_parts, _lib.ptr(gB), gB.stride(0), _lib.ptr(gWe), _lib.ptr(part),
            _lib.ptr(scratch), csr_t.num_rows, H, _lib.dtype_code(dtype), _lib.stream_handle(dev)),
            "mpnn_aggregate_bwd_src")
    return gWe


def _prepare(A, B, edge_attr, We, graph: MPGraph):
    H = A.size(1)
    _check_features("A", A, graph.num_dst)
    _check_features("B", B, graph.num_src, H)
    if We.dim() != 2 or We.size(0) != H or We.size(1) != edge_attr.size(1):
        raise ValueError(f"We must have shape [{H}, {edge_attr.size(1)}], got {tuple(We.shape)}")
    if B.dtype != A.dtype or We.dtype != A.dtype:
        raise TypeError(f"A, B, We must share a dtype, got {A.dtype}, {B.dtype}, {We.dtype}")
    if edge_attr.requires_grad:
        raise NotImplementedError("gradient with respect to edge_attr is not implemented by the HIP engine")
    return graph.sorted_edge_attr(edge_attr, A.dtype)


class _AggregateFused(torch.autograd.Function):
    """out = aggregate(AB[:, :H], AB[:, H:], ...) for a fused [N, 2H] projection (models/layers.py:104-105)."""

    @staticmethod
    def forward(ctx, AB, edge_attr, We, graph):
        H = AB.size(1) // 2
        A, B = AB[:, :H], AB[:, H:]
        ea_s, ea_t = _prepare(A, B, edge_attr, We, graph)
        We_c = We.contiguous()
        want = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
        out, counts, maskbits = _run_forward(A, B, ea_s, We_c, graph.by_dst, graph, want)
        if want:
            if maskbits is not None:
                graph.tpos  # built once per graph, outside the backward
            ctx.save_for_backward(AB, We_c, counts[0], counts[1], ea_t, maskbits)
            ctx.graph = graph
        return out

    @staticmethod
    def backward(ctx, g):
        AB, We, cnt, cnt_big, ea_t, maskbits = ctx.saved_tensors
        graph = ctx.graph
        H = AB.size(1) // 2
        g = g.contiguous()
        gAB = torch.empty_like(AB)
        times_counts(g, (cnt, cnt_big), graph.by_dst, gAB[:, :H])  # d out / d A: every edge of row n sees A[n]
        gWe = _run_backward_src(AB[:, :H], AB[:, H:], g, ea_t, We, graph.by_src, graph, gAB[:, H:], maskbits)
        return gAB, None, gWe, None


class _Aggregate(torch.autograd.Function):
    """Separate destination table A [Nd, H] and source table B [Ns, H] (partitioned graphs)."""

    @staticmethod
    def forward(ctx, A, B, edge_attr, We, graph):
        ea_s, ea_t = _prepare(A, B, edge_attr, We, graph)
        We_c = We.contiguous()
        want = any(ctx.needs_input_grad[i] for i in (0, 1, 3))
        out, counts, maskbits = _run_forward(A, B, ea_s, We_c, graph.by_dst, graph, want)
        if want:
            if maskbits is not None:
                graph.tpos
            ctx.save_for_backward(A, B, We_c, counts[0], counts[1], ea_t, maskbits)
            ctx.graph = graph
        return out

    @staticmethod
    def backward(ctx, g):
        A, B, We, cnt, cnt_big, ea_t, maskbits = ctx.saved_tensors
        graph = ctx.graph
        g = g.contiguous()
        gA = torch.empty((A.size(0), A.size(1)), dtype=A.dtype, device=A.device)
        times_counts(g, (cnt, cnt_big), graph.by_dst, gA)
        gB = torch.empty((graph.num_src, A.size(1)), dtype=A.dtype, device=A.device)
        gWe = _run_backward_src(A, B, g, ea_t, We, graph.by_src, graph, gB, maskbits)
        return gA, gB, None, gWe, None


@torch.compiler.disable
def aggregate_fused(AB: torch.Tensor, edge_attr: torch.Tensor, We: torch.Tensor, graph: MPGraph) -> torch.Tensor:
    """sum_{e: dst(e)=n} relu((AB[n,:H] + AB[src(e),H:]) + We @ edge_attr[e])  ->  [N, H]."""
    if AB.dim() != 2 or AB.size(1) % 2 != 0:
        raise ValueError(f"AB must have shape [N, 2H], got {tuple(AB.shape)}")
    if not AB.is_contiguous():
        AB = AB.contiguous()
    return _AggregateFused.apply(AB, edge_attr, We, graph)


@torch.compiler.disable
def aggregate(A: torch.Tensor, B: torch.Tensor, edge_attr: torch.Tensor, We: torch.Tensor,
              graph: MPGraph) -> torch.Tensor:
    """sum_{e: dst(e)=n} relu((A[n] + B[src(e)]) + We @ edge_attr[e])  ->  [num_dst, H]."""
    if A.stride(-1) != 1:
        A = A.contiguous()
    if B.stride(-1) != 1:
        B = B.contiguous()
    return _Aggregate.apply(A, B, edge_attr, We, graph)


# ----------------------------------------------------------------------------------------
# EGNN: message pass + coordinate update in one edge pass (no reference counterpart)
# ----------------------------------------------------------------------------------------
def _run_egnn_forward(A, B, ea_s, We, pos, w_pos, graph: MPGraph):
    """mpnn_egnn_fwd: (sum of messages [N, H], updated coordinates [N, 3])."""
    lib = _lib.load()
    N, H = A.shape
    dev, dtype = A.device, A.dtype
    csr = graph.by_dst
    out = torch.empty((N, H), dtype=dtype, device=dev)
    pos_out = torch.empty((N, 3), dtype=torch.float32, device=dev)
    part = graph.scratch("part_out", (csr.n_parts, H), torch.float32, dev) if csr.n_parts else None
    small = graph.scratch("part_small", (csr.n_parts, 4), torch.float32, dev) if csr.n_parts else None
    with _Timed("egnn_fwd", dev):
        _lib.check(lib.mpnn_egnn_fwd(
            _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(ea_s), ea_s.size(1), _lib.ptr(We),
            _lib.ptr(w_pos), _lib.ptr(pos), _lib.ptr(pos), _lib.ptr(csr.col), _lib.ptr(csr.items), csr.n_items,
            _lib.ptr(csr.split), csr.n_split, csr.n_parts, _lib.ptr(out), out.stride(0), _lib.ptr(pos_out),
            _lib.ptr(part), _lib.ptr(small), N, H, _lib.dtype_code(dtype), _lib.stream_handle(dev)),
            "mpnn_egnn_fwd")
    return out, pos_out


def _run_egnn_backward(A, B, g, gp, ea_s, ea_t, We, w_pos, pos, graph: MPGraph, gA, gB):
    """mpnn_egnn_bwd_dst + mpnn_egnn_bwd_src.  Fills gA and gB ([N, H] views with unit column stride) and returns
    (gWe [H, Fe], g_pos [N, 3], g_wpos [H])."""
    lib = _lib.load()
    N, H = A.shape
    Fe = ea_s.size(1)
    dev, dtype = A.device, A.dtype
    code, stream = _lib.dtype_code(dtype), _lib.stream_handle(dev)
    scratch = graph.scratch("gwe", (lib.mpnn_bwd_scratch_floats(H, Fe),), torch.float32, dev)

    csr = graph.by_dst
    S = torch.empty((N,), dtype=torch.float32, device=dev)
    g_wp = torch.empty((H,), dtype=dtype, device=dev)
    part = graph.scratch("part_out", (csr.n_parts, H), torch.float32, dev) if csr.n_parts else None
    small = graph.scratch("part_small", (csr.n_parts, 4), torch.float32, dev) if csr.n_parts else None
    with _Timed("egnn_bwd_dst", dev):
        _lib.check(lib.mpnn_egnn_bwd_dst(
            _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(g), g.stride(0), _lib.ptr(gp),
            _lib.ptr(ea_s), Fe, _lib.ptr(We), _lib.ptr(w_pos), _lib.ptr(pos), _lib.ptr(pos), _lib.ptr(csr.col),
            _lib.ptr(csr.items), csr.n_items, _lib.ptr(csr.split), csr.n_split, csr.n_parts, _lib.ptr(gA),
            gA.stride(0), _lib.ptr(S), _lib.ptr(g_wp), _lib.ptr(part), _lib.ptr(small), _lib.ptr(scratch), N, H,
            code, stream), "mpnn_egnn_bwd_dst")

    csr_t = graph.by_src
    gps = torch.empty((N, 3), dtype=torch.float32, device=dev)
    gWe = torch.empty((H, Fe), dtype=dtype, device=dev)
    part_t = graph.scratch("part_gB", (csr_t.n_parts, H), torch.float32, dev) if csr_t.n_parts else None
    small_t = graph.scratch("part_small_t", (csr_t.n_parts, 4), torch.float32, dev) if csr_t.n_parts else None
    with _Timed("egnn_bwd_src", dev):
        _lib.check(lib.mpnn_egnn_bwd_src(
            _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(g), g.stride(0), _lib.ptr(gp),
            _lib.ptr(ea_t), Fe, _lib.ptr(We), _lib.ptr(w_pos), _lib.ptr(pos), _lib.ptr(pos), _lib.ptr(csr_t.col),
            _lib.ptr(csr_t.items), csr_t.n_items, _lib.ptr(csr_t.split), csr_t.n_split, csr_t.n_parts,
            _lib.ptr(gB), gB.stride(0), _lib.ptr(gps), _lib.ptr(gWe), _lib.ptr(part_t), _lib.ptr(small_t),
            _lib.ptr(scratch), N, H, code, stream), "mpnn_egnn_bwd_src")
    g_pos = torch.addcmul(gps, gp, (1.0 + S).unsqueeze(1))     # gp * (1 + S) + gps
    return gWe, g_pos, g_wp


def _prepare_egnn(A, B, edge_attr, We, pos, w_pos, graph: MPGraph):
    if graph.num_dst != graph.num_src:
        raise ValueError("egnn_aggregate needs one node set (num_dst == num_src)")
    ea_s, ea_t = _prepare(A, B, edge_attr, We, graph)
    N, H = A.shape
    if pos.shape != (N, 3) or pos.dtype != torch.float32:
        raise ValueError(f"pos must be float32 [{N}, 3], got {pos.dtype} {tuple(pos.shape)}")
    if w_pos.shape != (H,) or w_pos.dtype != A.dtype:
        raise ValueError(f"w_pos must have shape [{H}] and dtype {A.dtype}")
    return ea_s, ea_t


class _EgnnAggregate(torch.autograd.Function):
    """(A [N,H], B [N,H], edge_attr, We [H,Fe], pos [N,3] float, w_pos [H]) -> (aggr [N,H], pos' [N,3])."""

    @staticmethod
    def forward(ctx, A, B, edge_attr, We, pos, w_pos, graph):
        ea_s, ea_t = _prepare_egnn(A, B, edge_attr, We, pos, w_pos, graph)
        We_c, pos_c, wp_c = We.contiguous(), pos.contiguous(), w_pos.contiguous()
        out, pos_out = _run_egnn_forward(A, B, ea_s, We_c, pos_c, wp_c, graph)
        ctx.save_for_backward(A, B, We_c, pos_c, wp_c, ea_s, ea_t)
        ctx.graph = graph
        return out, pos_out

    @staticmethod
    def backward(ctx, g, gp):
        A, B, We, pos, wp, ea_s, ea_t = ctx.saved_tensors
        g = torch.zeros_like(A) if g is None else g.contiguous()
        gp = torch.zeros_like(pos) if gp is None else gp.contiguous().float()
        gA, gB = torch.empty_like(A, memory_format=torch.contiguous_format), torch.empty_like(
            B, memory_format=torch.contiguous_format)
        gWe, g_pos, g_wp = _run_egnn_backward(A, B, g, gp, ea_s, ea_t, We, wp, pos, ctx.graph, gA, gB)
        return gA, gB, None, gWe, g_pos, g_wp, None


@torch.compiler.disable
def egnn_aggregate(A, B, edge_attr, We, pos, w_pos, graph: MPGraph):
    """m_e = relu((A[dst]+B[src]) + We@e);  returns (sum_e m_e,  pos + sum_e (pos[dst]-pos[src]) * (w_pos . m_e))."""
    if A.stride(-1) != 1:
        A = A.contiguous()
    if B.stride(-1) != 1:
        B = B.contiguous()
    return _EgnnAggregate.apply(A, B, edge_attr, We, pos, w_pos, graph)


# ----------------------------------------------------------------------------------------
# segment mean (global_mean_pool)
# ----------------------------------------------------------------------------------------
def _averaged(csr: CSR) -> int:
    """Segments that are real graphs; the ones after them are the padding of a fixed-capacity batch (register_pool)."""
    return csr.num_rows if csr.averaged is None else csr.averaged


def segment_mean_forward(x: torch.Tensor, csr: CSR) -> torch.Tensor:
    lib = _lib.load()
    x = x if x.stride(-1) == 1 else x.contiguous()
    H = x.size(1)
    out = torch.empty((csr.num_rows, H), dtype=x.dtype, device=x.device)
    _lib.check(lib.mpnn_segment_mean_fwd(_lib.ptr(x), x.stride(0), _lib.ptr(csr.rowptr), _lib.ptr(csr.col),
                                         _lib.ptr(out), out.stride(0), csr.num_rows, _averaged(csr), H,
                                         _lib.dtype_code(x.dtype), _lib.stream_handle(x.device)),
               "mpnn_segment_mean_fwd")
    return out


def segment_mean_backward(g: torch.Tensor, csr: CSR, n: int) -> torch.Tensor:
    lib = _lib.load()
    g = g.contiguous()
    H = g.size(1)
    gx = torch.empty((n, H), dtype=g.dtype, device=g.device)
    _lib.check(lib.mpnn_segment_mean_bwd(_lib.ptr(g), g.stride(0), _lib.ptr(csr.rowptr), _lib.ptr(csr.col),
                                         _lib.ptr(gx), gx.stride(0), csr.num_rows, _averaged(csr), H,
                                         _lib.dtype_code(g.dtype), _lib.stream_handle(g.device)),
               "mpnn_segment_mean_bwd")
    return gx


class _SegmentMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, csr: CSR):
        ctx.csr = csr
        ctx.n = x.size(0)
        return segment_mean_forward(x, csr)

    @staticmethod
    def backward(ctx, g):
        return segment_mean_backward(g, ctx.csr, ctx.n), None


_POOL_CACHE = {}


def register_pool(batch: torch.Tensor, num_graphs: int, rowptr: torch.Tensor, col: torch.Tensor,
                  averaged: int | None = None) -> None:
    """Seeds the segment structure of ``batch`` (sorted, graph b owns nodes rowptr[b]:rowptr[b+1]; ``col`` is
    the node list in segment order) so that global_mean_pool need not derive it - the batch assembler
    already knows it.  ``averaged`` < num_graphs: the graphs from there on are padding (the spare graph of a
    fixed-capacity batch, which may hold thousands of unused rows): their pooled rows are zeros and their nodes take
    a zero gradient, without a pass over them."""
    csr = CSR()
    csr.num_rows, csr.num_cols, csr.num_edges = int(num_graphs), batch.numel(), batch.numel()
    csr.rowptr, csr.col = rowptr, col
    csr.averaged = int(num_graphs) if averaged is None else int(averaged)
    _POOL_CACHE.clear()
    _POOL_CACHE[(batch.data_ptr(), batch._version, batch.numel(), str(batch.device), None)] = (batch, csr)


def pool_structure(batch: torch.Tensor, size: int | None) -> CSR:
    """Segments of ``batch`` (cached for the last batch vector seen: mini-batch loops bring a new one every step)."""
    if batch.dtype != torch.int64 or batch.dim() != 1:
        raise ValueError("batch must be an int64 vector with one entry per node")
    key = (batch.data_ptr(), batch._version, batch.numel(), str(batch.device), size)
    hit = _POOL_CACHE.get(key)
    if hit is not None:
        return hit[1]
    if size is None:
        size = int(batch.max().item()) + 1 if batch.numel() > 0 else 0  # host read, as in the reference
    nodes = torch.arange(batch.numel(), dtype=torch.int64, device=batch.device)
    csr = build_csr(batch.contiguous(), nodes, size, batch.numel())
    _POOL_CACHE.clear()
    _POOL_CACHE[key] = (batch, csr)
    return csr


@torch.compiler.disable
def global_mean_pool(x: torch.Tensor, batch: torch.Tensor, size: int | None = None) -> torch.Tensor:
    """Per-graph mean of node rows; same contract as PyG's global_mean_pool (models/mpnn.py:54,150)."""
    _lib.require_device(x, "x")
    if batch is None:
        return x.mean(dim=0, keepdim=True)
    if batch.numel() != x.size(0):
        raise ValueError("batch must be an int64 vector with one entry per node")
    return _SegmentMean.apply(x, pool_structure(batch, size))
