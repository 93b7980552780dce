"""Multi-GPU execution: one process per GPU, torch.distributed (RCCL on ROCm) between them.

Two schemes, both absent from the reference (which is single-process, single-device):

* Large single graph (Flickr-scale and up): nodes are cut into contiguous DESTINATION ranges balanced by in-edge
  count.  A rank owns the rows x[lo:hi], every edge whose target lies in its range, and the outputs of its range.
  Per layer there is one exchange in each direction (class Halo):
    forward  - the source-side projection B of the rows a rank's edges read travels to that rank;
    backward - the gradient with respect to those rows travels back and is summed at the owner.
  Two forms of the exchange:
    "all_gather"  every rank receives every row (padded all-gather, reduce-scatter back): one large collective,
                  right when a rank's edges reference most rows anyway (power-law graph: 96 % at 8 ranks);
    "rows"        only the rows a rank references travel (all-to-all with per-peer row lists built once),
                  the contributions are added at the owner in rank order (deterministic).
  Both start the collective asynchronously and run independent GEMMs of the layer meanwhile.
  A FIRST layer whose input features are replicated on every rank needs no exchange at all: every rank projects
  all N source rows itself (x_all below).
  Weight gradients are summed with one flat all-reduce.

* Batched small graphs (QM9, ModelNet): plain data parallelism; the only exchange is the flat gradient all-reduce
  (averaged).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import dense, ops
from . import _lib
from .dense import gemm_nt, gemm_nt_bf16, gemm_tn, gemm_tn_bf16
from .graph import make_graph
from .layers import MPNNLayerCat, MPNNLayerSum


def dst_range_bounds(dst: torch.Tensor, num_nodes: int, world: int) -> list:
    """bounds[r] = smallest node id n with (number of edges whose target is < n) >= ceil(r*E/world).

    Pure integer arithmetic; every rank computes the same list from the same edge list.
    """
    deg = torch.bincount(dst, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dst.device)
    rowptr[1:] = torch.cumsum(deg, 0)
    e = dst.numel()
    targets = torch.tensor([-(-(r * e) // world) for r in range(1, world)], dtype=torch.int64, device=dst.device)
    inner = torch.searchsorted(rowptr, targets, right=False).clamp_(max=num_nodes).tolist() if world > 1 else []
    bounds = [0]
    for b in inner:
        bounds.append(max(int(b), bounds[-1]))
    bounds.append(num_nodes)
    return bounds


class PartitionedGraph:
    """The piece of a destination-range partition owned by ``rank``: the edges into [lo, hi).

    The local destination rows are numbered 0 .. n_local-1.  The SOURCE rows can be addressed in three layouts,
    each with its own sorted structure (built on first use by ``graph_in``):

      "padded"  the layout a padded all-gather produces: rank q's rows at q * n_pad .. (world * n_pad rows);
      "rows"    only the rows this rank's edges reference, ascending global id (hence grouped by owner):
                position = rank of the id among the referenced ids (``ref_ids``);
      "global"  global node ids (N rows): for a source table every rank computes in full.
    """

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, rank: int, world: int, graph_factory=make_graph):
        src, dst = edge_index[0], edge_index[1]
        self.rank, self.world, self.num_nodes = rank, world, num_nodes
        self.bounds = dst_range_bounds(dst, num_nodes, world)
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        self.n_local = self.hi - self.lo
        self.n_pad = max(self.bounds[q + 1] - self.bounds[q] for q in range(world))
        mine = (dst >= self.lo) & (dst < self.hi)
        self.edge_ids = torch.nonzero(mine).squeeze(1)  # original ids of the local edges, ascending
        self.src_global = src[self.edge_ids]
        self.dst_local = dst[self.edge_ids] - self.lo
        self._factory = graph_factory
        self._graphs = {}
        self._ref = None

    def _referenced(self):
        """(sorted global ids of the source rows this rank's edges read, position of every edge's source in it)."""
        if self._ref is None:
            self._ref = torch.unique(self.src_global, return_inverse=True)
        return self._ref

    @property
    def ref_ids(self) -> torch.Tensor:
        return self._referenced()[0]

    def graph_in(self, layout: str):
        g = self._graphs.get(layout)
        if g is not None:
            return g
        if layout == "padded":
            b = torch.tensor(self.bounds, dtype=torch.int64, device=self.src_global.device)
            owner = (torch.searchsorted(b, self.src_global, right=True) - 1).clamp_(0, self.world - 1)
            src, num_src = owner * self.n_pad + (self.src_global - b[owner]), self.world * self.n_pad
        elif layout == "rows":
            ids, pos = self._referenced()
            src, num_src = pos, ids.numel()
        elif layout == "global":
            src, num_src = self.src_global, self.num_nodes
        else:
            raise ValueError(f"unknown source layout {layout!r}")
        g = self._factory(torch.stack([src, self.dst_local]), self.n_local, num_src)
        self._graphs[layout] = g
        return g

    @property
    def graph(self):
        return self.graph_in("padded")

    @property
    def local_edge_index(self) -> torch.Tensor:
        """[2, E_local]: sources in the padded layout, targets numbered inside the local range."""
        b = torch.tensor(self.bounds, dtype=torch.int64, device=self.src_global.device)
        owner = (torch.searchsorted(b, self.src_global, right=True) - 1).clamp_(0, self.world - 1)
        return torch.stack([owner * self.n_pad + (self.src_global - b[owner]), self.dst_local])

    def referenced_fraction(self) -> float:
        """Share of all N source rows that this rank's edges read."""
        return self.ref_ids.numel() / max(self.num_nodes, 1)

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.lo:self.hi]

    def local_edges(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.edge_ids]


# ------------------------------------------------------------------------------------------------------------------
# collectives
# ------------------------------------------------------------------------------------------------------------------
def _via_host(t: torch.Tensor, group) -> bool:
    """gloo moves host memory: device tensors are staged through the CPU (rehearsals of the multi-rank path on
    one GPU, or hosts without RCCL); with the nccl backend (RCCL) tensors go device to device."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Pending:
    """A collective in flight.  wait() makes the current stream wait for it (the host does not block with RCCL) and
    returns what ``finish`` builds from the received buffer."""

    def __init__(self, work, finish, keep=()):
        self.work, self.finish, self.keep = work, finish, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
        return self.finish()


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    if _via_host(t, group):
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_to_all(recv: torch.Tensor, send: torch.Tensor, recv_counts, send_counts, group):
    """Rows of ``send`` (split by send_counts, one part per peer) to the peers, rows of ``recv`` (split by
    recv_counts) from them.  Returns the Work handle (None when the call completed synchronously)."""
    if _via_host(send, group):
        host = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host, send.cpu(), list(recv_counts), list(send_counts), group=group)
        recv.copy_(host)
        return None
    return dist.all_to_all_single(recv, send, list(recv_counts), list(send_counts), group=group, async_op=True)


class Halo:
    """The two exchanges of a partitioned layer.

    start_gather(b_local)  -> pending; wait() gives the source table in the layout of ``self.graph``
    start_scatter(g_table) -> pending; wait() gives this rank's rows [n_local, H] of the sum over all ranks

    ``mode``: "all_gather", "rows", or "auto" (rows when the ranks reference less than 3/4 of the rows on average).
    ``wire_dtype`` (e.g. torch.bfloat16): rows are converted for the forward transfer (and, in "rows" mode, for the
    backward transfer, whose sum is taken in the table's own dtype after receipt); None keeps the table's dtype and
    with it the single-GPU results.
    Building a Halo is a collective call in "rows" / "auto" mode (the row lists are exchanged once).
    """

    def __init__(self, pg: PartitionedGraph, group=None, mode: str = "all_gather", wire_dtype=None):
        self.pg, self.group, self.wire_dtype = pg, group, wire_dtype
        self.n_local, self.n_pad, self.world, self.rank = pg.n_local, pg.n_pad, pg.world, pg.rank
        if mode == "auto":
            frac = torch.tensor([pg.referenced_fraction()], dtype=torch.float64)
            if pg.src_global.is_cuda and dist.get_backend(group) != "gloo":
                frac = frac.to(pg.src_global.device)
            dist.all_reduce(frac, op=dist.ReduceOp.SUM, group=group)       # every rank takes the same decision
            mode = "rows" if float(frac.item()) / pg.world < 0.75 else "all_gather"
        if mode not in ("all_gather", "rows"):
            raise ValueError(f"unknown halo mode {mode!r}")
        self.mode = mode
        self.layout = "padded" if mode == "all_gather" else "rows"
        self.graph = pg.graph_in(self.layout)
        if mode == "rows":
            self._exchange_row_lists()

    # -------------------------------------------------------------------------------------------- row lists
    def _exchange_row_lists(self) -> None:
        pg = self.pg
        ids = pg.ref_ids                                                   # sorted global ids this rank reads
        dev = ids.device
        b = torch.tensor(pg.bounds, dtype=torch.int64, device=dev)
        cut = torch.searchsorted(ids, b)                                   # ids of owner q: cut[q] .. cut[q+1]
        self.need_counts = (cut[1:] - cut[:-1]).tolist()
        owner = torch.repeat_interleave(torch.arange(self.world, device=dev), cut[1:] - cut[:-1])
        self.need_local = (ids - b[owner]).contiguous()                    # row numbers inside the owner's range
        counts = torch.tensor(self.need_counts, dtype=torch.int64, device=dev)
        got = torch.empty(self.world, dtype=torch.int64, device=dev)
        w = _all_to_all(got, counts, [1] * self.world, [1] * self.world, self.group)
        if w is not None:
            w.wait()
        self.serve_counts = got.tolist()
        self.serve_idx = torch.empty(sum(self.serve_counts), dtype=torch.int64, device=dev)
        w = _all_to_all(self.serve_idx, self.need_local, self.serve_counts, self.need_counts, self.group)
        if w is not None:
            w.wait()
        if self.serve_idx.numel() and (int(self.serve_idx.min()) < 0 or int(self.serve_idx.max()) >= max(self.n_local, 1)):
            raise RuntimeError("halo row lists name rows outside this rank's range: the ranks hold different partitions")
        off = [0]
        for c in self.serve_counts:
            off.append(off[-1] + c)
        self.serve_off = off
        # for the library's pack / unpack kernels (GPU): the served rows as int32, and for every local row the
        # positions of its contributions in the received buffer, ordered by sender rank (a stable sort keeps the
        # buffer order, which is rank order): each row's sum then has one fixed order without atomics
        self.serve_idx32 = self.serve_idx.to(torch.int32)
        order = torch.sort(self.serve_idx, stable=True).indices
        self.unpack_pos = order.to(torch.int32)
        count = torch.bincount(self.serve_idx, minlength=self.n_local)
        self.unpack_rowptr = torch.zeros(self.n_local + 1, dtype=torch.int32, device=dev)
        self.unpack_rowptr[1:] = torch.cumsum(count, 0).to(torch.int32)

    # -------------------------------------------------------------------------------------------- volumes
    def wire_bytes(self, h: int, elem_bytes: int) -> dict:
        """Bytes this rank receives in the forward exchange and sends in the backward one, per layer (own rows,
        which never leave the GPU, excluded)."""
        fwd_elem = torch.empty((), dtype=self.wire_dtype).element_size() if self.wire_dtype is not None else elem_bytes
        if self.mode == "all_gather":
            rows = (self.world - 1) * self.n_pad
            return {"mode": self.mode, "fwd_bytes_received": rows * h * fwd_elem, "bwd_bytes_sent": rows * h * elem_bytes}
        rows = sum(self.need_counts) - self.need_counts[self.rank]
        return {"mode": self.mode, "fwd_bytes_received": rows * h * fwd_elem, "bwd_bytes_sent": rows * h * fwd_elem}

    # -------------------------------------------------------------------------------------------- forward
    def start_gather(self, b_local: torch.Tensor) -> _Pending:
        h, dtype = b_local.size(1), b_local.dtype
        wire = self.wire_dtype or dtype
        if self.mode == "all_gather":
            send = b_local.new_empty((self.n_pad, h), dtype=wire)
            send[:self.n_local] = b_local
            if self.n_local != self.n_pad:
                send[self.n_local:].zero_()
            out = b_local.new_empty((self.world * self.n_pad, h), dtype=wire)
            if _via_host(send, self.group):
                host = torch.empty(out.shape, dtype=wire)
                dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
                out.copy_(host)
                work = None
            else:
                work = dist.all_gather_into_tensor(out, send, group=self.group, async_op=True)
            return _Pending(work, lambda: out if wire == dtype else out.to(dtype), keep=(send,))
        send = self._pack(b_local, wire)
        recv = b_local.new_empty((sum(self.need_counts), h), dtype=wire)
        work = _all_to_all(recv, send, self.need_counts, self.serve_counts, self.group)
        return _Pending(work, lambda: recv if wire == dtype else recv.to(dtype), keep=(send,))

    def gather(self, b_local: torch.Tensor) -> torch.Tensor:
        return self.start_gather(b_local).wait()

    # -------------------------------------------------------------------------------------------- backward
    def start_scatter(self, g_table: torch.Tensor) -> _Pending:
        h, dtype = g_table.size(1), g_table.dtype
        if self.mode == "all_gather":
            g_table = g_table.contiguous()
            if dist.get_backend(self.group) == "gloo":       # gloo has no reduce-scatter: all-reduce, keep the own slice
                _all_reduce_sum(g_table, self.group)
                mine = g_table[self.rank * self.n_pad:(self.rank + 1) * self.n_pad]
                return _Pending(None, lambda: mine[:self.n_local])
            mine = g_table.new_empty((self.n_pad, h))
            work = dist.reduce_scatter_tensor(mine, g_table, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return _Pending(work, lambda: mine[:self.n_local], keep=(g_table,))
        wire = self.wire_dtype or dtype
        send = g_table.contiguous().to(wire)
        recv = g_table.new_empty((sum(self.serve_counts), h), dtype=wire)
        work = _all_to_all(recv, send, self.serve_counts, self.need_counts, self.group)

        def finish():
            if recv.is_cuda:   # mpnn_halo_unpack: per row, the contributions in sender-rank order, float32 sums
                out = g_table.new_empty((self.n_local, h))
                lib = _lib.load()
                _lib.check(lib.mpnn_halo_unpack(_lib.ptr(recv), _lib.dtype_code(wire), _lib.ptr(self.unpack_rowptr),
                                                _lib.ptr(self.unpack_pos), _lib.ptr(out), h, _lib.dtype_code(dtype),
                                                self.n_local, h, _lib.stream_handle(recv.device)), "mpnn_halo_unpack")
                return out
            # host tensors (the index arithmetic rehearsed over gloo on the CPU): every peer's list names each row
            # at most once, so one index_add_ never adds twice into a row, and the peers are added in rank order
            out = g_table.new_zeros((self.n_local, h))
            for q in range(self.world):
                lo, hi = self.serve_off[q], self.serve_off[q + 1]
                if hi > lo:
                    out.index_add_(0, self.serve_idx[lo:hi], recv[lo:hi].to(dtype))
            return out
        return _Pending(work, finish, keep=(send,))

    def _pack(self, b_local: torch.Tensor, wire) -> torch.Tensor:
        """The rows this rank serves, in the order of the peers' lists, in the wire format."""
        if not b_local.is_cuda or b_local.stride(1) != 1:
            return b_local.index_select(0, self.serve_idx).to(wire)
        h = b_local.size(1)
        send = b_local.new_empty((self.serve_idx32.numel(), h), dtype=wire)
        lib = _lib.load()
        _lib.check(lib.mpnn_halo_pack(_lib.ptr(b_local), b_local.stride(0), _lib.dtype_code(b_local.dtype),
                                      _lib.ptr(self.serve_idx32), _lib.ptr(send), _lib.dtype_code(wire),
                                      self.serve_idx32.numel(), h, _lib.stream_handle(b_local.device)), "mpnn_halo_pack")
        return send

    def scatter_sum(self, g_table: torch.Tensor) -> torch.Tensor:
        return self.start_scatter(g_table).wait()


class _HaloExchange(torch.autograd.Function):
    """[n_local, H] -> source table of ``halo.graph``; the backward returns each owner the sum of what all ranks
    computed for its rows.  (The composed path; the fused layer node below calls the Halo directly.)"""

    @staticmethod
    def forward(ctx, b_local, halo):
        ctx.halo = halo
        return halo.gather(b_local)

    @staticmethod
    def backward(ctx, g):
        return ctx.halo.scatter_sum(g), None


def halo_all_gather(b_local: torch.Tensor, pg: PartitionedGraph, group=None) -> torch.Tensor:
    if pg.world == 1:
        return b_local
    return _HaloExchange.apply(b_local, Halo(pg, group))


# ------------------------------------------------------------------------------------------------------------------
# the layer of one rank as a single autograd node (GPU)
# ------------------------------------------------------------------------------------------------------------------
def _kit(dtype):
    return (gemm_nt_bf16, gemm_tn_bf16) if dtype == torch.bfloat16 else (gemm_nt, gemm_tn)


def _aligned(*ts) -> bool:
    return all(t.data_ptr() % 16 == 0 for t in ts)


class _PartitionedLayerFn(torch.autograd.Function):
    """forward   B_loc = x Wb^T -> exchange starts | A = x Wa^T meanwhile -> edge pass -> y = relu(x Wx^T + aggr Wa'^T)
    backward  ga = (gy*[y>0]) Wa', gA = ga*cnt -> edge backward -> exchange of gB starts | gW_out and the target half
              of gWn meanwhile -> gx = gz Wx + [gA | gB] Wn, source half of gWn.

    ``x_all`` (first layer, input replicated on every rank, no input gradient): B for ALL rows is computed locally
    over the "global" source layout and nothing is exchanged; the weight all-reduce that follows every step sums the
    ranks' partial source-half gradients."""

    @staticmethod
    def forward(ctx, x, x_all, edge_attr, w_nodes, w_edge, w_out, graph, halo):
        nt, _ = _kit(x.dtype)
        bf = x.dtype == torch.bfloat16
        x = x if x.is_contiguous() else x.contiguous()
        f, h = x.size(1), w_out.size(0)
        wn = w_nodes.contiguous()
        if x_all is not None:
            x_all = x_all if x_all.is_contiguous() else x_all.contiguous()
            b_all = nt(x_all, wn[h:])
            a = nt(x, wn[:h])
        else:
            pending = halo.start_gather(nt(x, wn[h:]))     # source half first: it travels while the target half is computed
            a = nt(x, wn[:h])
            b_all = pending.wait()
        ea_s, ea_t = ops._prepare(a, b_all, edge_attr, w_edge, graph)
        w_edge_c = w_edge.contiguous()
        want = any(ctx.needs_input_grad[i] for i in (0, 3, 4, 5))
        aggr, cnt, maskbits = ops._run_forward(a, b_all, ea_s, w_edge_c, graph.by_dst, graph, want)
        wx, wa = w_out[:, :f], w_out[:, f:]
        if bf:
            wx, wa = wx.contiguous(), wa.contiguous()
        y = nt(x, wx, aggr, wa, relu=True)
        if want:
            if maskbits is not None:
                graph.tpos
            # the source table is read again only by the backward that recomputes the relu mask
            ctx.save_for_backward(x, x_all, a, b_all if maskbits is None else None, aggr, cnt[0], cnt[1], maskbits, y,
                                  ea_t, w_nodes, w_edge_c, w_out)
            ctx.graph, ctx.halo = graph, halo
        return y

    @staticmethod
    def backward(ctx, gy):
        x, x_all, a, b_all, aggr, cnt8, cnt_big, maskbits, y, ea_t, w_nodes, w_edge, w_out = ctx.saved_tensors
        cnt = (cnt8, cnt_big)
        graph, halo = ctx.graph, ctx.halo
        nt, tn = _kit(x.dtype)
        bf = x.dtype == torch.bfloat16
        f, h = x.size(1), w_out.size(0)
        need_x, need_wn, need_we, need_wout = (ctx.needs_input_grad[i] for i in (0, 3, 4, 5))
        gy = gy if gy.is_contiguous() else gy.contiguous()
        if gy.data_ptr() % 16:
            gy = gy.clone()
        wt_out = w_out.t().contiguous()                   # [F+H, H]
        g_ab = torch.empty((x.size(0), 2 * h), dtype=x.dtype, device=x.device)
        g_wout = torch.empty((h, f + h), dtype=x.dtype, device=x.device) if need_wout else None
        # float32 rows that are 16-byte aligned: the relu mask of gy rides on the operand load of the first GEMM, so
        # the weight-gradient pass that produces gz = gy*[y>0] can wait until the exchange is in flight
        late = (not bf) and need_wout and f % 4 == 0 and h % 4 == 0 and _aligned(gy, y, x, aggr)
        if late:
            ga = gemm_nt(gy, wt_out[f:], mask1=y, emul=cnt8, out2=g_ab[:, :h])
            gz = None
        else:
            if bf and need_wout:
                gz = torch.empty_like(gy)
                tn(gy, x, mask=y, out=g_wout, g_out=gz, x2=aggr)
            else:
                gz = torch.ops.aten.threshold_backward(gy, y, 0)
                if need_wout:
                    tn(gz, x, out=g_wout, x2=aggr)
            ga = nt(gz, wt_out[f:], emul=cnt8, out2=g_ab[:, :h])
        ops.count_fixup(ga, cnt, graph.by_dst, g_ab[:, :h])
        g_table = torch.empty((graph.num_src, h), dtype=x.dtype, device=x.device)
        g_we = ops._run_backward_src(a, b_all, ga, ea_t, w_edge, graph.by_src, graph, g_table, maskbits)
        pending = halo.start_scatter(g_table) if x_all is None else None
        if late:
            gz = torch.empty_like(gy)
            gemm_tn(gy, x, mask=y, out=g_wout, g_out=gz, x2=aggr)
        g_wn = torch.empty((2 * h, f), dtype=x.dtype, device=x.device) if need_wn else None
        if need_wn:
            tn(g_ab[:, :h], x, out=g_wn[:h])
        gx = None
        if x_all is None:
            g_ab[:, h:] = pending.wait()
            if need_wn:
                tn(g_ab[:, h:], x, out=g_wn[h:])
            if need_x:
                gx = nt(gz, wt_out[:f], g_ab, w_nodes.t().contiguous())
        elif need_wn:
            tn(g_table, x_all, out=g_wn[h:])              # this rank's share; the weight all-reduce adds the others'
        return gx, None, None, g_wn, g_we if need_we else None, g_wout, None, None


def _layer_weights(layer):
    if isinstance(layer, MPNNLayerSum):
        w_nodes, w_edge = layer.fused_linear.weight, layer.linear_3.weight
    elif isinstance(layer, MPNNLayerCat):
        f = layer.in_channels
        w = layer.linear.weight
        w_nodes, w_edge = torch.cat([w[:, :f], w[:, f:2 * f]], dim=0), w[:, 2 * f:]
    else:
        raise TypeError(f"unsupported layer type {type(layer).__name__}")
    return w_nodes, w_edge, layer.linear_out.weight


def partitioned_layer(layer, x_local: torch.Tensor, pg: PartitionedGraph, edge_attr_local: torch.Tensor,
                      group=None, aggregate_fn=None, dense_fns=None, halo: Halo | None = None,
                      x_all: torch.Tensor | None = None) -> torch.Tensor:
    """One MPNN layer on the local destination range (same arithmetic as layer.forward on the whole graph).

    ``halo``: the exchange to use (default: a padded all-gather); build it once per PartitionedGraph and pass it to
    every layer.  ``x_all``: the layer's input for ALL nodes, replicated on every rank (a first layer reading the
    dataset's features): no exchange, and no gradient with respect to the input.
    ``aggregate_fn`` / ``dense_fns`` = (project, update) replace the HIP operators by stand-ins with the same
    signatures; they exist for the CPU rehearsals of the exchange logic (the kernels need a GPU)."""
    w_nodes, w_edge, w_out = _layer_weights(layer)
    h = w_out.size(0)
    if x_all is not None and x_local.requires_grad:
        raise ValueError("x_all names a replicated layer input, which cannot carry a gradient")
    if pg.world > 1 and x_all is None and halo is None:
        halo = Halo(pg, group)
    graph = pg.graph_in("global") if x_all is not None else (halo.graph if halo is not None else pg.graph)
    if aggregate_fn is None and dense_fns is None and pg.world > 1:
        from . import layer_fn
        layer_fn._check_dtypes(x_local, edge_attr_local, w_nodes, w_edge, w_out)
        if x_local.dtype == torch.bfloat16 and x_local.size(1) % 8:
            raise ValueError("bfloat16 partitioned layers need an input width that is a multiple of 8")
        return _PartitionedLayerFn.apply(x_local, x_all, edge_attr_local, w_nodes, w_edge, w_out, graph, halo)
    # composed path: separate operators with the exchange as an autograd pair between them
    aggregate_fn = aggregate_fn or ops.aggregate
    project, update = dense_fns or (dense.project, dense.update)
    if x_all is not None:
        a = project(x_local, w_nodes[:h])
        b_all = project(x_all, w_nodes[h:])
    else:
        ab = project(x_local, w_nodes)
        a = ab[:, :h]
        b_all = ab[:, h:] if pg.world == 1 else _HaloExchange.apply(ab[:, h:], halo)
    aggr = aggregate_fn(a, b_all, edge_attr_local, w_edge, graph)
    return update(x_local, aggr, w_out)


def partitioned_model(model, x_local, pg: PartitionedGraph, edge_attr_local, group=None, aggregate_fn=None,
                      dense_fns=None, halo: Halo | None = None, x_all: torch.Tensor | None = None):
    """MPNN.forward (reference models/mpnn.py:86-101) on the local range; returns the local rows of the logits.
    ``x_all``: the replicated node features, for the exchange-free first layer."""
    if pg.world > 1 and halo is None and len(model.layers) > (1 if x_all is not None else 0):
        halo = Halo(pg, group)
    x = x_local
    for k, layer in enumerate(model.layers):
        # every layer ends in a relu: the F.relu(x) of the model loop is the identity
        x = partitioned_layer(layer, x, pg, edge_attr_local, group, aggregate_fn, dense_fns, halo,
                              x_all if k == 0 else None)
    return model.output_layer(x)


def allreduce_gradients(target, group=None, average: bool = False) -> None:
    """One all-reduce over every parameter gradient (sum, or mean for data parallelism).

    ``target``: a module, or a FlatAdam optimizer - its gradients already live in one flat buffer, which is reduced
    in place (no gather into a temporary, no copies back)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    flat = getattr(target, "flat_grad", None)
    if flat is not None:
        target.gather_stray_gradients()
        _all_reduce_sum(flat, group)
        if average:
            flat /= world
        return
    params = [p for p in target.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1).float() for p in params])
    _all_reduce_sum(flat, group)
    if average:
        flat /= world
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
