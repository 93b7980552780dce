"""One-launch layers for batches of small graphs and of point clouds (the library's island and region kernels).

A mini-batch of molecules (data/dataset.py:94-96) is a disjoint union of graphs of a few dozen nodes, and a layer
(models/layers.py:91-140) never moves data between them.  When the graph structure knows its islands (graph.Islands)
and they are small enough, the layer runs as mpnn_island_layer_fwd / mpnn_island_layer_bwd: one wavefront per
island, every intermediate in LDS, one launch forward and two backward, instead of the general sequence of GEMM and
edge-pass launches whose working sets would be a few kilobytes each.

Islands of up to 2048 nodes over rows of x at most 4 wide (the k-NN clouds of experiments/modelnet_exp.py:58 with
``data.pos`` as x) take the REGION form of the same idea, mpnn_region_layer_fwd / _bwd: the x of a whole island sits
in LDS and the projection of an edge's two ends is recomputed from it instead of fetched (MPNN layers, and the EGNN
layer for islands of up to 1024 nodes).
"""
from __future__ import annotations

import ctypes
import os
import struct

import torch

from . import _lib
from .graph import MPGraph
from .ops import _Timed

ROWS = 32              # height of a tile (one matrix-core tile): the largest island the kernels take
# Beyond this many islands the general path wins (measured on QM9-shaped batches: 1024 molecules 0.123 vs 0.160 ms
# per layer step, 2048 molecules 0.229 vs 0.213 ms): its GEMMs use the bfloat16 matrix cores at 2.7 times the rate.
COUNT_MAX = int(os.environ.get("MPNN_AMD_ISLANDS_MAX", "1536"))
REGIONS_ENABLED = os.environ.get("MPNN_AMD_REGIONS", "1") != "0"
REGION_ROWS = 128      # rows of one block of the region kernels
REGION_MAX_NODES = 2048


class Plan:
    __slots__ = ("rows", "tail_units", "egnn", "region_nodes", "partial_key", "partial_floats")

    def __init__(self, rows, tail_units, egnn, region_nodes=0):
        self.rows, self.tail_units, self.egnn = rows, tail_units, egnn
        self.region_nodes = region_nodes      # > 0: the region form, with this bound on the largest island
        self.partial_key, self.partial_floats = None, 0


def plan_for(graph: MPGraph, f: int, h: int, fe: int, egnn: bool) -> Plan | None:
    """The tile shape for this layer on this graph, or None when the island kernels do not apply (no islands known,
    islands too large or too many, widths outside the kernels' range)."""
    isl = getattr(graph, "islands", None)
    if isl is None:
        return None
    key = (f, h, fe, egnn, COUNT_MAX, REGIONS_ENABLED)
    plans = graph._scratch.setdefault("island_plans", {})    # per graph: the answer depends on the shapes only
    if key not in plans:
        plans[key] = _plan(isl, f, h, fe, egnn, graph.num_dst)
    return plans[key]


def _plan(isl, f: int, h: int, fe: int, egnn: bool, num_nodes: int) -> Plan | None:
    if isl.count == 0:
        return None
    if isl.max_nodes > ROWS:
        return _region_plan(isl, f, h, fe, egnn, num_nodes)
    if isl.count > COUNT_MAX:
        return None
    if h not in (64, 128) or not 1 <= fe <= 8 or f < 1:
        return None
    if _lib.load().mpnn_island_layer_waves(f, h, fe, int(egnn), 1) < 1:
        return None
    return Plan(ROWS, (isl.tail_rows + ROWS - 1) // ROWS, egnn)


def _region_plan(isl, f: int, h: int, fe: int, egnn: bool, num_nodes: int) -> Plan | None:
    if not REGIONS_ENABLED or isl.tail_rows or isl.max_nodes > REGION_MAX_NODES:
        return None
    if not _lib.load().mpnn_region_layer_fits(f, h, fe, isl.max_nodes, int(egnn)):
        return None
    # every island takes ceil(largest / 128) blocks: not worth it when most of them would find no rows
    groups = (isl.max_nodes + REGION_ROWS - 1) // REGION_ROWS
    if isl.count * groups * REGION_ROWS > 8 * max(num_nodes, 1):
        return None
    return Plan(ROWS, 0, egnn, isl.max_nodes)


# struct mpnn_island_layer, field for field (the layout _lib.IslandLayerDesc declares): one packing call fills it
_DESC = struct.Struct("@PPPiiiq" "Pq" "PPPP" "PP" "iiii" "PqP" "PqP" "PqP" "PPPP" "PP")
assert _DESC.size == ctypes.sizeof(_lib.IslandLayerDesc)


def _desc(plan: Plan, graph: MPGraph, x, ea_s, w_nodes, w_edge, w_out, y, pos=None, w_pos=None, pos_out=None,
          gy=None, gpos_out=None, gx=None, gpos=None, g_wn=None, g_we=None, g_wout=None, g_wpos=None,
          partial=None, gx_part=None) -> _lib.IslandLayerDesc:
    isl, csr = graph.islands, graph.dst_rows
    d = _lib.IslandLayerDesc()
    P = lambda t: 0 if t is None else t.data_ptr()
    _DESC.pack_into(
        d, 0, csr.rowptr.data_ptr(), csr.col.data_ptr(), isl.ptr.data_ptr(),
        isl.count, plan.tail_units, plan.rows, x.size(0),
        x.data_ptr(), x.stride(0),
        ea_s.data_ptr(), w_nodes.data_ptr(), w_edge.data_ptr(), w_out.data_ptr(),
        P(pos), P(w_pos),
        x.size(1), w_out.size(0), ea_s.size(1), _lib.dtype_code(x.dtype),
        y.data_ptr(), y.stride(0), P(pos_out),
        P(gy), 0 if gy is None else gy.stride(0), P(gpos_out),
        P(gx), 0 if gx is None else gx.stride(0), P(gpos),
        P(g_wn), P(g_we), P(g_wout), P(g_wpos),
        P(partial), P(gx_part))
    return d


def _partial_floats(plan: Plan, graph: MPGraph, f: int, h: int, fe: int) -> int:
    """Floats of the per-block partial weight gradients of the plan's backward (asked once per plan and shape)."""
    key = (graph.islands.count, f, h, fe)
    if plan.partial_key != key:
        lib = _lib.load()
        if plan.region_nodes:
            floats = lib.mpnn_region_layer_partial_floats(key[0], plan.region_nodes, f, h, fe, int(plan.egnn))
        else:
            floats = lib.mpnn_island_layer_partial_floats(key[0], plan.tail_units, f, h, fe, int(plan.egnn))
        if floats < 0:
            raise _lib.MpnnError("the layer's shape was rejected when asking for the size of its partial sums")
        plan.partial_key, plan.partial_floats = key, floats
    return plan.partial_floats


def _inputs(x, edge_attr, w_nodes, w_edge, w_out, graph: MPGraph):
    n, f = x.shape
    h = w_out.size(0)
    if n != graph.num_dst:
        raise ValueError(f"x has {n} rows, the graph {graph.num_dst} nodes")
    if w_nodes.shape != (2 * h, f) or w_out.shape != (h, f + h) or w_edge.shape != (h, edge_attr.size(1)):
        raise ValueError(f"weights {tuple(w_nodes.shape)}, {tuple(w_edge.shape)}, {tuple(w_out.shape)} do not fit "
                         f"x [{n}, {f}] and edge_attr [{edge_attr.size(0)}, {edge_attr.size(1)}]")
    if edge_attr.requires_grad:
        raise NotImplementedError("gradient with respect to edge_attr is not implemented by the HIP engine")
    x = x if x.is_contiguous() else x.contiguous()
    ea_s = graph.sorted_edge_attr(edge_attr, x.dtype, "dst")[0]
    return x, ea_s, w_nodes.contiguous(), w_edge.contiguous(), w_out.contiguous()


def sorted_attr(plan: Plan, graph: MPGraph, edge_attr, dtype):
    """edge_attr in the layout(s) the plan's kernels read: destination order for the island form (what forward()
    hands on as ``saved[1]``); for the region form the pair (destination order, source order) = ``saved[1]``,
    ``saved[6]``.  (The graph's own sorted copies, never copies of them: a batch refilled in place rewrites them.)"""
    if not plan.region_nodes:
        return graph.sorted_edge_attr(edge_attr, dtype, "dst")[0]
    return graph.sorted_edge_attr(edge_attr, dtype)


def _region_forward(plan: Plan, graph: MPGraph, x, edge_attr, w_nodes, w_edge, w_out, keep: bool, pos=None, w_pos=None):
    """``saved`` = (x, edge_attr in destination order, w_nodes, w_edge, w_out, the kept sums or None, edge_attr in
    source order[, pos, w_pos])."""
    n = x.size(0)
    x, _, w_nodes, w_edge, w_out = _inputs(x, edge_attr, w_nodes, w_edge, w_out, graph)
    ea_s, ea_t = sorted_attr(plan, graph, edge_attr, x.dtype)
    lib = _lib.load()
    y = torch.empty((n, w_out.size(0)), dtype=x.dtype, device=x.device)
    # the sums over incoming edges, kept for the backward (which would otherwise walk the edges once more)
    aggr = torch.empty((n, w_out.size(0)), dtype=torch.float32, device=x.device) if keep else None
    pos_out = None
    if plan.egnn:
        pos, w_pos = pos.contiguous(), w_pos.contiguous()
        pos_out = torch.empty_like(pos)
    d = _desc(plan, graph, x, ea_s, w_nodes, w_edge, w_out, y, pos, w_pos, pos_out)
    with _Timed("region_layer_fwd", x.device):
        _lib.check(lib.mpnn_region_layer_fwd(ctypes.byref(d), plan.region_nodes, _lib.ptr(aggr),
                                             _lib.stream_handle(x.device)), "mpnn_region_layer_fwd")
    saved = (x, ea_s, w_nodes, w_edge, w_out, aggr, ea_t)
    return y, pos_out, (saved + (pos, w_pos) if plan.egnn else saved)


def _region_backward(plan: Plan, graph: MPGraph, gy, y, saved, needs, gpos_out=None, need_pos=True):
    x, ea_s, w_nodes, w_edge, w_out, aggr, ea_t = saved[:7]
    pos, w_pos = saved[7:9] if plan.egnn else (None, None)
    if aggr is not None and (aggr.dtype != torch.float32 or not aggr.is_contiguous()):
        aggr = None                                   # (recomputed)
    need_x, need_wn, need_we, need_wout = needs
    lib = _lib.load()
    dev, dt = x.device, x.dtype
    n, f = x.shape
    h = w_out.size(0)
    gy = gy if gy.is_contiguous() else gy.contiguous()
    gx = torch.empty((n, f), dtype=dt, device=dev) if need_x else None
    g_wn = torch.empty_like(w_nodes) if need_wn else None
    g_we = torch.empty_like(w_edge) if need_we else None
    g_wout = torch.empty_like(w_out) if need_wout else None
    gpos = g_wpos = None
    if plan.egnn:
        gpos_out = gpos_out.contiguous().float()
        gpos = torch.empty_like(pos) if need_pos else None     # (not wanted: a first layer's backward is one pass shorter)
        g_wpos = torch.empty_like(w_pos)
    floats = _partial_floats(plan, graph, f, h, ea_s.size(1))
    partial = graph.scratch("region_partial", (max(floats, 1),), torch.float32, dev)
    gx_part = graph.scratch("region_gx", (n, f), torch.float32, dev) if need_x else None
    ga = graph.scratch("region_ga", (n * h + n,), torch.float32, dev)     # (+ one float per node with coordinates)
    d = _desc(plan, graph, x, ea_s, w_nodes, w_edge, w_out, y, pos, w_pos, None, gy, gpos_out, gx, gpos, g_wn, g_we,
              g_wout, g_wpos, partial, gx_part)
    src = graph.by_src
    with _Timed("region_layer_bwd", dev):
        _lib.check(lib.mpnn_region_layer_bwd(ctypes.byref(d), plan.region_nodes, _lib.ptr(src.rowptr), _lib.ptr(src.col),
                                             _lib.ptr(ea_t), _lib.ptr(aggr), _lib.ptr(ga), _lib.stream_handle(dev)),
                   "mpnn_region_layer_bwd")
    return gx, g_wn, g_we, g_wout, gpos, g_wpos


def forward(plan: Plan, graph: MPGraph, x, edge_attr, w_nodes, w_edge, w_out, pos=None, w_pos=None, keep=True):
    """y (and pos' for the EGNN form) of one layer.  Returns (y, pos_out | None, saved) where ``saved`` are the
    contiguous tensors the backward needs (``keep`` false: no backward will follow)."""
    if plan.region_nodes:
        return _region_forward(plan, graph, x, edge_attr, w_nodes, w_edge, w_out, keep, pos, w_pos)
    x, ea_s, w_nodes, w_edge, w_out = _inputs(x, edge_attr, w_nodes, w_edge, w_out, graph)
    lib = _lib.load()
    y = torch.empty((x.size(0), w_out.size(0)), dtype=x.dtype, device=x.device)
    pos_out = None
    if plan.egnn:
        pos = pos.contiguous()
        w_pos = w_pos.contiguous()
        pos_out = torch.empty_like(pos)
    d = _desc(plan, graph, x, ea_s, w_nodes, w_edge, w_out, y, pos, w_pos, pos_out)
    with _Timed("island_layer_fwd", x.device):
        _lib.check(lib.mpnn_island_layer_fwd(ctypes.byref(d), _lib.stream_handle(x.device)), "mpnn_island_layer_fwd")
    return y, pos_out, (x, ea_s, w_nodes, w_edge, w_out, pos, w_pos)


def backward(plan: Plan, graph: MPGraph, gy, y, saved, needs, gpos_out=None, need_pos=True):
    """Gradients of one layer.  ``needs`` = (x, w_nodes, w_edge, w_out) wanted; returns (gx, g_wn, g_we, g_wout,
    gpos, g_wpos), None where not wanted (gpos / g_wpos only for the EGNN form)."""
    if plan.region_nodes:
        return _region_backward(plan, graph, gy, y, saved, needs, gpos_out, need_pos)
    x, ea_s, w_nodes, w_edge, w_out, pos, w_pos = saved
    need_x, need_wn, need_we, need_wout = needs
    lib = _lib.load()
    dev, dt = x.device, x.dtype
    n, f = x.shape
    h = w_out.size(0)
    gy = gy if gy.is_contiguous() else gy.contiguous()
    gx = torch.empty((n, f), dtype=dt, device=dev) if need_x else None
    g_wn = torch.empty_like(w_nodes) if need_wn else None
    g_we = torch.empty_like(w_edge) if need_we else None
    g_wout = torch.empty_like(w_out) if need_wout else None
    gpos = g_wpos = None
    if plan.egnn:
        gpos_out = gpos_out.contiguous().float()
        gpos = torch.empty_like(pos)
        g_wpos = torch.empty_like(w_pos)
    floats = _partial_floats(plan, graph, f, h, ea_s.size(1))
    partial = graph.scratch("island_partial", (max(floats, 1),), torch.float32, dev)
    gx_part = graph.scratch("island_gx", (n, f), torch.float32, dev) if need_x else None
    d = _desc(plan, graph, x, ea_s, w_nodes, w_edge, w_out, y, pos, w_pos, None, gy, gpos_out, gx, gpos, g_wn, g_we,
              g_wout, g_wpos, partial, gx_part)
    with _Timed("island_layer_bwd", dev):
        _lib.check(lib.mpnn_island_layer_bwd(ctypes.byref(d), _lib.stream_handle(dev)), "mpnn_island_layer_bwd")
    return gx, g_wn, g_we, g_wout, gpos, g_wpos
