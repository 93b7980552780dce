"""One autograd node for a whole MPNN layer (GPU; float32, or bfloat16 storage with float32 accumulation).

Forward and backward of MPNNLayerSum / MPNNLayerCat (reference models/layers.py:25-69 and :91-140) as a fixed
sequence of library calls, so that nothing between them needs an elementwise pass:

  forward    AB   = x Wn^T                                  NT GEMM        (Wn = fused_linear, or [Wi ; Wj] of `linear`)
             aggr = AGG(AB[:, :H], AB[:, H:], e, We)        edge pass      (+ cnt, relu mask bits)
             y    = relu(x Wx^T + aggr Wa^T)                NT GEMM, 2 segments, relu in the epilogue
  backward   ga   = (gy * [y>0]) Wa ;  gAB[:, :H] = ga*cnt  NT GEMM, relu mask on the operand, * cnt in the epilogue
             gAB[:, H:], gWe = edge backward(ga)            edge pass over the transpose
             gx   = (gy * [y>0]) Wx + gAB Wn                NT GEMM, 2 segments (no separate add)
             gWn  = gAB^T x ; gW_out = (gy*[y>0])^T [x | aggr]   TN GEMMs

Wide inputs (F >= 2 H, float32: Flickr's 500 features against 64 channels) read x ONCE per direction instead of
twice - the reference reads it at models/layers.py:104 and again at :140:

  forward    [A | B | P] = x [Wn ; Wx]^T                    one NT GEMM, 3 H columns
             y = relu(P + aggr Wa^T)                        NT GEMM over H columns with P as its addend
  backward   gz = gy * [y>0], gW_out[:, F:] = gz^T aggr     small TN GEMM; gz lands in columns 2H.. of [gA | gB | gz]
             ga, gA, gB, gWe as above
             gx = [gA | gB | gz] [Wn ; Wx]                  one NT GEMM, K = 3 H
             [gWn ; gW_out[:, :F]] = [gA | gB | gz]^T x     one TN GEMM
"""
from __future__ import annotations

import torch

from . import islands, ops
from .dense import gemm_nt, gemm_nt_bf16, gemm_tn, gemm_tn_bf16
from .graph import MPGraph


def fused_x(dtype, f: int, h: int) -> bool:
    """Whether the layer reads x once per direction (projection and the x-part of the update as one product).  It
    costs one more [N, H] matrix written and read, so it pays when x is at least twice as wide as that."""
    return dtype == torch.float32 and f >= 2 * h and f % 4 == 0 and h % 4 == 0


def forward_impl(x, edge_attr, w_nodes, w_edge, w_out, graph: MPGraph, want: bool):
    """The forward sequence of library calls.  Returns (y, ab, aggr, cnt, maskbits, ea); cnt = (byte counts, exact
    counts of the long rows or None) and maskbits are None when ``want`` (a backward will follow) is false, maskbits
    also when no mask layout exists.  ``ea`` is the sorted layout of edge_attr the backward reads: transpose order
    on the general path, destination order when the graph's islands let the layer run as one launch (the pair of
    both orders for the point-cloud form, see islands.sorted_attr)."""
    bf = x.dtype == torch.bfloat16
    x = x if x.is_contiguous() else x.contiguous()
    f = x.size(1)
    h = w_out.size(0)
    plan = islands.plan_for(graph, f, h, edge_attr.size(1), False)
    if plan is not None:
        # one launch for the whole layer; its backward recomputes the intermediates (the point-cloud form keeps its
        # sums over edges), so the slots that carry them
        # on the general path hold unwritten tensors of the same shapes
        y, _, saved = islands.forward(plan, graph, x, edge_attr, w_nodes, w_edge, w_out, keep=want)
        n = x.size(0)
        cnt = (torch.empty((n, h), dtype=torch.uint8, device=x.device), None) if want else None
        width = (3 if fused_x(x.dtype, f, h) else 2) * h      # the shape the general path would return
        # (the point-cloud form keeps its sums for the backward, in float32)
        region = plan.region_nodes > 0
        aggr = saved[5] if region and saved[5] is not None and saved[5].dtype == x.dtype else x.new_empty((n, h))
        return y, x.new_empty((n, width)), aggr, cnt, None, ((saved[1], saved[6]) if region else saved[1])
    fused = fused_x(x.dtype, f, h) and x.data_ptr() % 16 == 0
    if fused:
        ab = gemm_nt(x, torch.cat([w_nodes, w_out[:, :f]], dim=0))     # [A | B | P], P = x Wx^T
    else:
        ab = gemm_nt_bf16(x, w_nodes.contiguous()) if bf else gemm_nt(x, w_nodes)
    a, b = ab[:, :h], ab[:, h:2 * h]
    ea_s, ea_t = ops._prepare(a, b, edge_attr, w_edge, graph)
    aggr, cnt, maskbits = ops._run_forward(a, b, ea_s, w_edge.contiguous(), graph.by_dst, graph, want)
    if fused:
        y = gemm_nt(aggr, w_out[:, f:], add=ab[:, 2 * h:], relu=True)
    elif bf:
        y = gemm_nt_bf16(x, w_out[:, :f].contiguous(), aggr, w_out[:, f:].contiguous(), relu=True)
    else:
        y = gemm_nt(x, w_out[:, :f], aggr, w_out[:, f:], relu=True)
    if want and maskbits is not None:
        graph.tpos                       # built once per graph, outside the backward
    return y, ab, aggr, cnt, maskbits, ea_t


def backward_impl(gy, x, ab, aggr, cnt, maskbits, y, ea_t, w_nodes, w_edge, w_out, graph: MPGraph, needs):
    """The backward sequence.  ``needs`` = (x, w_nodes, w_edge, w_out) gradients wanted; returns them in that order
    (None where not wanted).  ``ea_t`` is the ``ea`` forward_impl returned."""
    need_x, need_wn, need_we, need_wout = needs
    bf = x.dtype == torch.bfloat16
    f = x.size(1)
    h = w_out.size(0)
    plan = islands.plan_for(graph, f, h, w_edge.size(1), False)
    if plan is not None:   # the choice forward_impl made: it depends on the graph and the shapes only
        if plan.region_nodes:      # ea_t: the pair (destination order, source order), see islands.sorted_attr
            kept = aggr if aggr.dtype == torch.float32 else None
            saved = (x, ea_t[0], w_nodes.contiguous(), w_edge.contiguous(), w_out.contiguous(), kept, ea_t[1])
        else:
            saved = (x, ea_t, w_nodes.contiguous(), w_edge.contiguous(), w_out.contiguous(), None, None)
        return islands.backward(plan, graph, gy, y, saved, needs)[:4]
    w_edge = w_edge.contiguous()
    gy = gy if gy.is_contiguous() else gy.contiguous()
    if gy.data_ptr() % 16:
        gy = gy.clone()                               # a view into a larger buffer: fresh allocations are aligned
    wt_out = w_out.t().contiguous()                   # [F+H, H]
    g_ab = torch.empty_like(ab)
    g_wout = torch.empty((h, f + h), dtype=gy.dtype, device=gy.device) if need_wout else None
    if ab.size(1) == 3 * h:    # forward_impl read x once (fused_x): so does the backward
        gz = g_ab[:, 2 * h:]
        if need_wout and all(t.data_ptr() % 16 == 0 for t in (gy, y, aggr)):
            gemm_tn(gy, aggr, mask=y, out=g_wout[:, f:], g_out=gz)
        else:
            gz.copy_(torch.ops.aten.threshold_backward(gy, y, 0))
            if need_wout:
                gemm_tn(gz, aggr, out=g_wout[:, f:])
        ga = gemm_nt(gz, wt_out[f:], emul=cnt[0], out2=g_ab[:, :h])
        ops.count_fixup(ga, cnt, graph.by_dst, g_ab[:, :h])
        g_we = ops._run_backward_src(ab[:, :h], ab[:, h:2 * h], ga, ea_t, w_edge, graph.by_src, graph,
                                     g_ab[:, h:2 * h], maskbits)
        gx = gemm_nt(g_ab[:, :2 * h], w_nodes.t().contiguous(), gz, wt_out[:f]) if need_x else None
        g_wn = None
        if need_wn or need_wout:
            g_cat = gemm_tn(g_ab, x)                  # [gWn ; gW_out[:, :F]]
            g_wn = g_cat[:2 * h] if need_wn else None
            if need_wout:
                g_wout[:, :f] = g_cat[2 * h:]
        return gx, g_wn, g_we if need_we else None, g_wout
    if bf:
        # gz = gy * [y > 0] is a by-product of the first weight-gradient product (whole 256-byte rows)
        if g_wout is not None:
            gz = torch.empty_like(gy)
            gemm_tn_bf16(gy, x, mask=y, out=g_wout, g_out=gz, x2=aggr)   # both halves of gW_out in one pass over gy
        else:
            gz = torch.ops.aten.threshold_backward(gy, y, 0)
        ga = gemm_nt_bf16(gz, wt_out[f:], emul=cnt[0], out2=g_ab[:, :h])
        ops.count_fixup(ga, cnt, graph.by_dst, g_ab[:, :h])
        g_we = ops._run_backward_src(ab[:, :h], ab[:, h:], ga, ea_t, w_edge, graph.by_src, graph, g_ab[:, h:], maskbits)
        gx = gemm_nt_bf16(gz, wt_out[:f], g_ab, w_nodes.t().contiguous()) if need_x else None
        g_wn = gemm_tn_bf16(g_ab, x) if need_wn else None
        return gx, g_wn, g_we if need_we else None, g_wout
    g, m = gy, y
    # gz = gy * [y > 0] leaves the first weight-gradient GEMM as a by-product (written in whole 512-byte rows),
    # so the four GEMMs after it read one matrix instead of (gy, y); needs 16-byte aligned rows
    if g_wout is not None and h % 4 == 0 and f % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (gy, y, x, aggr)):
        gz = torch.empty_like(gy)
        gemm_tn(gy, x, mask=y, out=g_wout, g_out=gz, x2=aggr)   # both halves of gW_out in one pass over gy
        g, m = gz, None
    elif g_wout is not None:
        gemm_tn(gy, x, mask=y, out=g_wout, x2=aggr)
    ga = gemm_nt(g, wt_out[f:], mask1=m, emul=cnt[0], out2=g_ab[:, :h])
    ops.count_fixup(ga, cnt, graph.by_dst, g_ab[:, :h])
    g_we = ops._run_backward_src(ab[:, :h], ab[:, h:], ga, ea_t, w_edge, graph.by_src, graph, g_ab[:, h:], maskbits)
    gx = gemm_nt(g, wt_out[:f], g_ab, w_nodes.t().contiguous(), mask1=m) if need_x else None
    g_wn = gemm_tn(g_ab, x) if need_wn else None
    return gx, g_wn, g_we if need_we else None, g_wout


class _LayerFn(torch.autograd.Function):
    """The layer as one autograd node, float32 or bfloat16 storage (float32 accumulation throughout).  (The layer
    of one rank of a partitioned graph, with its two exchanges, is partition._PartitionedLayerFn.)"""

    @staticmethod
    def forward(ctx, x, edge_attr, w_nodes, w_edge, w_out, graph: MPGraph):
        want = any(ctx.needs_input_grad[i] for i in (0, 2, 3, 4))
        x = x if x.is_contiguous() else x.contiguous()
        ctx.plan = islands.plan_for(graph, x.size(1), w_out.size(0), edge_attr.size(1), False)
        if ctx.plan is not None:   # the whole layer in one launch; `saved` is all its backward needs
            y, _, saved = islands.forward(ctx.plan, graph, x, edge_attr, w_nodes, w_edge, w_out, keep=want)
            if want:
                ctx.save_for_backward(y, *saved)
                ctx.graph = graph
            return y
        y, ab, aggr, cnt, maskbits, ea_t = forward_impl(x, edge_attr, w_nodes, w_edge, w_out, graph, want)
        if want:
            ctx.save_for_backward(x, ab, aggr, cnt[0], cnt[1], maskbits, y, ea_t, w_nodes, w_edge, w_out)
            ctx.graph = graph
        return y

    @staticmethod
    def backward(ctx, gy):
        needs = tuple(ctx.needs_input_grad[i] for i in (0, 2, 3, 4))
        if ctx.plan is not None:
            y, *saved = ctx.saved_tensors
            gx, g_wn, g_we, g_wout = islands.backward(ctx.plan, ctx.graph, gy, y, tuple(saved), needs)[:4]
            return gx, None, g_wn, g_we, g_wout, None
        x, ab, aggr, cnt8, cnt_big, maskbits, y, ea_t, w_nodes, w_edge, w_out = ctx.saved_tensors
        gx, g_wn, g_we, g_wout = backward_impl(gy, x, ab, aggr, (cnt8, cnt_big), maskbits, y, ea_t, w_nodes, w_edge, w_out,
                                               ctx.graph, needs)
        return gx, None, g_wn, g_we, g_wout, None


class _EgnnLayerFn(torch.autograd.Function):
    """EGNN layer (message pass + coordinate update in one edge pass, then the update) as one node, float32 or
    bfloat16 storage, every product on the library's MFMA kernels:

      forward    AB = x Wn^T ; (aggr, pos') = EGNN edge pass(AB, e, We, pos, w_pos) ; y = relu(x Wx^T + aggr Wa^T)
      backward   gz = gy*[y>0], gW_out = gz^T [x | aggr]            TN GEMM (gz as its by-product when rows are aligned)
                 ga = gz Wa                                         NT GEMM
                 gAB, gWe, g_pos, g_wpos = EGNN backward passes     destination view, then source view
                 gx = gz Wx + gAB Wn ; gWn = gAB^T x                NT GEMM (2 segments), TN GEMM
    """

    @staticmethod
    def forward(ctx, x, pos, edge_attr, w_nodes, w_edge, w_out, w_pos, graph: MPGraph):
        bf = x.dtype == torch.bfloat16
        nt = gemm_nt_bf16 if bf else gemm_nt
        x = x if x.is_contiguous() else x.contiguous()
        f, h = x.size(1), w_out.size(0)
        ctx.plan = islands.plan_for(graph, f, h, edge_attr.size(1), True)
        ctx.w_pos_shape = w_pos.shape
        if ctx.plan is not None:   # the whole layer in one launch; `saved` is all its backward needs
            y, pos_out, saved = islands.forward(ctx.plan, graph, x, edge_attr, w_nodes, w_edge, w_out, pos, w_pos.reshape(-1))
            ctx.save_for_backward(y, *saved)
            ctx.graph = graph
            return y, pos_out
        ab = nt(x, w_nodes.contiguous())
        a, b = ab[:, :h], ab[:, h:]
        pos_c, wp, we = pos.contiguous(), w_pos.contiguous(), w_edge.contiguous()
        ea_s, ea_t = ops._prepare_egnn(a, b, edge_attr, we, pos_c, wp, graph)
        aggr, pos_out = ops._run_egnn_forward(a, b, ea_s, we, pos_c, wp, graph)
        wx, wa = w_out[:, :f], w_out[:, f:]
        if bf:
            wx, wa = wx.contiguous(), wa.contiguous()
        y = nt(x, wx, aggr, wa, relu=True)
        ctx.save_for_backward(x, ab, aggr, y, pos_c, ea_s, ea_t, w_nodes, we, w_out, wp)
        ctx.graph = graph
        return y, pos_out

    @staticmethod
    def backward(ctx, gy, gp):
        if ctx.plan is not None:
            y, *saved = ctx.saved_tensors
            needs = tuple(ctx.needs_input_grad[i] for i in (0, 3, 4, 5))
            gx, g_wn, g_we, g_wout, g_pos, g_wp = islands.backward(ctx.plan, ctx.graph, gy, y, tuple(saved), needs, gp,
                                                                   need_pos=ctx.needs_input_grad[1])
            w_pos_shape = ctx.w_pos_shape
            return (gx, g_pos if ctx.needs_input_grad[1] else None, None, g_wn, g_we, g_wout,
                    g_wp.reshape(w_pos_shape) if ctx.needs_input_grad[6] else None, None)
        x, ab, aggr, y, pos, ea_s, ea_t, w_nodes, we, w_out, wp = ctx.saved_tensors
        bf = x.dtype == torch.bfloat16
        nt, tn = (gemm_nt_bf16, gemm_tn_bf16) if bf else (gemm_nt, gemm_tn)
        f, h = x.size(1), w_out.size(0)
        gy = gy if gy.is_contiguous() else gy.contiguous()
        if gy.data_ptr() % 16:
            gy = gy.clone()
        gp = gp.contiguous().float()
        wt_out = w_out.t().contiguous()               # [F+H, H]
        g_ab = torch.empty_like(ab)
        need_wout = ctx.needs_input_grad[5]
        rows_aligned = bf or (f % 4 == 0 and h % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (y, x, aggr)))
        g_wout = None
        if need_wout and rows_aligned:
            gz = torch.empty_like(gy)
            g_wout = torch.empty((h, f + h), dtype=gy.dtype, device=gy.device)
            tn(gy, x, mask=y, out=g_wout, g_out=gz, x2=aggr)
        else:
            gz = torch.ops.aten.threshold_backward(gy, y, 0)      # gy * [y > 0], elementwise
            if need_wout:
                g_wout = tn(gz, x, x2=aggr)
        ga = nt(gz, wt_out[f:])
        g_we, g_pos, g_wp = ops._run_egnn_backward(ab[:, :h], ab[:, h:], ga, gp, ea_s, ea_t, we, wp, pos, ctx.graph,
                                                   g_ab[:, :h], g_ab[:, h:])
        gx = nt(gz, wt_out[:f], g_ab, w_nodes.t().contiguous()) if ctx.needs_input_grad[0] else None
        g_wn = tn(g_ab, x) if ctx.needs_input_grad[3] else None
        return (gx, g_pos if ctx.needs_input_grad[1] else None, None, g_wn,
                g_we if ctx.needs_input_grad[4] else None, g_wout, g_wp if ctx.needs_input_grad[6] else None, None)


def supported(x, edge_attr, *weights) -> bool:
    """float32 on the GPU, or bfloat16 with a hidden width that is a multiple of 8 (16-byte rows; the input width
    may be anything - layer_forward pads it with zero columns)."""
    if not x.is_cuda:
        return False
    ts = (x, edge_attr) + weights
    if all(t.dtype == torch.float32 for t in ts):
        return True
    if all(t.dtype == torch.bfloat16 for t in ts):
        return x.dim() == 2 and weights[-1].size(0) % 8 == 0
    return False


def _pad_width(x, w_nodes, w_out):
    """Zero columns up to the next multiple of 8 in x and in the weight columns that meet it (differentiable
    copies: the gradients come back in the unpadded shapes)."""
    f = x.size(1)
    pad = -f % 8
    pad_cols = torch.nn.functional.pad
    return (pad_cols(x, (0, pad)), pad_cols(w_nodes, (0, pad)),
            torch.cat([pad_cols(w_out[:, :f], (0, pad)), w_out[:, f:]], dim=1))


def _pad_features(x, w_nodes, w_out):
    """bfloat16 rows must be 16 bytes wide: an input width that is not a multiple of 8 (QM9: 11 atom features,
    point clouds: 3 coordinates) is padded with zero columns, in x and in the weight columns that meet it.  The
    pads are differentiable copies, so the gradients come back in the unpadded shapes."""
    if x.size(1) % 8:
        return _pad_width(x, w_nodes, w_out)
    if x.data_ptr() % 16 or not x.is_contiguous():
        x = x.clone(memory_format=torch.contiguous_format)
    return x, w_nodes, w_out


def _check_dtypes(x, edge_attr, *weights):
    if not supported(x, edge_attr, *weights):
        raise TypeError(
            "the layer runs in float32, or in bfloat16 with a hidden width that is a multiple of 8, on the GPU; got "
            f"x {x.dtype} on {x.device}, weights {[tuple(w.shape) for w in weights]} - there is no fallback path")


def layer_forward_eager(x, edge_attr, w_nodes, w_edge, w_out, graph: MPGraph) -> torch.Tensor:
    _check_dtypes(x, edge_attr, w_nodes, w_edge, w_out)
    if x.dtype == torch.bfloat16 and islands.plan_for(graph, x.size(1), w_out.size(0), edge_attr.size(1), False) is None:
        x, w_nodes, w_out = _pad_features(x, w_nodes, w_out)     # (the one-launch kernels take any input width)
    return _LayerFn.apply(x, edge_attr, w_nodes, w_edge, w_out, graph)


def egnn_layer_forward_eager(x, pos, edge_attr, w_nodes, w_edge, w_out, w_pos, graph: MPGraph):
    """(y [N, H], pos' [N, 3]) of one EGNN layer; pos and its gradient are float32 whatever the feature dtype."""
    _check_dtypes(x, edge_attr, w_nodes, w_edge, w_out, w_pos)
    if x.dtype == torch.bfloat16 and islands.plan_for(graph, x.size(1), w_out.size(0), edge_attr.size(1), True) is None:
        x, w_nodes, w_out = _pad_features(x, w_nodes, w_out)
    return _EgnnLayerFn.apply(x, pos.float(), edge_attr, w_nodes, w_edge, w_out, w_pos, graph)


# The same two functions for callers that a tracing compiler may be looking at: it must not trace into the autograd
# nodes.  Callers that have just seen torch.compiler.is_compiling() return False call the *_eager forms and save the
# marker's few microseconds per call (a tenth of the host time of a layer on a batch of molecules).
layer_forward = torch.compiler.disable(layer_forward_eager)
egnn_layer_forward = torch.compiler.disable(egnn_layer_forward_eager)
