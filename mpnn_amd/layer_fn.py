"""One autograd node for a whole MPNN layer (float32, GPU).

Forward and backward of MPNNLayerSum / MPNNLayerCat (reference models/layers.py:25-69 and :91-140) as a fixed
sequence of library calls, so that nothing between them needs an elementwise pass:

  forward    AB   = x Wn^T                                  NT GEMM        (Wn = fused_linear, or [Wi ; Wj] of `linear`)
             aggr = AGG(AB[:, :H], AB[:, H:], e, We)        edge pass      (+ cnt, relu mask bits)
             y    = relu(x Wx^T + aggr Wa^T)                NT GEMM, 2 segments, relu in the epilogue
  backward   ga   = (gy * [y>0]) Wa ;  gAB[:, :H] = ga*cnt  NT GEMM, relu mask on the operand, * cnt in the epilogue
             gAB[:, H:], gWe = edge backward(ga)            edge pass over the transpose
             gx   = (gy * [y>0]) Wx + gAB Wn                NT GEMM, 2 segments (no separate add)
             gWn  = gAB^T x ; gW_out = (gy*[y>0])^T [x | aggr]   TN GEMMs
"""
from __future__ import annotations

import torch

from . import ops
from .dense import gemm_nt, gemm_nt_bf16, gemm_tn, gemm_tn_bf16
from .graph import MPGraph


class _LayerFn(torch.autograd.Function):
    """``halo`` (partitioned graphs, see partition.Halo): the rows of x are this rank's destination range, the
    source-side projection is exchanged across the ranks before the edge pass and its gradient summed back to the
    owners after the backward edge pass; everything else is local."""

    @staticmethod
    def forward(ctx, x, edge_attr, w_nodes, w_edge, w_out, graph: MPGraph, halo=None):
        x = x if x.is_contiguous() else x.contiguous()
        f = x.size(1)
        h = w_out.size(0)
        ab = gemm_nt(x, w_nodes)
        a = ab[:, :h]
        b = ab[:, h:] if halo is None else halo.gather(ab[:, h:])
        ea_s, ea_t = ops._prepare(a, b, edge_attr, w_edge, graph)
        w_edge_c = w_edge.contiguous()
        want = any(ctx.needs_input_grad[i] for i in (0, 2, 3, 4))
        aggr, cnt, maskbits = ops._run_forward(a, b, ea_s, w_edge_c, graph.by_dst, graph, want)
        y = gemm_nt(x, w_out[:, :f], aggr, w_out[:, f:], relu=True)
        if want:
            if maskbits is not None:
                graph.tpos
            # the gathered source table is only read again by the recomputing backward
            b_all = b if (halo is not None and maskbits is None) else None
            ctx.save_for_backward(x, ab, aggr, cnt, maskbits, y, ea_t, w_nodes, w_edge_c, w_out, b_all)
            ctx.graph, ctx.halo = graph, halo
        return y

    @staticmethod
    def backward(ctx, gy):
        x, ab, aggr, cnt, maskbits, y, ea_t, w_nodes, w_edge, w_out, b_all = ctx.saved_tensors
        graph, halo = ctx.graph, ctx.halo
        f = x.size(1)
        h = w_out.size(0)
        gy = gy if gy.is_contiguous() else gy.contiguous()
        wt_out = w_out.t().contiguous()               # [F+H, H]
        g_ab = torch.empty_like(ab)
        g_wout = torch.empty_like(w_out) if ctx.needs_input_grad[4] else None
        g, m = gy, y
        # gz = gy * [y > 0] leaves the first weight-gradient GEMM as a by-product (written in whole 512-byte rows),
        # so the four GEMMs after it read one matrix instead of (gy, y); needs 16-byte aligned rows
        if g_wout is not None and h % 4 == 0 and f % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (gy, y, x, aggr)):
            gz = torch.empty_like(gy)
            gemm_tn(gy, x, mask=y, out=g_wout, g_out=gz, x2=aggr)   # both halves of gW_out in one pass over gy
            g, m = gz, None
        elif g_wout is not None:
            gemm_tn(gy, x, mask=y, out=g_wout, x2=aggr)
        ga = gemm_nt(g, wt_out[f:], mask1=m, emul=cnt, out2=g_ab[:, :h])
        if halo is None:
            g_we = ops._run_backward_src(ab[:, :h], ab[:, h:], ga, ea_t, w_edge, graph.by_src, graph, g_ab[:, h:], maskbits)
        else:   # gradient for every source row this rank's edges read, then summed back to the owners
            gb_all = torch.empty((graph.num_src, h), dtype=ab.dtype, device=ab.device)
            g_we = ops._run_backward_src(ab[:, :h], b_all, ga, ea_t, w_edge, graph.by_src, graph, gb_all, maskbits)
            halo.scatter_sum(gb_all, g_ab[:, h:])
        gx = None
        if ctx.needs_input_grad[0]:
            gx = gemm_nt(g, wt_out[:f], g_ab, w_nodes.t().contiguous(), mask1=m)
        g_wn = gemm_tn(g_ab, x) if ctx.needs_input_grad[2] else None
        return gx, None, g_wn, g_we if ctx.needs_input_grad[3] else None, g_wout, None, None


class _LayerFnBf16(torch.autograd.Function):
    """The same node for bfloat16 models: bf16 storage, float32 accumulation.  The products with a huge row count
    run on mpnn_gemm_nt_bf16, the three weight-gradient products (reductions over the rows) on mpnn_gemm_tn_bf16,
    the first of which also leaves the relu-masked upstream gradient."""

    @staticmethod
    def forward(ctx, x, edge_attr, w_nodes, w_edge, w_out, graph: MPGraph):
        f = x.size(1)
        h = w_out.size(0)
        ab = gemm_nt_bf16(x, w_nodes.contiguous())
        a, b = ab[:, :h], ab[:, h:]
        ea_s, ea_t = ops._prepare(a, b, edge_attr, w_edge, graph)
        w_edge_c = w_edge.contiguous()
        want = any(ctx.needs_input_grad[i] for i in (0, 2, 3, 4))
        aggr, cnt, maskbits = ops._run_forward(a, b, ea_s, w_edge_c, graph.by_dst, graph, want)
        wt = w_out.t().contiguous()                   # [F+H, H]; its transpose halves are needed both ways
        y = gemm_nt_bf16(x, w_out[:, :f].contiguous(), aggr, w_out[:, f:].contiguous(), relu=True)
        if want:
            if maskbits is not None:
                graph.tpos
            ctx.save_for_backward(x, ab, aggr, cnt, maskbits, y, ea_t, w_nodes, w_edge_c, wt)
            ctx.graph = graph
        return y

    @staticmethod
    def backward(ctx, gy):
        x, ab, aggr, cnt, maskbits, y, ea_t, w_nodes, w_edge, wt = ctx.saved_tensors
        graph = ctx.graph
        f = x.size(1)
        h = wt.size(1)
        gy = gy if gy.is_contiguous() else gy.contiguous()
        g_ab = torch.empty_like(ab)
        g_wout = torch.empty((h, f + h), dtype=gy.dtype, device=gy.device) if ctx.needs_input_grad[4] else None
        # gz = gy * [y > 0] is a by-product of the first weight-gradient product (whole 256-byte rows)
        if g_wout is not None and gy.data_ptr() % 16 == 0:
            gz = torch.empty_like(gy)
            gemm_tn_bf16(gy, x, mask=y, out=g_wout, g_out=gz, x2=aggr)   # both halves of gW_out in one pass over gy
        else:
            gz = torch.ops.aten.threshold_backward(gy, y, 0)
            if g_wout is not None:
                torch.mm(gz.t(), torch.cat([x, aggr], 1), out=g_wout)
        ga = gemm_nt_bf16(gz, wt[f:], emul=cnt, out2=g_ab[:, :h])
        g_we = ops._run_backward_src(ab[:, :h], ab[:, h:], ga, ea_t, w_edge, graph.by_src, graph, g_ab[:, h:], maskbits)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = gemm_nt_bf16(gz, wt[:f], g_ab, w_nodes.t().contiguous())
        g_wn = gemm_tn_bf16(g_ab, x) if ctx.needs_input_grad[2] else None
        return gx, None, g_wn, g_we if ctx.needs_input_grad[3] else None, g_wout, None


def supported(x, edge_attr, *weights) -> bool:
    if not x.is_cuda:
        return False
    ts = (x, edge_attr) + weights
    if all(t.dtype == torch.float32 for t in ts):
        return True
    # bfloat16: every width a multiple of 8 (16-byte rows); weights are checked by their shapes
    if all(t.dtype == torch.bfloat16 for t in ts):
        h = weights[-1].size(0)
        return x.dim() == 2 and x.size(1) % 8 == 0 and h % 8 == 0 and x.is_contiguous() and x.data_ptr() % 16 == 0
    return False


@torch.compiler.disable
def layer_forward(x, edge_attr, w_nodes, w_edge, w_out, graph: MPGraph, halo=None) -> torch.Tensor:
    if x.dtype == torch.bfloat16:
        if halo is not None:
            raise NotImplementedError("the bfloat16 layer node has no halo exchange")
        return _LayerFnBf16.apply(x, edge_attr, w_nodes, w_edge, w_out, graph)
    return _LayerFn.apply(x, edge_attr, w_nodes, w_edge, w_out, graph, halo)
