"""MPNNLayerCat / MPNNLayerSum with the reference's constructor, forward signature and
parameter names (reference models/layers.py:11-69 and :72-140), running on the HIP engine.

Both layers reduce to one fused edge operator after a node-level projection:

  sum mode   relu((xs[dst] + xt[src]) + W3 e)              models/layers.py:104-110,128
  cat mode   relu(W [x[dst] | x[src] | e]) with W = [Wi | Wj | We] column blocks
             = relu((Wi x)[dst] + (Wj x)[src] + We e)      models/layers.py:22,56

so the per-edge GEMM of cat mode becomes a per-node GEMM and neither mode materialises an
[E, *] tensor.  Parameters are created in the reference's order with torch.nn.Linear's
default initialisation, so a seed reproduces the reference's initial weights, and
state_dict keys match (linear / linear_out; fused_linear / linear_3 / linear_out).
"""
from __future__ import annotations

import torch
from torch.nn import Linear

from . import compiled_ops, layer_fn
from .graph import graph_for, graph_for_eager


class _SplitCatWeight(torch.autograd.Function):
    """linear.weight = [Wi | Wj | We] (models/layers.py:22) -> ([Wi ; Wj] stacked row-wise, We), two copies forward and
    two backward.  The same thing written with torch.cat and slices costs a dozen small launches per step, backward
    included, which is a fifth of a replayed step on a batch of molecules."""

    @staticmethod
    def forward(ctx, w, f):
        h = w.size(0)
        ctx.f = f
        w_nodes = w[:, :2 * f].reshape(h, 2, f).transpose(0, 1).reshape(2 * h, f)   # one strided copy
        return w_nodes, w[:, 2 * f:].contiguous()

    @staticmethod
    def backward(ctx, g_nodes, g_edge):
        f = ctx.f
        h = g_nodes.size(0) // 2
        g = torch.empty((h, 2 * f + g_edge.size(1)), dtype=g_nodes.dtype, device=g_nodes.device)
        g[:, :2 * f].view(h, 2, f).copy_(g_nodes.view(2, h, f).transpose(0, 1))
        g[:, 2 * f:].copy_(g_edge)
        return g, None


class MPNNLayerCat(torch.nn.Module):
    def __init__(self, in_channels, edge_channels, out_channels):
        super().__init__()
        self.in_channels, self.edge_channels, self.out_channels = in_channels, edge_channels, out_channels
        self.linear = Linear((in_channels * 2) + edge_channels, out_channels, bias=False)
        self.linear_out = Linear(out_channels + in_channels, out_channels, bias=False)

    def forward(self, x, edge_index, edge_attr):
        # [Wi ; Wj] stacked row-wise gives the [N, 2H] table (target half, source half) in one GEMM.
        w_nodes, w_edge = _SplitCatWeight.apply(self.linear.weight, self.in_channels)
        if torch.compiler.is_compiling():      # torch.compile(model), experiments/flickr_exp.py:30: one opaque operator
            return compiled_ops.layer(x, edge_index, edge_attr, w_nodes, w_edge, self.linear_out.weight)
        graph = graph_for_eager(edge_index, x.size(0))
        return layer_fn.layer_forward_eager(x, edge_attr, w_nodes, w_edge, self.linear_out.weight, graph)


class MPNNLayerSum(torch.nn.Module):
    def __init__(self, in_channels, edge_channels, out_channels):
        super().__init__()
        self.in_channels, self.edge_channels, self.out_channels = in_channels, edge_channels, out_channels
        self.fused_linear = Linear(in_channels, out_channels * 2, bias=False)
        self.linear_3 = Linear(edge_channels, out_channels, bias=False)
        self.linear_out = Linear(out_channels + in_channels, out_channels, bias=False)

    def forward(self, x, edge_index, edge_attr):
        # xs = AB[:, :H] (target half), xt = AB[:, H:] (source half) of the fused projection
        if torch.compiler.is_compiling():      # torch.compile(model), experiments/flickr_exp.py:30: one opaque operator
            return compiled_ops.layer(x, edge_index, edge_attr, self.fused_linear.weight, self.linear_3.weight,
                                      self.linear_out.weight)
        graph = graph_for_eager(edge_index, x.size(0))
        return layer_fn.layer_forward_eager(x, edge_attr, self.fused_linear.weight, self.linear_3.weight,
                                            self.linear_out.weight, graph)


class EGNNLayer(torch.nn.Module):
    """MPNNLayerSum plus a coordinate update (no counterpart in the reference).

    h, pos -> relu(linear_out([h | sum_e m_e])),  pos + sum_e (pos_i - pos_j) * coord_linear(m_e)
    with m_e = relu(xs_i + xt_j + linear_3(edge_attr)) as in models/layers.py:104-110,128 and coord_linear a
    bias-free Linear(H, 1).  The message sum and the coordinate update share one edge pass, which exists for the
    16-byte vector layout only: ``out_channels`` must be a multiple of 4 (8 for bfloat16 models) and at most 256.
    """

    def __init__(self, in_channels, edge_channels, out_channels):
        super().__init__()
        if out_channels % 4 or not 0 < out_channels <= 256:
            raise ValueError(f"EGNNLayer needs out_channels to be a multiple of 4 in [4, 256], got {out_channels}")
        self.in_channels, self.edge_channels, self.out_channels = in_channels, edge_channels, out_channels
        self.fused_linear = Linear(in_channels, out_channels * 2, bias=False)
        self.linear_3 = Linear(edge_channels, out_channels, bias=False)
        self.linear_out = Linear(out_channels + in_channels, out_channels, bias=False)
        self.coord_linear = Linear(out_channels, 1, bias=False)

    def forward(self, x, pos, edge_index, edge_attr):
        if torch.compiler.is_compiling():      # under a tracing compiler: the marked forms (a graph break each)
            graph = graph_for(edge_index, x.size(0))
            return layer_fn.egnn_layer_forward(x, pos, edge_attr, self.fused_linear.weight, self.linear_3.weight,
                                               self.linear_out.weight, self.coord_linear.weight.view(-1), graph)
        graph = graph_for_eager(edge_index, x.size(0))
        return layer_fn.egnn_layer_forward_eager(x, pos, edge_attr, self.fused_linear.weight, self.linear_3.weight,
                                                 self.linear_out.weight, self.coord_linear.weight.view(-1), graph)
