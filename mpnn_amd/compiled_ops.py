"""The layer and the pooling as ``torch.library`` operators, for ``torch.compile``.

The reference wraps every model in ``torch.compile(model)`` (experiments/flickr_exp.py:30, qm9_exp.py:34,
modelnet_exp.py:50).  Traced code cannot look up the cached graph structure, read device pointers or call ctypes,
so while a model is being compiled its layers call the operators below instead of the eager path: each is opaque
to the tracer (one node in the captured graph), carries a shape function for fake tensors and an autograd formula
whose backward is again one operator, and finds the sorted structure of ``edge_index`` in the same cache the
eager path uses - the cache stays outside the traced region, keyed on the tensor as before.  Eager calls and
compiled calls run the same sequence of library kernels and return the same bits.

  mpnn_amd::layer_fwd   (x, edge_index, edge_attr, w_nodes, w_edge, w_out, train) -> (y, ab, aggr, cnt, cnt_big, maskbits)
  mpnn_amd::layer_bwd   (gy, x, edge_index, edge_attr, ab, aggr, cnt, cnt_big, maskbits, y, weights..., needs...) -> 4 gradients
  mpnn_amd::mean_pool / mean_pool_bwd                     global_mean_pool (models/mpnn.py:54,150)
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import _lib, islands, layer_fn, ops
from .graph import BYTE_COUNT_MAX, default_chunk, graph_for


def _mask_words(h: int, dtype) -> int:
    return _lib.load().mpnn_mask_words(int(h), _lib.dtype_code(dtype)) if ops.USE_MASKBITS else 0


def _long_rows_cap(num_edges: int) -> int:
    """Rows the table of exact counts can need for an edge list of this size, from the integer rules of the graph
    build alone (a split row has more than `chunk` edges, an unsplit long row more than 255): the operator's output
    shapes then depend on input SHAPES only, and Dynamo captures it without being told to capture operators with
    data-dependent output shapes."""
    chunk = default_chunk(num_edges)
    return num_edges // (chunk + 1) + (num_edges // (BYTE_COUNT_MAX + 1) if chunk > BYTE_COUNT_MAX else 0) + 1


@torch.library.custom_op("mpnn_amd::layer_fwd", mutates_args=(), device_types="cuda")
def layer_fwd(x: Tensor, edge_index: Tensor, edge_attr: Tensor, w_nodes: Tensor, w_edge: Tensor, w_out: Tensor,
              train: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    graph = graph_for(edge_index, x.size(0))
    x = x if x.is_contiguous() and x.data_ptr() % 16 == 0 else x.clone(memory_format=torch.contiguous_format)
    y, ab, aggr, cnt, maskbits, _ = layer_fn.forward_impl(x, edge_attr, w_nodes, w_edge, w_out, graph, train)
    h = w_out.size(0)
    cnt, cnt_big = cnt if cnt is not None else (None, None)
    if cnt is None:
        cnt = torch.empty((0, h), dtype=torch.uint8, device=x.device)
    if train:
        # a fixed-capacity table (see _long_rows_cap); the backward reads the rows the graph's records name
        cap = _long_rows_cap(edge_index.size(1))
        table = torch.empty((cap, h), dtype=torch.float32, device=x.device)
        if cnt_big is not None:
            if cnt_big.size(0) > cap:
                raise _lib.MpnnError(f"{cnt_big.size(0)} long rows exceed the capacity {cap} derived from the edge "
                                     "count: was the graph structure built with a non-default chunk?")
            table[:cnt_big.size(0)] = cnt_big
        cnt_big = table
    else:
        cnt_big = torch.empty((0, h), dtype=torch.float32, device=x.device)
    if maskbits is None:
        maskbits = torch.empty((edge_index.size(1) if train else 0, _mask_words(h, x.dtype) if train else 0),
                               dtype=torch.int32, device=x.device)
    return y, ab, aggr, cnt, cnt_big, maskbits


@layer_fwd.register_fake
def _(x, edge_index, edge_attr, w_nodes, w_edge, w_out, train):
    n, h = x.size(0), w_out.size(0)
    y = x.new_empty((n, h))
    # the projection carries the x-part of the update as a third block of columns when x is read once (fused_x)
    ab = x.new_empty((n, (3 if layer_fn.fused_x(x.dtype, x.size(1), h) else 2) * h))
    aggr = x.new_empty((n, h))
    cnt = x.new_empty((n if train else 0, h), dtype=torch.uint8)
    # exact counts of the rows longer than a byte can count: a table of fixed capacity
    cnt_big = x.new_empty((_long_rows_cap(edge_index.size(1)) if train else 0, h), dtype=torch.float32)
    maskbits = x.new_empty((edge_index.size(1) if train else 0, _mask_words(h, x.dtype) if train else 0),
                           dtype=torch.int32)
    return y, ab, aggr, cnt, cnt_big, maskbits


@torch.library.custom_op("mpnn_amd::layer_bwd", mutates_args=(), device_types="cuda")
def layer_bwd(gy: Tensor, x: Tensor, edge_index: Tensor, edge_attr: Tensor, ab: Tensor, aggr: Tensor, cnt: Tensor,
              cnt_big: Tensor, maskbits: Tensor, y: Tensor, w_nodes: Tensor, w_edge: Tensor, w_out: Tensor, need_x: bool, need_wn: bool,
              need_we: bool, need_wout: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    graph = graph_for(edge_index, x.size(0))
    x = x if x.is_contiguous() and x.data_ptr() % 16 == 0 else x.clone(memory_format=torch.contiguous_format)
    plan = islands.plan_for(graph, x.size(1), w_out.size(0), edge_attr.size(1), False)
    one_launch = plan is not None
    # the layout layer_fwd's path reads
    ea_t = islands.sorted_attr(plan, graph, edge_attr, x.dtype) if one_launch else graph.sorted_edge_attr(edge_attr, x.dtype)[1]
    bits = maskbits if maskbits.size(1) > 0 else None
    csr = graph.by_dst if not one_launch else None
    has_long = csr is not None and (csr.n_split + csr.n_big) > 0
    counts = (cnt, cnt_big if has_long else None)
    grads = layer_fn.backward_impl(gy, x, ab, aggr, counts, bits, y, ea_t, w_nodes, w_edge, w_out, graph,
                                   (need_x, need_wn, need_we, need_wout))
    none = lambda like: like.new_empty((0,))
    return tuple(g if g is not None else none(x) for g in grads)


@layer_bwd.register_fake
def _(gy, x, edge_index, edge_attr, ab, aggr, cnt, cnt_big, maskbits, y, w_nodes, w_edge, w_out, need_x, need_wn, need_we,
      need_wout):
    e = lambda like, on: torch.empty_like(like, memory_format=torch.contiguous_format) if on else x.new_empty((0,))
    return e(x, need_x), e(w_nodes, need_wn), e(w_edge, need_we), e(w_out, need_wout)


def _layer_setup(ctx, inputs, output):
    x, edge_index, edge_attr, w_nodes, w_edge, w_out, train = inputs
    y, ab, aggr, cnt, cnt_big, maskbits = output
    ctx.set_materialize_grads(False)
    ctx.train = train
    ctx.save_for_backward(x, edge_index, edge_attr, ab, aggr, cnt, cnt_big, maskbits, y, w_nodes, w_edge, w_out)


def _layer_backward(ctx, gy, *unused):
    if not ctx.train:
        raise RuntimeError("mpnn_amd::layer_fwd ran with train=False: no backward information was kept")
    needs = ctx.needs_input_grad
    x = ctx.saved_tensors[0]
    if gy is None:
        gy = torch.zeros((x.size(0), ctx.saved_tensors[-1].size(0)), dtype=x.dtype, device=x.device)
    gx, gwn, gwe, gwout = torch.ops.mpnn_amd.layer_bwd(gy, *ctx.saved_tensors, needs[0], needs[3], needs[4], needs[5])
    return (gx if needs[0] else None, None, None, gwn if needs[3] else None, gwe if needs[4] else None,
            gwout if needs[5] else None, None)


layer_fwd.register_autograd(_layer_backward, setup_context=_layer_setup)


def layer(x, edge_index, edge_attr, w_nodes, w_edge, w_out) -> Tensor:
    """The layer through the registered operator (what the layer modules call while being compiled)."""
    layer_fn._check_dtypes(x, edge_attr, w_nodes, w_edge, w_out)
    if x.dtype == torch.bfloat16 and x.size(1) % 8:
        x, w_nodes, w_out = layer_fn._pad_width(x, w_nodes, w_out)
    train = torch.is_grad_enabled() and any(t.requires_grad for t in (x, w_nodes, w_edge, w_out))
    return torch.ops.mpnn_amd.layer_fwd(x, edge_index, edge_attr, w_nodes, w_edge, w_out, train)[0]


# ----------------------------------------------------------------------------------------------------------------
# global_mean_pool
# ----------------------------------------------------------------------------------------------------------------
# The number of graphs of a batch (batch.max() + 1) is known on the device only, so the pooled output has a
# data-dependent first dimension.  Dynamo breaks the graph at such operators (the model still runs: two captured
# graphs around an eager call, as with PyG's own global_mean_pool, which reads the count back to the host at the
# same place) unless `torch._dynamo.config.capture_dynamic_output_shape_ops` is set.  That switch is process-wide and
# belongs to the application, so importing this package leaves it alone; capture_pooling_in_one_graph() - or
# MPNN_AMD_DYNAMO_CAPTURE=1 in the environment - sets it for callers who want MPNN_QM9 / MPNN_ModelNet / EGNN in ONE
# graph (torch.compile(model, fullgraph=True)).
import os as _os


def capture_pooling_in_one_graph(enable: bool = True) -> None:
    """Sets torch._dynamo.config.capture_dynamic_output_shape_ops (process-wide): the pooled models then compile
    into a single graph."""
    import torch._dynamo.config as _dynamo_config
    _dynamo_config.capture_dynamic_output_shape_ops = bool(enable)


if _os.environ.get("MPNN_AMD_DYNAMO_CAPTURE", "0") == "1":
    capture_pooling_in_one_graph()


@torch.library.custom_op("mpnn_amd::mean_pool", mutates_args=(), device_types="cuda")
def mean_pool(x: Tensor, batch: Tensor) -> Tensor:
    return ops.segment_mean_forward(x, ops.pool_structure(batch, None))


@mean_pool.register_fake
def _(x, batch):
    graphs = torch.library.get_ctx().new_dynamic_size()     # batch.max() + 1: known only on the device
    return x.new_empty((graphs, x.size(1)))


@torch.library.custom_op("mpnn_amd::mean_pool_bwd", mutates_args=(), device_types="cuda")
def mean_pool_bwd(g: Tensor, batch: Tensor) -> Tensor:
    csr = ops.pool_structure(batch, None)
    return ops.segment_mean_backward(g, csr, batch.numel())


@mean_pool_bwd.register_fake
def _(g, batch):
    return g.new_empty((batch.size(0), g.size(1)))


def _pool_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1])


def _pool_backward(ctx, g):
    return torch.ops.mpnn_amd.mean_pool_bwd(g, ctx.saved_tensors[0]), None


mean_pool.register_autograd(_pool_backward, setup_context=_pool_setup)
