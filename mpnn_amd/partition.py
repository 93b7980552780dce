"""Multi-GPU execution: one process per GPU, torch.distributed (RCCL on ROCm) between them.

Two schemes, both absent from the reference (which is single-process, single-device):

* Large single graph (Flickr-scale and up): nodes are cut into contiguous DESTINATION ranges
  balanced by in-edge count.  A rank owns the rows x[lo:hi], every edge whose target lies in
  its range, and the outputs of its range.  Per layer there is one exchange in each direction:
  forward  - all-gather of the source-side projection B (the "halo" rows a rank's edges read);
  backward - reduce-scatter of the gradient with respect to the gathered B back to the owners.
  Weight gradients are summed with one flat all-reduce.

* Batched small graphs (QM9, ModelNet): plain data parallelism; the only exchange is the flat
  gradient all-reduce (averaged).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import dense, layer_fn, ops
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
    """The piece of a destination-range partition owned by ``rank``.

    ``graph`` is an MPGraph with ``num_dst = hi - lo`` local destination rows and
    ``num_src = world * n_pad`` source rows, addressed in the layout the halo all-gather
    produces: rank q's rows occupy [q * n_pad, q * n_pad + (bounds[q+1] - bounds[q])).
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
        b = torch.tensor(self.bounds, dtype=torch.int64, device=edge_index.device)
        src_l = src[self.edge_ids]
        owner = torch.searchsorted(b, src_l, right=True) - 1
        owner.clamp_(0, world - 1)
        self.local_edge_index = torch.stack([owner * self.n_pad + (src_l - b[owner]), dst[self.edge_ids] - self.lo])
        self.graph = graph_factory(self.local_edge_index, self.n_local, world * self.n_pad)

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.lo:self.hi]

    def local_edges(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.edge_ids]


def _via_host(t: torch.Tensor, group) -> bool:
    """gloo moves host memory: device tensors are staged through the CPU (rehearsals of the multi-rank path on
    one GPU, or hosts without RCCL); with the nccl backend (RCCL) tensors go device to device."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_into(out: torch.Tensor, send: torch.Tensor, group) -> None:
    if _via_host(send, group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, send, group=group)


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    if _via_host(t, group):
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class _HaloAllGather(torch.autograd.Function):
    """[n_local, H] -> [world * n_pad, H]; the backward returns each owner the sum of what all ranks computed."""

    @staticmethod
    def forward(ctx, b_local, n_pad, world, group):
        n_local, h = b_local.shape
        send = b_local
        if n_local != n_pad or not b_local.is_contiguous():
            send = b_local.new_zeros((n_pad, h))
            send[:n_local] = b_local
        out = b_local.new_empty((world * n_pad, h))
        _all_gather_into(out, send, group)
        ctx.n_local, ctx.n_pad, ctx.group = n_local, n_pad, group
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if dist.get_backend(ctx.group) == "gloo":
            # gloo has no reduce-scatter: all-reduce, then keep the own slice
            _all_reduce_sum(g, ctx.group)
            r = dist.get_rank(ctx.group)
            out = g[r * ctx.n_pad:(r + 1) * ctx.n_pad]
        else:
            out = g.new_empty((ctx.n_pad, g.size(1)))
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=ctx.group)
        return out[:ctx.n_local], None, None, None


class Halo:
    """The two exchanges of a partitioned layer as plain calls (used inside the fused layer node, which carries
    its own backward): gather() = forward of _HaloAllGather, scatter_sum() = its backward."""

    def __init__(self, pg: PartitionedGraph, group=None):
        self.n_local, self.n_pad, self.world, self.group = pg.n_local, pg.n_pad, pg.world, group

    def gather(self, b_local: torch.Tensor) -> torch.Tensor:
        """[n_local, H] (any row stride) -> [world * n_pad, H], rank q's rows at q * n_pad."""
        h = b_local.size(1)
        send = b_local
        if self.n_local != self.n_pad or not b_local.is_contiguous():
            send = b_local.new_empty((self.n_pad, h))
            send[:self.n_local] = b_local
            if self.n_local != self.n_pad:
                send[self.n_local:].zero_()
        out = b_local.new_empty((self.world * self.n_pad, h))
        _all_gather_into(out, send, self.group)
        return out

    def scatter_sum(self, g_all: torch.Tensor, out: torch.Tensor) -> None:
        """out[n_local, H] (any row stride) = this rank's rows of the sum of every rank's g_all."""
        if dist.get_backend(self.group) == "gloo":   # gloo has no reduce-scatter: all-reduce, then keep the own slice
            _all_reduce_sum(g_all, self.group)
            r = dist.get_rank(self.group)
            mine = g_all[r * self.n_pad:(r + 1) * self.n_pad]
        else:
            mine = g_all.new_empty((self.n_pad, g_all.size(1)))
            dist.reduce_scatter_tensor(mine, g_all, op=dist.ReduceOp.SUM, group=self.group)
        out.copy_(mine[:self.n_local])


def halo_all_gather(b_local: torch.Tensor, pg: PartitionedGraph, group=None) -> torch.Tensor:
    if pg.world == 1:
        return b_local
    return _HaloAllGather.apply(b_local, pg.n_pad, pg.world, group)


def partitioned_layer(layer, x_local: torch.Tensor, pg: PartitionedGraph, edge_attr_local: torch.Tensor,
                      group=None, aggregate_fn=None, dense_fns=None) -> torch.Tensor:
    """One MPNN layer on the local destination range (same arithmetic as layer.forward on the whole graph).

    ``aggregate_fn`` / ``dense_fns`` = (project, update) replace the HIP operators by stand-ins with the same
    signatures; they exist for the CPU rehearsals of the exchange logic (the kernels need a GPU)."""
    if isinstance(layer, MPNNLayerSum):
        w_nodes, w_edge = layer.fused_linear.weight, layer.linear_3.weight
    elif isinstance(layer, MPNNLayerCat):
        f = layer.in_channels
        w = layer.linear.weight
        w_nodes, w_edge = torch.cat([w[:, :f], w[:, f:2 * f]], dim=0), w[:, 2 * f:]
    else:
        raise TypeError(f"unsupported layer type {type(layer).__name__}")
    w_out = layer.linear_out.weight
    if (aggregate_fn is None and pg.world > 1 and x_local.dtype == torch.float32
            and layer_fn.supported(x_local, edge_attr_local, w_nodes, w_edge, w_out)):
        # the whole layer as one autograd node with the two exchanges inside
        return layer_fn.layer_forward(x_local, edge_attr_local, w_nodes, w_edge, w_out, pg.graph, Halo(pg, group))
    aggregate_fn = aggregate_fn or ops.aggregate
    project, update = dense_fns or (dense.project, dense.update)
    ab = project(x_local, w_nodes)
    h = ab.size(1) // 2
    b_all = halo_all_gather(ab[:, h:], pg, group)
    aggr = aggregate_fn(ab[:, :h], b_all, edge_attr_local, w_edge, pg.graph)
    return update(x_local, aggr, layer.linear_out.weight)


def partitioned_model(model, x_local, pg: PartitionedGraph, edge_attr_local, group=None, aggregate_fn=None,
                      dense_fns=None):
    """MPNN.forward (reference models/mpnn.py:86-101) on the local range; returns the local rows of the logits."""
    x = x_local
    for layer in model.layers:
        # every layer ends in a relu: the F.relu(x) of the model loop is the identity
        x = partitioned_layer(layer, x, pg, edge_attr_local, group, aggregate_fn, dense_fns)
    return model.output_layer(x)


def allreduce_gradients(module: torch.nn.Module, group=None, average: bool = False) -> None:
    """One all-reduce over a flat buffer holding every parameter gradient (sum, or mean for data parallelism)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = [p for p in module.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1).float() for p in params])
    _all_reduce_sum(flat, group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
