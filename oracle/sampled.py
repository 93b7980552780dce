"""Sampled-row checks of full-size runs against the CPU oracle.  TEST INFRASTRUCTURE ONLY.

A 100 M-edge layer is far beyond what the CPU oracle evaluates in seconds, but every output row and every
gradient depends only on the edges that touch it, so the oracle can be evaluated on just those edges:

  * forward row n needs the in-edges of n and the feature rows of their sources;
  * with an upstream gradient that is zero outside a set D of target rows (the situation the reference's own
    loss creates, training/train.py:22: only graph 0 of a batch enters the loss), EVERY gradient of the layer
    - input rows, all three weights - depends only on the in-edges of D, while the kernels under test still walk
    all edges of the graph;
  * for the bare edge operator with a dense upstream gradient G, the gradient of source row s needs the
    out-edges of s, the target-side rows of their targets and G at those targets.

The subgraph is cut out with index operations on whatever device the inputs live on (bookkeeping only); the
arithmetic runs on the CPU in float64 through oracle.reference_oracle (models/layers.py:104-110,128,140 and
PyG's add-aggregation), so that summation error of the checker itself does not eat the 1e-5 budget on rows
with 10^5..10^6 in-edges.  Nothing under mpnn_amd/ may import this module.
"""
from __future__ import annotations

import torch

from . import flip_bounds as FB
from . import reference_oracle as O


def edges_touching(edge_index: torch.Tensor, rows: torch.Tensor, side: int, num_nodes: int) -> torch.Tensor:
    """Ids of the edges whose `side` end (0 = source, 1 = target) is one of `rows`, ascending (edge order kept)."""
    flag = torch.zeros(num_nodes, dtype=torch.bool, device=edge_index.device)
    flag[rows] = True
    return torch.nonzero(flag[edge_index[side]]).squeeze(1)


def compact(edge_index_sub: torch.Tensor, extra_nodes: torch.Tensor):
    """(nodes, edge_index in positions of `nodes`): `nodes` = sorted ids of every end point plus `extra_nodes`."""
    nodes = torch.unique(torch.cat([edge_index_sub.reshape(-1), extra_nodes.reshape(-1)]))
    return nodes, torch.searchsorted(nodes, edge_index_sub)


def sample_rows(edge_index: torch.Tensor, num_nodes: int, side: int, count: int, hubs: int, seed: int):
    """About `count` distinct rows: the `hubs` rows with the most edges on `side` (0 = source, 1 = target), every
    eighth of the next 64, and `count` uniformly drawn rows (rows without edges included).  Sorted ids."""
    deg = torch.bincount(edge_index[side], minlength=num_nodes)
    top = torch.topk(deg, min(hubs + 64, num_nodes)).indices
    gen = torch.Generator().manual_seed(seed)
    drawn = torch.randint(0, num_nodes, (count,), generator=gen).to(edge_index.device)
    return torch.unique(torch.cat([top[:hubs], top[hubs::8], drawn]))


def layer_sparse_upstream(layer_fn, x, edge_index, edge_attr, weights, rows, gy_rows):
    """One layer (`layer_fn` = reference_oracle.layer_sum or layer_cat, `weights` in its argument order) on the
    target rows `rows`, and ALL gradients of <y[rows], gy_rows>.

    Returns a dict: nodes (ids of the rows of x the gradient can reach, sorted), y_rows [len(rows), H],
    gx [len(nodes), F] (gradient rows of `nodes`; every other row of the input gradient is exactly zero),
    gw (list, one gradient per weight), num_edges (edges evaluated).  All float64, on the CPU."""
    n = x.size(0)
    eids = edges_touching(edge_index, rows, 1, n)
    nodes, ei_sub = compact(edge_index[:, eids], rows)
    xs = x[nodes].detach().double().cpu().requires_grad_(True)
    ea = edge_attr[eids].detach().double().cpu()
    ws = [w.detach().double().cpu().requires_grad_(True) for w in weights]
    y = layer_fn(xs, ei_sub.cpu(), ea, *ws)
    at = torch.searchsorted(nodes, rows).cpu()
    y_rows = y[at]
    (y_rows * gy_rows.detach().double().cpu()).sum().backward()
    return {"nodes": nodes, "y_rows": y_rows.detach(), "gx": xs.grad, "gw": [w.grad for w in ws],
            "num_edges": int(eids.numel())}


def layer_sum_sparse_upstream(x, edge_index, edge_attr, w_fused, w_edge, w_out, rows, gy_rows):
    """layer_sparse_upstream for MPNNLayerSum (models/layers.py:91-140) with named weight gradients."""
    r = layer_sparse_upstream(O.layer_sum, x, edge_index, edge_attr, [w_fused, w_edge, w_out], rows, gy_rows)
    r["gw_fused"], r["gw_edge"], r["gw_out"] = r["gw"]
    return r


def layer_sum_sparse_upstream_with_flip_bounds(x, edge_index, edge_attr, w_nodes, w_edge, w_out, rows, gy_rows, delta):
    """layer_sum_sparse_upstream for results whose relu decisions may differ from the float64 ones (bfloat16
    storage; float32 at scale): the same subgraph evaluated by flip_bounds.LayerTrace, which returns next to
    every gradient how much of it hangs on relus that are zero to within `delta` of their summands' rounding.
    For MPNNLayerCat pass flip_bounds.cat_to_sum_weights(linear.weight, F) and map the gradient back with
    flip_bounds.cat_grad_from_sum.

    Returns {"nodes", "y_rows", "num_edges", "open" (undecided message relus), and "gx", "gw_nodes", "gw_edge",
    "gw_out" as (gradient, bound, var) triples for flip_bounds.report}.  float64, CPU."""
    n = x.size(0)
    eids = edges_touching(edge_index, rows, 1, n)
    nodes, ei_sub = compact(edge_index[:, eids], rows)
    t = FB.LayerTrace(x[nodes], ei_sub, edge_attr[eids], w_nodes, w_edge, w_out, delta)
    at = torch.searchsorted(nodes, rows).cpu()
    gy = torch.zeros(nodes.numel(), gy_rows.size(1), dtype=torch.float64)
    gy[at] = gy_rows.detach().double().cpu()
    out = t.backward(gy)
    return {"nodes": nodes, "y_rows": t.y[at], "num_edges": int(eids.numel()), "open": int(t.open_msg.sum()),
            "gx": out["x"], "gw_nodes": out["w_nodes"], "gw_edge": out["w_edge"], "gw_out": out["w_out"]}


def aggregate_rows(a, b, edge_attr, w_edge, edge_index, g, dst_rows, src_rows):
    """The bare edge operator out[n] = sum_{e -> n} relu((a[n] + b[src e]) + w_edge . ea[e]) with a DENSE upstream
    gradient g: out[dst_rows], d/da on dst_rows (in-edges of those rows), d/db on src_rows (out-edges of those).

    Returns float64 CPU tensors out_rows, ga_rows, gb_rows and the number of edges evaluated."""
    n_dst, n_src = a.size(0), b.size(0)
    res = {}
    # in-edges of the sampled targets: forward rows and the target-side gradient
    eids = edges_touching(edge_index, dst_rows, 1, n_dst)
    sub = edge_index[:, eids]
    d_nodes, d_pos = torch.unique(torch.cat([sub[1], dst_rows])), None
    s_nodes = torch.unique(sub[0])
    ei_sub = torch.stack([torch.searchsorted(s_nodes, sub[0]), torch.searchsorted(d_nodes, sub[1])]).cpu()
    a_sub = a[d_nodes].detach().double().cpu().requires_grad_(True)
    b_sub = b[s_nodes].detach().double().cpu()
    we = w_edge.detach().double().cpu()
    out = _aggregate_two_tables(a_sub, b_sub, edge_attr[eids].detach().double().cpu(), we, ei_sub)
    d_pos = torch.searchsorted(d_nodes, dst_rows).cpu()
    (out[d_pos] * g[dst_rows].detach().double().cpu()).sum().backward()
    res["out_rows"], res["ga_rows"] = out[d_pos].detach(), a_sub.grad[d_pos]
    res["num_edges"] = int(eids.numel())
    # out-edges of the sampled sources: the source-side gradient (their targets' sums are incomplete here, but
    # the gradient of b[s] only runs through the edges that leave s, all of which are present)
    eids = edges_touching(edge_index, src_rows, 0, n_src)
    sub = edge_index[:, eids]
    d_nodes = torch.unique(sub[1])
    s_nodes = torch.unique(torch.cat([sub[0], src_rows]))
    ei_sub = torch.stack([torch.searchsorted(s_nodes, sub[0]), torch.searchsorted(d_nodes, sub[1])]).cpu()
    a_sub = a[d_nodes].detach().double().cpu()
    b_sub = b[s_nodes].detach().double().cpu().requires_grad_(True)
    out = _aggregate_two_tables(a_sub, b_sub, edge_attr[eids].detach().double().cpu(), we, ei_sub)
    (out * g[d_nodes].detach().double().cpu()).sum().backward()
    res["gb_rows"] = b_sub.grad[torch.searchsorted(s_nodes, src_rows).cpu()]
    res["num_edges"] += int(eids.numel())
    return res


def _aggregate_two_tables(a, b, edge_attr, w_edge, edge_index):
    """reference_oracle.aggregate_sum_form with separate target and source tables (models/layers.py:128 + add)."""
    es = torch.nn.functional.linear(edge_attr, w_edge)
    msg = torch.relu(torch.add(torch.add(a.index_select(0, edge_index[1]), b.index_select(0, edge_index[0])), es))
    idx = edge_index[1].view(-1, 1).expand_as(msg)
    return msg.new_zeros(a.size(0), msg.size(1)).scatter_add_(0, idx, msg)


def error_report(pairs, rtol: float, kink_frac: float = 1e-5) -> dict:
    """Compares (name, got, want[, floor]) tuples under |got - want| <= floor * rtol * max|want| + rtol * |want|
    (floor 1 unless given; the GPU tests use 4 for gradients).

    kink_frac: share of the entries of one tensor that may lie outside (two correct evaluations of a relu network
    can put a pre-activation that is zero to float32 rounding on different sides; a gradient then differs by one
    whole term).  Returns {"ok", "max_rel_err" (largest |got - want| / max|want| over the tensors, taken over the
    entries inside the bound), "entries", "outside" (entries beyond the bound, at most kink_frac per tensor),
    "worst"}."""
    worst, max_rel, entries, outside, ok = "", 0.0, 0, 0, True
    for name, got, want, *rest in pairs:
        floor = rest[0] if rest else 1.0      # optional 4th entry: the absolute floor in units of rtol * max|want|
        got, want = got.detach().double().cpu(), want.detach().double().cpu()
        if got.shape != want.shape:
            return {"ok": False, "max_rel_err": float("inf"), "entries": entries, "outside": outside,
                    "worst": f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"}
        if want.numel() == 0:
            continue
        scale = max(float(want.abs().max()), 1e-30)
        err = (got - want).abs()
        out = err > rtol * floor * scale + rtol * want.abs()
        bad = int(out.sum())
        rel = float(err[~out].max()) / scale if bad < want.numel() else float("inf")   # over the entries inside
        entries += want.numel()
        outside += bad
        if rel > max_rel:
            max_rel, worst = rel, name
        if bad > kink_frac * want.numel() or not bool(torch.isfinite(got).all()):
            ok = False
            worst = f"{name}: {bad}/{want.numel()} entries outside rtol {rtol}"
            break
    return {"ok": ok, "max_rel_err": max_rel, "entries": entries, "outside": outside, "worst": worst}
