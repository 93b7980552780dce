"""Synthetic graphs of the benchmark shapes (no dataset can be downloaded here).

Each generator is seeded, runs on any torch device and returns a dict with ``x``,
``edge_index`` (int64 [2,E], row 0 = source, row 1 = target, PyG convention), ``edge_attr``
and, where the workload has them, ``batch`` / ``pos`` / ``y``.

Shapes follow the reference's datasets as its code uses them:
  flickr    89,250 nodes, 899,756 directed edges, 500 features, edge_attr = ones(E,4)
            (reference data/dataset.py:36-37)
  qm9       molecules of ~18 atoms / ~38 directed bonds, 11 features, 4 bond types, 19 targets
  modelnet  32 clouds x 1024 points, k-nearest-neighbour edges (k=20), distance edge feature
            normalised by the maximum (reference experiments/modelnet_exp.py:58, T.Distance(norm=True))
  powerlaw  S3DIS-scale stand-in: 4M nodes, 100M edges, power-law in-degree, uniform sources
"""
from __future__ import annotations

import math

import torch


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def flickr(device="cpu", seed=0, num_nodes=89_250, num_edges=899_756, num_features=500, num_classes=7):
    g = _gen(device, seed)
    pairs = num_edges // 2
    a = torch.randint(0, num_nodes, (pairs,), generator=g, device=device)
    off = torch.randint(1, num_nodes, (pairs,), generator=g, device=device)
    b = (a + off) % num_nodes  # never a self loop
    ei = torch.cat([torch.stack([a, b]), torch.stack([b, a])], dim=1)
    ei = ei[:, torch.randperm(ei.size(1), generator=g, device=device)]
    x = torch.rand(num_nodes, num_features, generator=g, device=device)
    x = x / x.sum(dim=1, keepdim=True)  # NormalizeFeatures
    return {
        "x": x, "edge_index": ei, "edge_attr": torch.ones(ei.size(1), 4, device=device),
        "y": torch.randint(0, num_classes, (num_nodes,), generator=g, device=device),
        "train_mask": torch.rand(num_nodes, generator=g, device=device) < 0.5,
        "num_nodes": num_nodes,
    }


def qm9(device="cpu", seed=0, num_graphs=128):
    g = _gen(device, seed)
    sizes = torch.round(torch.randn(num_graphs, generator=g, device=device) * 3 + 18).clamp_(4, 29).long()
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(num_graphs, device=device), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    local = torch.arange(n, device=device) - start[batch]
    # spanning tree: node i>0 of each molecule bonds to a random earlier atom of the same molecule
    child = torch.nonzero(local > 0).squeeze(1)
    r = torch.rand(child.numel(), generator=g, device=device)
    parent = start[batch[child]] + (r * local[child].float()).long().clamp_(max=29)
    parent = torch.minimum(parent, child - 1)
    # ring-closing bonds: about one more bond per molecule than the tree has, to reach ~19 bonds per 18 atoms
    extra_a = start + (torch.rand(num_graphs, generator=g, device=device) * sizes.float()).long()
    extra_b = start + (torch.rand(num_graphs, generator=g, device=device) * sizes.float()).long()
    keep = extra_a != extra_b
    ua = torch.cat([child, extra_a[keep]])
    ub = torch.cat([parent, extra_b[keep]])
    ei = torch.cat([torch.stack([ua, ub]), torch.stack([ub, ua])], dim=1)
    ei = ei[:, torch.randperm(ei.size(1), generator=g, device=device)]
    x = torch.rand(n, 11, generator=g, device=device)
    x = x / x.sum(dim=1, keepdim=True)
    bond = torch.randint(0, 4, (ei.size(1),), generator=g, device=device)
    return {
        "x": x, "edge_index": ei, "edge_attr": torch.nn.functional.one_hot(bond, 4).float(), "batch": batch,
        "pos": torch.randn(n, 3, generator=g, device=device),
        "y": torch.randn(num_graphs, 19, generator=g, device=device), "num_nodes": n,
    }


def modelnet(device="cpu", seed=0, num_clouds=32, points=1024, k=20):
    g = _gen(device, seed)
    pos = torch.rand(num_clouds, points, 3, generator=g, device=device) * 2 - 1
    d = torch.cdist(pos, pos)
    d.diagonal(dim1=1, dim2=2).fill_(float("inf"))
    dist, nbr = d.topk(k, dim=2, largest=False)                       # [C, P, k]
    base = (torch.arange(num_clouds, device=device) * points).view(-1, 1, 1)
    dst = (torch.arange(points, device=device).view(1, -1, 1) + base).expand(-1, -1, k)
    src = nbr + base
    ei = torch.stack([src.reshape(-1), dst.reshape(-1)])              # edges j -> i for every neighbour j of i
    ea = (dist / dist.max()).reshape(-1, 1)
    n = num_clouds * points
    return {
        "x": pos.reshape(n, 3), "pos": pos.reshape(n, 3), "edge_index": ei, "edge_attr": ea.contiguous(),
        "batch": torch.arange(num_clouds, device=device).repeat_interleave(points),
        "y": torch.randint(0, 10, (num_clouds,), generator=g, device=device), "num_nodes": n,
    }


def powerlaw(device="cpu", seed=0, num_nodes=4_000_000, num_edges=100_000_000, num_features=128,
             edge_features=4, alpha=2.1, max_degree=None, feature_dtype=torch.float32, with_features=True):
    """Power-law in-degree (P(deg = k) ~ k^-alpha in the tail), uniform sources, random edge order.

    Node of popularity rank r receives edges with probability ~ (r + r0)^(-1/(alpha-1)); r0 is
    solved so that the expected largest in-degree is ``max_degree`` (default E/100, i.e. 10^6 at
    full size).  Ranks are scattered over node ids by a random permutation.
    """
    g = _gen(device, seed)
    gamma = 1.0 / (alpha - 1.0)
    if max_degree is None:
        max_degree = max(num_edges // 100, 1)
    ranks = torch.arange(num_nodes, dtype=torch.float64, device=device)

    def top_share(r0):
        w = (ranks + r0) ** (-gamma)
        return float(w[0] / w.sum())

    lo, hi = 1e-3, float(num_nodes)
    target = max_degree / num_edges
    for _ in range(60):  # bisection on r0: the top node's share falls as r0 grows
        mid = math.sqrt(lo * hi)
        if top_share(mid) > target:
            lo = mid
        else:
            hi = mid
    w = (ranks + math.sqrt(lo * hi)) ** (-gamma)
    cdf = torch.cumsum(w / w.sum(), 0)
    perm = torch.randperm(num_nodes, generator=g, device=device)
    dst = torch.empty(num_edges, dtype=torch.int64, device=device)
    step = 1 << 24
    for s in range(0, num_edges, step):
        u = torch.rand(min(step, num_edges - s), generator=g, device=device, dtype=torch.float64)
        dst[s:s + u.numel()] = perm[torch.searchsorted(cdf, u).clamp_(max=num_nodes - 1)]
    src = torch.randint(0, num_nodes, (num_edges,), generator=g, device=device)
    out = {"edge_index": torch.stack([src, dst]), "num_nodes": num_nodes}
    if with_features:
        out["x"] = powerlaw_features(device, seed, num_nodes, num_features, dtype=feature_dtype)
        out["edge_attr"] = powerlaw_edge_attr(device, seed, num_edges, edge_features, dtype=feature_dtype)
    return out


# Node features and edge attributes of the power-law graph are drawn in blocks of 2^20 rows, every block from a
# generator of its own (seeded by the data seed and the block number): a rank of a partitioned run can then draw
# exactly its rows of x and its edges' attributes - the values a single GPU draws for them - without ever holding
# the other ranks' share.
_BLOCK = 1 << 20


def _block(device, seed, kind, block, rows, width):
    g = torch.Generator(device=device)
    g.manual_seed(((seed * 2 + kind) << 32) + block + 1)
    draw = torch.randn if kind == 0 else torch.rand
    return draw(rows, width, generator=g, device=device)


def powerlaw_features(device, seed, num_nodes, num_features=128, lo=0, hi=None, dtype=torch.float32):
    """Rows lo .. hi - 1 of the power-law graph's node features (standard normal)."""
    hi = num_nodes if hi is None else hi
    out = torch.empty((hi - lo, num_features), dtype=dtype, device=device)
    for b in range(lo // _BLOCK, (max(hi, lo + 1) - 1) // _BLOCK + 1 if hi > lo else 0):
        b0 = b * _BLOCK
        vals = _block(device, seed, 0, b, min(_BLOCK, num_nodes - b0), num_features)
        a, z = max(lo, b0), min(hi, b0 + vals.size(0))
        out[a - lo:z - lo] = vals[a - b0:z - b0].to(dtype)
    return out


def powerlaw_edge_attr(device, seed, num_edges, edge_features=4, edge_ids=None, dtype=torch.float32):
    """Edge attributes (uniform in [0, 1)) of all edges, or of the edges ``edge_ids`` (any order) - drawn block by
    block, so that only the requested rows are ever kept."""
    count = num_edges if edge_ids is None else edge_ids.numel()
    out = torch.empty((count, edge_features), dtype=dtype, device=device)
    for b in range((num_edges + _BLOCK - 1) // _BLOCK):
        b0 = b * _BLOCK
        rows = min(_BLOCK, num_edges - b0)
        if edge_ids is None:
            out[b0:b0 + rows] = _block(device, seed, 1, b, rows, edge_features).to(dtype)
            continue
        mine = torch.nonzero((edge_ids >= b0) & (edge_ids < b0 + rows)).squeeze(1)
        if mine.numel():
            out[mine] = _block(device, seed, 1, b, rows, edge_features)[edge_ids[mine] - b0].to(dtype)
    return out
