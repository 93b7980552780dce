"""Randomised sweep of the point-cloud whole-layer kernels (mpnn_region_layer_fwd / _bwd) against the float64 oracle
with flip-aware gradient bars (oracle/flip_bounds.py), MPNN layers (sum and cat form) and the EGNN layer.  Run by hand on
the GPU: python tests/fuzz_regions.py [seed] [cases]

Every case draws island sizes around the kernels' boundaries (16-row half tiles, 32-row tiles, 128-row blocks), rows
with no edge, rows with more than 32 in-edges or out-edges (several staged pieces per row, groups of a wave with
different piece counts), F = 1..4, Fe = 1..8, sum and cat form, float32 and bfloat16.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from mpnn_amd import EGNNLayer, MPNNLayerCat, MPNNLayerSum, islands  # noqa: E402
from mpnn_amd.graph import clear_cache, graph_for                  # noqa: E402
from oracle import flip_bounds as FB                               # noqa: E402

DEV = "cuda:0"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = torch.Generator().manual_seed(seed)
R = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
taken = 0
for c in range(cases):
    f, fe = R(1, 4), [1, 1, R(2, 8)][R(0, 2)]
    egnn = R(0, 2) == 0
    mode = "sum" if egnn or R(0, 1) else "cat"
    dtype = torch.bfloat16 if R(0, 3) == 0 else torch.float32
    count = R(1, 10)
    edge = [16, 32, 128][R(0, 2)]
    sizes = [max(33, edge * R(1, 6) + R(-2, 2)) for _ in range(count)]
    sizes[0] = max(sizes)
    if R(0, 4) == 0:
        sizes[0] = R(1500, 2048)
    if egnn:                                                      # with coordinates: islands of up to 1024 nodes
        sizes = [min(v, 1024) for v in sizes]
    src, dst, off = [], [], 0
    for nn in sizes:
        e = R(nn, 8 * nn)
        src.append(torch.randint(0, nn, (e,), generator=g) + off)
        dst.append(torch.randint(0, nn, (e,), generator=g) + off)
        src.append(torch.arange(nn - 1) + off)                  # a chain keeps the island in one piece
        dst.append(torch.arange(1, nn) + off)
        for _ in range(R(0, 2)):                                  # hubs: a row with many in-edges, one with many out-edges
            hub, fan = R(0, nn - 1), R(33, 150)
            src.append(torch.randint(0, nn, (fan,), generator=g) + off)
            dst.append(torch.full((fan,), hub + off))
            hub = R(0, nn - 1)
            src.append(torch.full((fan,), hub + off))
            dst.append(torch.randint(0, nn, (fan,), generator=g) + off)
        off += nn
    ei = torch.stack([torch.cat(src), torch.cat(dst)]).long()
    ei = ei[:, torch.randperm(ei.size(1), generator=g)]
    n, h = off, 64
    x = torch.randn(n, f, generator=g).to(dtype)
    ea = torch.randn(ei.size(1), fe, generator=g).to(dtype)
    r = torch.randn(n, h, generator=g)
    pos, rp = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g)
    torch.manual_seed(1000 * seed + c)
    layer = EGNNLayer(f, fe, h) if egnn else (MPNNLayerSum if mode == "sum" else MPNNLayerCat)(f, fe, h)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(p.to(dtype).float())
    sd = {k: v.detach().double() for k, v in layer.state_dict().items()}
    if mode == "sum":
        wn, we, wo = sd["fused_linear.weight"], sd["linear_3.weight"], sd["linear_out.weight"]
    else:
        (wn, we), wo = FB.cat_to_sum_weights(sd["linear.weight"], f), sd["linear_out.weight"]
    if egnn:
        trace = FB.LayerTrace(x.double(), ei, ea.double(), wn, we, wo, FB.DELTA[dtype], pos, sd["coord_linear.weight"])
        want = trace.backward(r, gpos=rp)
    else:
        trace = FB.LayerTrace(x.double(), ei, ea.double(), wn, we, wo, FB.DELTA[dtype])
        want = trace.backward(r)
    clear_cache()
    layer = layer.to(DEV).to(dtype)
    xg = x.to(DEV).requires_grad_(True)
    pg = pos.to(DEV).requires_grad_(True)
    eig = ei.to(DEV)
    if egnn:
        y, pos_out = layer(xg, pg, eig, ea.to(DEV))
        ((y.float() * r.to(DEV)).sum() + (pos_out * rp.to(DEV)).sum()).backward()
    else:
        y = layer(xg, eig, ea.to(DEV))
        (y.float() * r.to(DEV)).sum().backward()
    graph = graph_for(eig, n)
    plan = islands.plan_for(graph, f, h, fe, egnn)
    used = plan is not None and plan.region_nodes > 0
    taken += used
    grads = {k: p.grad for k, p in layer.named_parameters()}
    rtol = 1e-5 if dtype == torch.float32 else 2e-2
    none = (None, None)
    entries = [("y", y, (trace.y,) + none), ("x", xg.grad, want["x"], 4.0), ("w_out", grads["linear_out.weight"], want["w_out"], 4.0)]
    if mode == "sum":
        entries += [("w_nodes", grads["fused_linear.weight"], want["w_nodes"], 4.0), ("w_edge", grads["linear_3.weight"], want["w_edge"], 4.0)]
    else:
        lin = tuple(FB.cat_grad_from_sum(want["w_nodes"][i], want["w_edge"][i], h) for i in range(3))
        entries.append(("linear", grads["linear.weight"], lin, 4.0))
    if egnn:
        entries += [("pos'", pos_out, (trace.pos_out,) + none), ("pos", pg.grad, want["pos"], 4.0),
                    ("w_pos", grads["coord_linear.weight"].reshape(-1), want["w_pos"], 4.0)]
    rep = FB.report(entries, rtol)
    tag = (f"case {c}: {'egnn' if egnn else mode} {str(dtype)[6:]} F={f} Fe={fe} islands={len(sizes)} largest={max(sizes)} nodes={n} "
           f"edges={ei.size(1)} region={used}")
    if not rep["ok"]:
        print("FAILED", tag, "::", rep["worst"])
        sys.exit(1)
    print("ok", tag, f"max_rel_err={rep['max_rel_err']:.2e} through_flips={rep['through_flips']}")
print(f"{cases} cases passed, {taken} on the region kernels")
