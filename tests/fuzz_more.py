"""Randomised parity sweep, second part: k-NN graphs and mean pooling (bit-exact / float64), the EGNN edge operator
against the oracle in float64, bfloat16 layers against the float32 oracle on bf16-rounded inputs.
Usage: python tests/fuzz_more.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import MPNNLayerCat, MPNNLayerSum, egnn_aggregate
from mpnn_amd.graph import MPGraph
from mpnn_amd.knn import knn_graph
from mpnn_amd.ops import global_mean_pool
from oracle import reference_oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
R = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))


def rel(got, want):
    want = want.detach().double()
    top = max(float(want.abs().max()), 1e-30) if want.numel() else 1.0
    return float((got.detach().cpu().double() - want).abs().max()) / top if want.numel() else 0.0


worst = {"egnn": 0.0, "pool": 0.0, "bf16": 0.0}   # bf16: layer outputs
for c in range(cases):
    # ---- k-NN graph: bit-exact
    clouds, points = R(1, 5), R(2, 700)
    k = R(1, min(32, points - 1))
    pos = torch.rand(clouds * points, 3, generator=g) * 2 - 1
    if points > 8:
        pos[points // 2] = pos[0]; pos[points - 1] = pos[0]
    want_ei, want_ea = O.knn_graph(pos, clouds, k)
    ei, ea = knn_graph(pos.to(dev), clouds, k)
    assert torch.equal(ei.cpu(), want_ei) and torch.equal(ea.cpu(), want_ea), f"case {c} knn clouds={clouds} points={points} k={k}"
    # ---- mean pooling
    n, H, B = R(1, 3000), R(1, 150), R(1, 60)
    batch = torch.sort(torch.randint(0, B, (n,), generator=g)).values
    x = torch.randn(n, H, generator=g)
    xw = x.double().requires_grad_(True)
    want = O.global_mean_pool(xw, batch, B)
    r = torch.randn(B, H, generator=g)
    (want * r.double()).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    got = global_mean_pool(xg, batch.to(dev), B)
    (got * r.to(dev)).sum().backward()
    worst["pool"] = max(worst["pool"], rel(got, want), rel(xg.grad, xw.grad))
    assert worst["pool"] < 1e-5, f"case {c} pool n={n} H={H} B={B}: {worst['pool']:.2e}"
    # ---- EGNN edge operator
    # (at least two nodes: with one node every edge is a self loop, the two halves of the coordinate gradient cancel
    # exactly in float64 and only to float32 rounding of their thousands of terms on the GPU)
    n, e, H, Fe = R(2, 500), R(0, 8000), R(1, 64) * 4, R(1, 5)     # the EGNN pass exists for widths 4, 8, .., 256
    skew = R(0, 1)
    dst = ((torch.rand(e, generator=g) ** 6 * n).long().clamp_(max=n - 1) if skew else torch.randint(0, n, (e,), generator=g))
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), dst])
    a, b = torch.randn(n, H, generator=g), torch.randn(n, H, generator=g)
    ea = torch.rand(e, Fe, generator=g)
    we, wp = torch.randn(H, Fe, generator=g) * 0.5, torch.randn(H, generator=g) / H
    p3 = torch.rand(n, 3, generator=g) * 2 - 1
    r, rp = torch.randn(n, H, generator=g), torch.randn(n, 3, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (a, b, we, p3, wp)]
    aggr_w, pos_w = O.egnn_layer(leaves[0], leaves[1], ea.double(), leaves[2], leaves[3], leaves[4], ei)
    ((aggr_w * r.double()).sum() + (pos_w * rp.double()).sum()).backward()
    graph = MPGraph(ei.to(dev), n, chunk=R(1, 3) * 16, row_order=("natural", "items")[R(0, 1)])
    gl = [t.to(dev).requires_grad_(True) for t in (a, b, we, p3, wp)]
    aggr_g, pos_g = egnn_aggregate(gl[0], gl[1], ea.to(dev), gl[2], gl[3], gl[4], graph)
    ((aggr_g * r.to(dev)).sum() + (pos_g * rp.to(dev)).sum()).backward()
    errs = [rel(aggr_g, aggr_w), rel(pos_g, pos_w)] + [rel(t.grad, w.grad) if w.grad is not None and w.grad.numel() else 0.0 for t, w in zip(gl, leaves)]
    # a message whose pre-activation is zero to float32 rounding may fall on the other side of the relu on the GPU:
    # the outputs agree, the gradients that pass through the mask differ by that one term
    pre = (a.double()[ei[1]] + b.double()[ei[0]] + ea.double() @ we.double().t())
    if e and int((pre.abs() < 1.5e-7 * float(pre.abs().max())).sum()) > 0:
        errs = errs[:2]
    worst["egnn"] = max(worst["egnn"], *errs)
    assert worst["egnn"] < 2e-5, f"case {c} egnn n={n} e={e} H={H} Fe={Fe} skew={skew}: {errs}"
    # ---- bfloat16 layer (widths that are multiples of 8 take the fused node, others the composed path)
    n, e, f, h, fe = R(2, 1500), R(0, 12000), R(1, 20) * 8, R(1, 16) * 8, R(1, 5)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
    rnd = lambda *s: torch.randn(*s, generator=g).bfloat16().float()
    x, ea, r = rnd(n, f), torch.rand(e, fe, generator=g).bfloat16().float(), rnd(n, h)
    mode = "sum" if R(0, 1) else "cat"
    torch.manual_seed(seed * 1000 + c)
    layer = (MPNNLayerSum if mode == "sum" else MPNNLayerCat)(f, fe, h)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(p.bfloat16().float())
    ws = [p.detach().clone().requires_grad_(True) for p in layer.parameters()]
    xc = x.clone().requires_grad_(True)
    want = (O.layer_sum if mode == "sum" else O.layer_cat)(xc, ei, ea, *ws)
    (want * r).sum().backward()
    layer = layer.to(dev).bfloat16()
    xg = x.to(dev).bfloat16().requires_grad_(True)
    got = layer(xg, ei.to(dev), ea.to(dev).bfloat16())
    (got.float() * r.to(dev)).sum().backward()
    def l2(got_, want_):
        want_ = want_.detach().double().cpu()
        return float((got_.detach().cpu().double() - want_).norm() / max(float(want_.norm()), 1e-30))
    errs = [rel(got.float(), want), l2(xg.grad.float(), xc.grad)] + [l2(p.grad.float(), w.grad) for p, w in zip(layer.parameters(), ws)]
    # Gradients of a relu network in bfloat16 differ from the float32 ones by whole terms wherever a rounded
    # pre-activation changes sign (a few entries, each off by a whole term): they are compared in the Frobenius
    # norm, with the same op sequence run by torch in bfloat16 as the yardstick.  One flipped output mask moves a
    # whole row, 1/sqrt(rows) of the norm, so on these small problems the bound only catches gross errors.
    wb = [w.detach().to(dev).bfloat16().requires_grad_(True) for w in ws]
    xb = x.to(dev).bfloat16().requires_grad_(True)
    ref = (O.layer_sum if mode == "sum" else O.layer_cat)(xb, ei.to(dev), ea.to(dev).bfloat16(), *wb)
    (ref.float() * r.to(dev)).sum().backward()
    base = [rel(ref.float(), want), l2(xb.grad.float(), xc.grad)] + [l2(w2.grad.float(), w.grad) for w2, w in zip(wb, ws)]
    worst["bf16"] = max(worst["bf16"], errs[0])
    assert errs[0] < 2e-2, f"case {c} bf16 layer {mode} n={n} e={e} f={f} h={h} fe={fe}: output error {errs[0]:.3e}"
    for ours, theirs in zip(errs[1:], base[1:]):
        assert ours <= max(0.15, 4.0 * theirs), f"case {c} bf16 layer {mode} n={n} e={e} f={f} h={h} fe={fe}: gradient errors {errs} (torch bf16: {base})"
    if c % 10 == 9:
        print(f"{c + 1} cases ok; worst errors relative to the largest entry: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()), flush=True)
print("fuzz ok")
