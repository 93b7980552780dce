"""Randomised parity sweep on the GPU: GEMM entry points against float64 products, the fused edge operator against a
float64 statement of it and whole layers against the oracle in float64, over odd shapes and both row orders.  Usage: python tests/fuzz_parity.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt, gemm_tn
from mpnn_amd.graph import MPGraph
from mpnn_amd.ops import aggregate_fused

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
R = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))


def close(got, want, rtol, what):
    want = want.to(torch.float64); got = got.detach().cpu().to(torch.float64)
    top = max(float(want.abs().max()), 1e-30) if want.numel() else 1.0
    err = float((got - want).abs().max()) / top if want.numel() else 0.0   # relative to the largest expected entry
    assert err <= rtol, f"{what}: relative error {err:.3e} above {rtol:.1e}"
    return err


worst = 0.0
worst_layer = 0.0
boundary_cases = 0
for c in range(cases):
    # ---- GEMMs
    n, k1, m = R(1, 3000), R(1, 300), R(1, 300)
    k2 = R(0, 1) * R(1, 200)
    x1, w1 = torch.randn(n, k1, generator=g), torch.randn(m, k1, generator=g)
    x2 = torch.randn(n, k2, generator=g) if k2 else None
    w2 = torch.randn(m, k2, generator=g) if k2 else None
    relu = bool(R(0, 1))
    mask = torch.randn(n, k1, generator=g) if R(0, 2) == 0 else None
    want = (x1.double() * ((mask > 0).double() if mask is not None else 1.0)) @ w1.double().t()
    if k2:
        want = want + x2.double() @ w2.double().t()
    if relu:
        want = want.clamp_min(0)
    got = gemm_nt(x1.to(dev), w1.to(dev), x2.to(dev) if k2 else None, w2.to(dev) if k2 else None,
                  mask1=mask.to(dev) if mask is not None else None, relu=relu)
    # products are accumulated over k1 + k2 terms of magnitude ~1: scale the bound by the row norm
    scale = (x1.double().abs() @ w1.double().abs().t()).max() + (x2.double().abs() @ w2.double().abs().t()).max() if k2 else (x1.double().abs() @ w1.double().abs().t()).max()
    err = float((got.cpu().double() - want).abs().max())
    assert err <= 2e-6 * float(scale), f"case {c} gemm_nt n={n} k1={k1} k2={k2} m={m} relu={relu} mask={mask is not None}: {err:.3e} vs {float(scale):.3e}"
    gy, xx = torch.randn(n, m, generator=g), torch.randn(n, k1, generator=g)
    mk = torch.randn(n, m, generator=g) if R(0, 1) else None
    want = (gy.double() * ((mk > 0).double() if mk is not None else 1.0)).t() @ xx.double()
    got = gemm_tn(gy.to(dev), xx.to(dev), mask=mk.to(dev) if mk is not None else None)
    scale = (gy.double().abs().t() @ xx.double().abs()).max()
    err = float((got.cpu().double() - want).abs().max())
    assert err <= 2e-6 * float(scale), f"case {c} gemm_tn n={n} m={m} k={k1} mask={mk is not None}: {err:.3e}"
    # ---- edge operator, both row orders
    nn, e, H, Fe = R(1, 400), R(0, 6000), R(1, 40) * R(1, 4), R(1, 6)
    skew = R(0, 1)
    dst = ((torch.rand(e, generator=g) ** 6 * nn).long().clamp_(max=nn - 1) if skew else torch.randint(0, nn, (e,), generator=g))
    ei = torch.stack([torch.randint(0, nn, (e,), generator=g), dst])
    ab = torch.randn(nn, 2 * H, generator=g)
    ea = torch.rand(e, Fe, generator=g)
    we = torch.randn(H, Fe, generator=g) * 0.5
    r = torch.randn(nn, H, generator=g)
    # float64 statement of the operator (the float32 CPU sum of ~1000 terms per row carries 1e-5 of its own)
    leaves = [t.double().requires_grad_(True) for t in (ab, we)]
    a_, b_ = leaves[0][:, :H], leaves[0][:, H:]
    msg = torch.relu(a_[ei[1]] + b_[ei[0]] + ea.double() @ leaves[1].t())
    want_out = torch.zeros(nn, H, dtype=torch.float64).index_add_(0, ei[1], msg)
    (want_out * r.double()).sum().backward()
    # a message whose pre-activation is zero to float32 rounding may fall on the other side of the relu on the GPU
    pre = (a_[ei[1]] + b_[ei[0]] + ea.double() @ leaves[1].t()).detach()
    at_kink = bool(e) and int((pre.abs() < 1.5e-7 * float(pre.abs().max())).sum()) > 0
    for order in ("natural", "items"):
        graph = MPGraph(ei.to(dev), nn, chunk=R(1, 3) * 16, row_order=order)
        gl = [t.clone().to(dev).requires_grad_(True) for t in (ab, we)]
        out = aggregate_fused(gl[0], ea.to(dev), gl[1], graph)
        (out * r.to(dev)).sum().backward()
        for name, got, want in (("out", out, want_out), ("gAB", gl[0].grad, leaves[0].grad), ("gWe", gl[1].grad, leaves[1].grad)):
            if want.numel() and not (at_kink and name != "out"):
                worst = max(worst, close(got, want.detach(), 2e-5, f"case {c} {order} n={nn} e={e} H={H} Fe={Fe} {name}"))
    # ---- whole layers (fused autograd node when the shapes allow it) against the oracle run in float64
    from mpnn_amd import MPNNLayerCat, MPNNLayerSum
    from oracle import reference_oracle as O
    F_in, Hl = (R(1, 40) * R(1, 4), R(1, 32) * 4) if R(0, 2) else (R(1, 150), R(1, 130))   # a third with odd widths
    nl, el = R(2, 2500), R(0, 20000)
    eil = torch.stack([torch.randint(0, nl, (el,), generator=g), torch.randint(0, nl, (el,), generator=g)])
    xl, eal, rl = torch.randn(nl, F_in, generator=g), torch.rand(el, Fe, generator=g), torch.randn(nl, Hl, generator=g)
    mode = "sum" if R(0, 1) else "cat"
    torch.manual_seed(seed * 1000 + c)
    layer = (MPNNLayerSum if mode == "sum" else MPNNLayerCat)(F_in, Fe, Hl)
    ws = [p.detach().double().requires_grad_(True) for p in layer.parameters()]
    xw = xl.double().requires_grad_(True)
    want = (O.layer_sum if mode == "sum" else O.layer_cat)(xw, eil, eal.double(), *ws)
    (want * rl.double()).sum().backward()
    layer = layer.to(dev)
    xg = xl.to(dev).requires_grad_(True)
    out = layer(xg, eil.to(dev), eal.to(dev))
    (out * rl.to(dev)).sum().backward()
    pairs = [("out", out, want), ("gx", xg.grad, xw.grad)] + [(f"gW{i}", p.grad, w.grad) for i, (p, w) in enumerate(zip(layer.parameters(), ws))]
    # A pre-activation within float32 rounding of zero may fall on the other side of the relu on the GPU: the
    # output hardly changes, the gradients of that row do.  Such cases are counted, not compared.
    with torch.no_grad():
        if mode == "sum":
            ab_ = xw @ ws[0].t()
            pre_m = ab_[eil[1], :Hl] + ab_[eil[0], Hl:] + eal.double() @ ws[1].t()
            w_out = ws[2]
        else:
            pre_m = torch.cat([xw[eil[1]], xw[eil[0]], eal.double()], 1) @ ws[0].t()
            w_out = ws[1]
        aggr_ = torch.zeros(nl, Hl, dtype=torch.float64).index_add_(0, eil[1], pre_m.clamp_min(0))
        pre_u = torch.cat([xw, aggr_], 1) @ w_out.t()
        scale_m = float(pre_m.abs().max()) if el else 1.0
        near = int((pre_m.abs() < 1.5e-7 * scale_m).sum()) + int((pre_u.abs() < 1.5e-7 * float(pre_u.abs().max())).sum())
    if near:
        boundary_cases += 1
        pairs = pairs[:1]
    for name, got, wnt in pairs:
        worst_layer = max(worst_layer, close(got, wnt.detach(), 1e-5, f"case {c} layer {mode} n={nl} e={el} F={F_in} H={Hl} Fe={Fe} {name}"))
    if c % 10 == 9:
        print(f"{c + 1} cases ok (worst edge-operator error so far, relative to the largest entry: {worst:.2e}; layers {worst_layer:.2e}, {boundary_cases} with a pre-activation at the relu boundary: outputs only)", flush=True)
print("fuzz ok")
