"""Randomised sweep of the training entry points: the fused masked cross-entropy against float64, FlatAdam over
parameter lists of odd sizes against torch.optim.Adam.  Usage: python tests/fuzz_train.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.loss import masked_cross_entropy
from mpnn_amd.optim import FlatAdam

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
R = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
worst_ce = worst_adam = 0.0
for c in range(cases):
    n, C = R(1, 20000), R(1, 60)
    logits = torch.randn(n, C, generator=g) * R(1, 8)
    y = torch.randint(0, C, (n,), generator=g)
    mask = None if R(0, 3) == 0 else torch.rand(n, generator=g) < (R(1, 10) / 10)
    if mask is not None and not mask.any():
        mask[R(0, n - 1)] = True
    lw = logits.double().requires_grad_(True)
    sel = slice(None) if mask is None else mask
    want = torch.nn.functional.cross_entropy(lw[sel], y[sel])
    (gw,) = torch.autograd.grad(want, lw)
    lg = logits.to(dev).requires_grad_(True)
    got = masked_cross_entropy(lg, y.to(dev), None if mask is None else mask.to(dev))
    (gg,) = torch.autograd.grad(got, lg)
    e1 = abs(float(got) - float(want)) / max(abs(float(want)), 1e-30)
    e2 = float((gg.cpu().double() - gw).abs().max()) / max(float(gw.abs().max()), 1e-30)
    worst_ce = max(worst_ce, e1, e2)
    assert e1 < 1e-5 and e2 < 1e-5, f"case {c} masked CE n={n} C={C} mask={mask is not None}: loss {e1:.2e} grad {e2:.2e}"
    # Adam over a ragged list of parameters, a few steps with changing gradients
    shapes = [tuple(R(1, 40) for _ in range(R(1, 2))) for _ in range(R(1, 6))]
    pa = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    lr = 10.0 ** -R(1, 4)
    oa, ob = FlatAdam(pa, lr=lr), torch.optim.Adam(pb, lr=lr)
    for step in range(R(1, 5)):
        grads = [torch.randn(*s, generator=g).to(dev) for s in shapes]
        for ps, opt in ((pa, oa), (pb, ob)):
            opt.zero_grad()
            sum((p * gr).sum() for p, gr in zip(ps, grads)).backward()
            opt.step()
    for a, b in zip(pa, pb):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        worst_adam = max(worst_adam, err)
        assert err < 2e-5, f"case {c} adam shapes={shapes} lr={lr}: {err:.2e}"
    if c % 10 == 9:
        print(f"{c + 1} cases ok (worst: cross-entropy {worst_ce:.2e}, adam {worst_adam:.2e})", flush=True)
print("fuzz ok")
