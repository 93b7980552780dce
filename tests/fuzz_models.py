"""Randomised sweep of whole models: MPNN (node level) and MPNN_QM9 (graph level, mean pooling) with random widths,
depths, modes and batches against the oracle's model_forward in float64.  Logits are compared entry by entry;
parameter gradients in the Frobenius norm (a pre-activation that is zero to float32 rounding may fall on either side
of a relu, which moves single gradient terms, not the norm).  Usage: python tests/fuzz_models.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib, io
import torch
from mpnn_amd import MPNN, MPNN_QM9
from oracle import reference_oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
R = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
worst_logits = worst_grad = 0.0
for c in range(cases):
    graph_level = bool(R(0, 1))
    mode = "sum" if R(0, 1) else "cat"
    layers = R(1, 3)
    f, h, out = R(1, 60), (R(1, 24) * 4 if R(0, 2) else R(1, 70)), R(1, 20)
    fe = R(1, 5) if graph_level else 4          # MPNN fixes 4 edge channels (models/mpnn.py:72-81)
    n, e = R(2, 2000), R(0, 15000)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
    x, ea = torch.randn(n, f, generator=g), torch.rand(e, fe, generator=g)
    torch.manual_seed(seed * 1000 + c)
    with contextlib.redirect_stdout(io.StringIO()):     # the constructors print their mode like the reference's
        model = MPNN_QM9(f, fe, h, out, num_layers=layers, mode=mode) if graph_level else MPNN(f, h, out, num_layers=layers, mode=mode)
    if graph_level:
        b = R(1, min(50, n))
        batch = torch.sort(torch.cat([torch.arange(b), torch.randint(0, b, (n - b,), generator=g)])).values   # every graph non-empty
        rows = b
    else:
        batch, rows = None, n
    r = torch.randn(rows, out, generator=g)
    params = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    want = O.model_forward(params, mode, x.double(), ei, ea.double(), batch)
    (want * r.double()).sum().backward()
    model = model.to(dev)
    got = model(x.to(dev), ei.to(dev), batch.to(dev), ea.to(dev)) if graph_level else model(x.to(dev), ei.to(dev), ea.to(dev))
    (got * r.to(dev)).sum().backward()
    tag = f"case {c} {'MPNN_QM9' if graph_level else 'MPNN'} {mode} layers={layers} n={n} e={e} f={f} h={h} fe={fe} out={out}"
    top = max(float(want.detach().abs().max()), 1e-30)
    err = float((got.detach().cpu().double() - want.detach()).abs().max()) / top
    worst_logits = max(worst_logits, err)
    assert err < 2e-5, f"{tag}: logits off by {err:.2e} of the largest"
    for k, p in model.named_parameters():
        w = params[k].grad
        if w is None or float(w.norm()) == 0.0:
            continue
        ge = float((p.grad.detach().cpu().double() - w).norm() / w.norm())
        worst_grad = max(worst_grad, ge)
        assert ge < 2e-2, f"{tag}: gradient of {k} off by {ge:.2e} in norm"
    if c % 10 == 9:
        print(f"{c + 1} models ok (worst: logits {worst_logits:.2e} of the largest entry, gradients {worst_grad:.2e} in norm)", flush=True)
print("fuzz ok")
