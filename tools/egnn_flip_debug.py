"""Where the 2-layer EGNN model's weight gradients leave the float64 oracle: rows of the first layer's projection
weight gradient with entries outside the bar (a relu decision that falls on the other side of zero moves one whole
row of it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth, EGNN
from oracle import reference_oracle as O

dev = torch.device("cuda:0")
d = synth.qm9(dev, 0, 1024)
x, ea, pos, ei, batch = d["x"].float(), d["edge_attr"].float(), d["pos"].float().clone(), d["edge_index"], d["batch"]
torch.manual_seed(5)
model = EGNN(x.size(1), ea.size(1), 64, 19, num_layers=2)
params = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.state_dict().items()}
b = int(batch.max()) + 1
gen = torch.Generator().manual_seed(6)
r, rp = torch.randn(b, 19, generator=gen), torch.randn(x.size(0), 3, generator=gen) * 0.1
xc, pc = x.double().cpu().requires_grad_(True), pos.double().cpu().requires_grad_(True)
want, wp = O.egnn_model_forward(params, xc, pc, ei.cpu(), ea.double().cpu(), batch.cpu())
((want * r.double()).sum() + (wp * rp.double()).sum()).backward()
m = model.cuda()
xg, pg = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
got, gpos = m(xg, pg, ei, batch, ea, return_pos=True)
((got * r.cuda()).sum() + (gpos * rp.cuda()).sum()).backward()
for k, p in m.named_parameters():
    w = params[k].grad
    g = p.grad.double().cpu()
    s = float(w.abs().max())
    err = (g - w).abs()
    bad = err > 1e-5 * 4 * s + 1e-5 * w.abs()
    rows = sorted(set(torch.nonzero(bad)[:, 0].tolist())) if bad.dim() == 2 else []
    print(f"{k:34s} max err/scale {float(err.max()) / s:.2e}  bad {int(bad.sum())}/{bad.numel()}  rows {rows[:12]}")

# the smallest |pre-activation| of the second layer per column, in float64: a decision can only flip when it is
# within float32 rounding of zero
import torch.nn.functional as F
with torch.no_grad():
    P = {k: v.detach() for k, v in params.items()}
    xd, pd = x.double().cpu(), pos.double().cpu()
    h1, p1 = O.egnn_full_layer(xd, pd, ei.cpu(), ea.double().cpu(), P["layers.0.fused_linear.weight"],
                               P["layers.0.linear_3.weight"], P["layers.0.linear_out.weight"], P["layers.0.coord_linear.weight"])
    h1 = F.relu(h1)
    fused = F.linear(h1, P["layers.1.fused_linear.weight"])
    xs, xt = fused.chunk(2, dim=1)
    es = F.linear(ea.double().cpu(), P["layers.1.linear_3.weight"])
    pre = xs[ei[1].cpu()] + xt[ei[0].cpu()] + es
    small = pre.abs().min(dim=0)
    print("smallest |pre| of layer 2: column", int(small.values.argmin()), float(small.values.min()),
          " column 42:", float(small.values[42]), " typical |pre|:", float(pre.abs().median()))
