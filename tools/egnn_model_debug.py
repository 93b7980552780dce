"""Per-tensor error of the 2-layer EGNN model on the GPU and of the float32 CPU oracle, both against float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth, EGNN
from oracle import reference_oracle as O

wl = sys.argv[1] if len(sys.argv) > 1 else "qm9"
pos_scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
d = synth.qm9("cpu", 0, 1024) if wl == "qm9" else synth.modelnet("cpu", 0)
x, ea, pos, ei, batch = d["x"], d["edge_attr"], d["pos"].float(), d["edge_index"], d["batch"]
torch.manual_seed(5)
model = EGNN(x.size(1), ea.size(1), 64, 19, 2)
b = int(batch.max()) + 1
gen = torch.Generator().manual_seed(6)
r, rp = torch.randn(b, 19, generator=gen), torch.randn(x.size(0), 3, generator=gen) * pos_scale


def cpu(dt):
    params = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in model.state_dict().items()}
    xc, pc = x.detach().clone().to(dt).requires_grad_(True), pos.detach().clone().to(dt).requires_grad_(True)
    want, wp = O.egnn_model_forward(params, xc, pc, ei, ea.to(dt), batch)
    ((want * r.to(dt)).sum() + (wp * rp.to(dt)).sum()).backward()
    return dict(logits=want.detach(), pos=wp.detach(), gx=xc.grad, gpos=pc.grad, **{k: p.grad for k, p in params.items()})


ref, c32 = cpu(torch.float64), cpu(torch.float32)
m = model.cuda()
xg, pg = x.cuda().requires_grad_(True), pos.cuda().requires_grad_(True)
got, gpos = m(xg, pg, ei.cuda(), batch.cuda(), ea.cuda(), return_pos=True)
((got * r.cuda()).sum() + (gpos * rp.cuda()).sum()).backward()
gpu = dict(logits=got.detach(), pos=gpos.detach(), gx=xg.grad, gpos=pg.grad, **{k: p.grad for k, p in m.named_parameters()})
for k in ref:
    w = ref[k]
    s = float(w.abs().max())
    e_gpu = float((gpu[k].double().cpu() - w).abs().max()) / s
    e_cpu = float((c32[k].double() - w).abs().max()) / s
    print(f"{wl} {k:36s} scale {s:9.3g}  gpu {e_gpu:.2e}  cpu32 {e_cpu:.2e}")
