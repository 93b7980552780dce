"""Host-side profile (cProfile) of eager layer steps: where the Python time of one forward + backward goes.
usage: eager_profile.py [qm9|modelnet] [steps]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import MPNNLayerSum, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "qm9"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
d = synth.qm9(dev, 0, 1024) if kind == "qm9" else synth.modelnet(dev, 0)
x, ei, ea = d["x"].requires_grad_(True), d["edge_index"], d["edge_attr"]
layer = MPNNLayerSum(x.size(1), ea.size(1), 64).to(dev)
r = torch.randn(x.size(0), 64, device=dev)


def step():
    for p in layer.parameters():
        p.grad = None
    x.grad = None
    layer(x, ei, ea).backward(r)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{kind}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per eager step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
