"""Host-side profile (cProfile, calling thread) of whole eager training steps on one cached QM9-shaped batch:
where the Python time of forward + loss + backward + Adam goes.  usage: train_profile.py [batch] [steps]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mpnn_amd import MPNN_QM9, synth
from mpnn_amd.data import GraphStore

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)       # the backward's Python shows up in the profile
store = GraphStore.from_batched(synth.qm9(dev, 0, 4096))
b = store.collate(list(range(batch)))
model = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)


def parts():
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    out = model(b.x, b.edge_index, b.batch, b.edge_attr); t.append(time.perf_counter())
    loss = F.mse_loss(out[0], b.y[0]); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    return [t[i + 1] - t[i] for i in range(5)]


for _ in range(30):
    parts()
torch.cuda.synchronize()
t0 = time.perf_counter()
acc = [0.0] * 5
for _ in range(steps):
    acc = [a + p for a, p in zip(acc, parts())]
torch.cuda.synchronize()
print(f"batch {batch}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step; host us: " +
      "  ".join(f"{n} {a / steps * 1e6:.0f}" for n, a in zip(("zero_grad", "forward", "loss", "backward", "adam"), acc)))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    parts()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
