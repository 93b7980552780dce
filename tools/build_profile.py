"""Where the time of building the index structure of a fresh QM9-shaped batch goes (what a training loop that gets
new tensors every step pays inside its first layer call): host time per piece and the GPU's share.
usage: build_profile.py [batch]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth
from mpnn_amd.graph import MPGraph

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
d = synth.qm9(dev, 0, batch)
ei, n, ea = d["edge_index"], d["num_nodes"], d["edge_attr"]


def build():
    g = MPGraph(ei, n)
    g.sorted_edge_attr(ea, torch.float32, "dst")
    return g


for _ in range(20):
    build()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    build()
torch.cuda.synchronize()
print(f"batch {batch}: N={n} E={ei.size(1)}: build + sorted edge_attr {(time.perf_counter() - t0) / 200 * 1e6:.0f} us")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(200):
    build()
ev[1].record()
torch.cuda.synchronize()
print(f"  (stream time per build {ev[0].elapsed_time(ev[1]) / 200 * 1e3:.0f} us)")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    build()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
