"""Wall-clock cost of MPGraph construction (CSR + transpose + work items) for the batched workloads."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth
from mpnn_amd.graph import MPGraph

dev = torch.device("cuda:0")
cases = {"qm9_b128": lambda: synth.qm9(dev, 0, 128), "qm9_b1024": lambda: synth.qm9(dev, 0, 1024),
         "modelnet_b32": lambda: synth.modelnet(dev, 0), "flickr": lambda: synth.flickr(dev, 0)}
which = sys.argv[1:] or list(cases)
for name in which:
    d = cases[name]()
    ei, n = d["edge_index"], d["num_nodes"]
    for _ in range(5):
        MPGraph(ei, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        MPGraph(ei, n)
    torch.cuda.synchronize()
    print(f"{name}: N={n} E={ei.size(1)}  graph build {1e3 * (time.perf_counter() - t0) / reps:.3f} ms", flush=True)
