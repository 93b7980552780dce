"""Time of a graph build (sorted views, work items, island search) from a raw edge_index, per workload: mean wall-clock
time of back-to-back builds with one synchronisation at the end, i.e. what a caller pays per mini-batch.
Usage: python tools/build_bench.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth
from mpnn_amd.graph import MPGraph

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["qm9_128", "qm9", "modelnet", "flickr", "powerlaw_1M"]
for name in names:
    if name == "qm9_128":
        d = synth.qm9(dev, 0, 128)
    elif name == "qm9":
        d = synth.qm9(dev, 0, 1024)
    elif name == "modelnet":
        d = synth.modelnet(dev, 0)
    elif name == "flickr":
        d = synth.flickr(dev, 0)
    else:
        d = synth.powerlaw(dev, 0, num_nodes=1_000_000, num_edges=25_000_000)
    ei, n = d["edge_index"], d["x"].size(0)
    reps = 5 if ei.size(1) > 5_000_000 else 50
    for full in (False, True):
        def build():
            g = MPGraph(ei, n)
            if full:
                g.by_src   # the general kernels' structures as well
            return g
        for _ in range(3):
            build()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            build()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:12s} N={n:8d} E={ei.size(1):9d} {'everything' if full else 'as needed '}: {dt * 1e3:7.3f} ms", flush=True)
