"""Wall-clock breakdown of the host side of one eager layer step (no profiler hooks: perf_counter around the pieces).
usage: eager_breakdown.py [qm9|modelnet] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mpnn_amd import MPNNLayerSum, islands, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "qm9"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
d = synth.qm9(dev, 0, 1024) if kind == "qm9" else synth.modelnet(dev, 0)
x, ei, ea = d["x"].requires_grad_(True), d["edge_index"], d["edge_attr"]
layer = MPNNLayerSum(x.size(1), ea.size(1), 64).to(dev)
r = torch.randn(x.size(0), 64, device=dev)
params = list(layer.parameters())

acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t = time.perf_counter_ns()
        out = fn(*a, **k)
        acc[name] = acc.get(name, 0) + time.perf_counter_ns() - t
        return out
    return wrapper


def loop(fn, n=steps):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    return host / n * 1e6, (time.perf_counter() - t) / n * 1e6


def step():
    for p in params:
        p.grad = None
    x.grad = None
    layer(x, ei, ea).backward(r)


def fwd_nograd():
    with torch.no_grad():
        layer(x, ei, ea)


def fwd_grad():
    layer(x, ei, ea)


print(f"{kind}: [host us, wall us] per call")
def step_gen():
    for p in layer.parameters():
        p.grad = None
    x.grad = None
    layer(x, ei, ea).backward(r)


for _ in range(3):
    print("  whole step          ", "%.1f %.1f" % loop(step, 300))
    with torch.autograd.set_multithreading_enabled(False):
        print("  whole step, backward on the calling thread", "%.1f %.1f" % loop(step, 300))
print("  whole step, layer.parameters() walked every step", "%.1f %.1f" % loop(step_gen, 300))
print("  forward, no_grad    ", "%.1f %.1f" % loop(fwd_nograd))
print("  forward, with node  ", "%.1f %.1f" % loop(fwd_grad))
y = layer(x, ei, ea)
print("  backward only       ", "%.1f %.1f" % loop(lambda: y.backward(r, retain_graph=True)))
print("  torch.empty         ", "%.1f %.1f" % loop(lambda: torch.empty((1024, 64), device=dev)))
islands.forward = timed("islands.forward", islands.forward)
islands.backward = timed("islands.backward", islands.backward)
islands._desc = timed("  _desc", islands._desc)
islands._inputs = timed("  _inputs", islands._inputs)
lib = islands._lib.load()
for name in ("mpnn_island_layer_fwd", "mpnn_island_layer_bwd", "mpnn_region_layer_fwd", "mpnn_region_layer_bwd"):
    setattr(lib, name, timed("  " + name, getattr(lib, name)))
islands._lib.stream_handle = timed("  stream_handle", islands._lib.stream_handle)
acc.clear()
loop(step)
for k, v in acc.items():
    print(f"  {k:30s} {v / (steps + 30) / 1e3:.1f} us per step")
