"""Mean time a wave of the point-cloud whole-layer kernels spends in each phase (library built with
-DMPNN_ISLAND_PROF, named by MPNN_AMD_LIB).  usage: region_prof.py [clouds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth, _lib, MPNNLayerSum, islands
from mpnn_amd.graph import graph_for

dev = torch.device("cuda:0")
d = synth.modelnet(dev, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 32)
x, ea, ei = d["x"], d["edge_attr"], d["edge_index"]
h = 64
layer = MPNNLayerSum(3, 1, h).to(dev)
lib = _lib.load()
lib.mpnn_island_prof_buffer.restype = ctypes.c_int32
lib.mpnn_island_prof_buffer.argtypes = [ctypes.c_void_p]
graph = graph_for(ei, x.size(0))
plan = islands.plan_for(graph, 3, h, 1, False)
assert plan is not None and plan.region_nodes
blocks = graph.islands.count * ((plan.region_nodes + 127) // 128)
buf = torch.zeros((blocks * 8, 16), dtype=torch.int64, device=dev)
w = [p.detach() for p in (layer.fused_linear.weight, layer.linear_3.weight, layer.linear_out.weight)]
gy = torch.randn(x.size(0), h, device=dev)


def report(name, slots):
    t = buf.cpu().double() * 0.01   # 100 MHz ticks -> us
    live = t[:, 0] > 0
    t = t[live]
    start = t[:, 0].min()
    print(f"{name}: {int(live.sum())} waves, first start -> last end {float(t[:, slots - 1].max() - start):.1f} us, "
          f"start spread {float(t[:, 0].max() - start):.1f} us")
    for k in range(1, slots):
        dt = t[:, k] - t[:, k - 1]
        print(f"   phase {k:2d}: mean {float(dt.mean()):7.2f} us   max {float(dt.max()):7.2f}")


for _ in range(3):
    y, _, saved = islands.forward(plan, graph, x, ea, *w)
    islands.backward(plan, graph, gy, y, saved, (True, True, True, True))
torch.cuda.synchronize()
lib.mpnn_island_prof_buffer(buf.data_ptr())
buf.zero_()
y, _, saved = islands.forward(plan, graph, x, ea, *w)
torch.cuda.synchronize()
report("forward (table+weights | walk | zero+aggr+sync | update | store)", 6)
# the backward's three kernels write the same slots: profile them one at a time by what they leave
buf.zero_()
islands.backward(plan, graph, gy, y, saved, (True, True, True, True))
torch.cuda.synchronize()
print("backward: slots 0-3 are overwritten by the pass over the sources (it runs second); 4-8 belong to the first pass")
report("backward", 9)
lib.mpnn_island_prof_buffer(None)
