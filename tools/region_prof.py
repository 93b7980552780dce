"""Mean time a wave of the point-cloud whole-layer kernels spends in each phase (library built with
-DMPNN_ISLAND_PROF, named by MPNN_AMD_LIB).  usage: region_prof.py [clouds] [egnn]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth, _lib, EGNNLayer, MPNNLayerSum, islands
from mpnn_amd.graph import graph_for

dev = torch.device("cuda:0")
d = synth.modelnet(dev, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 32)
x, ea, ei = d["x"], d["edge_attr"], d["edge_index"]
h = 64
egnn = len(sys.argv) > 2 and sys.argv[2] == "egnn"
layer = (EGNNLayer if egnn else MPNNLayerSum)(3, 1, h).to(dev)
lib = _lib.load()
lib.mpnn_island_prof_buffer.restype = ctypes.c_int32
lib.mpnn_island_prof_buffer.argtypes = [ctypes.c_void_p]
graph = graph_for(ei, x.size(0))
plan = islands.plan_for(graph, 3, h, 1, egnn)
assert plan is not None and plan.region_nodes
blocks = graph.islands.count * ((plan.region_nodes + 127) // 128)
buf = torch.zeros((blocks * 8, 16), dtype=torch.int64, device=dev)
w = [p.detach() for p in (layer.fused_linear.weight, layer.linear_3.weight, layer.linear_out.weight)]
extra = (d["pos"].float().contiguous(), layer.coord_linear.weight.detach().reshape(-1)) if egnn else ()
gy = torch.randn(x.size(0), h, device=dev)
gp = torch.randn(x.size(0), 3, device=dev) if egnn else None


def report(name, first, last):
    """Phases first+1 .. last of the timestamps (slot `first` is the reference point)."""
    t = buf.cpu().double() * 0.01   # 100 MHz ticks -> us
    live = t[:, first] > 0
    t = t[live]
    start = t[:, first].min()
    print(f"{name}: {int(live.sum())} waves, first start -> last end {float(t[:, last].max() - start):.1f} us, "
          f"start spread {float(t[:, first].max() - start):.1f} us")
    for k in range(first + 1, last + 1):
        dt = t[:, k] - t[:, k - 1]
        print(f"   phase {k:2d}: mean {float(dt.mean()):7.2f} us   max {float(dt.max()):7.2f}")


for _ in range(3):
    y, _, saved = islands.forward(plan, graph, x, ea, *w, *extra)
    islands.backward(plan, graph, gy, y, saved, (True, True, True, True), gp)
torch.cuda.synchronize()
lib.mpnn_island_prof_buffer(buf.data_ptr())
buf.zero_()
y, _, saved = islands.forward(plan, graph, x, ea, *w, *extra)
torch.cuda.synchronize()
report("forward (1 loads + barrier | 2 edge walk | 3 zero rows, keep sums, barrier | 4 update product | 5 store)", 0, 5)
# the backward's three kernels write the same slots: profile them one at a time by what they leave
buf.zero_()
islands.backward(plan, graph, gy, y, saved, (True, True, True, True), gp)
torch.cuda.synchronize()
report("backward, destination tiles (1 loads, gz, barrier | 2 kept sums, barrier | 3 gW_out tiles, barrier | 4 gz W_out, barrier | "
       "5 ga to memory | 6 edge walk | 7 barrier | 8 weight-gradient shares)", 0, 8)
report("backward, source rows (10 loads + barrier | 11 edge walk | 12 weight-gradient shares)", 9, 12)
lib.mpnn_island_prof_buffer(None)
