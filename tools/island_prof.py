"""Mean time a wave of the island kernels spends in each phase (library built with -DMPNN_ISLAND_PROF, named by
MPNN_AMD_LIB).  usage: island_prof.py [mpnn|egnn] [f32|bf16] [F]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth, _lib, MPNNLayerSum, EGNNLayer

kind = sys.argv[1] if len(sys.argv) > 1 else "mpnn"
dtype = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
dev = torch.device("cuda:0")
d = synth.qm9(dev, 0, 1024)
x, ea, pos, ei = d["x"].to(dtype), d["edge_attr"].to(dtype), d["pos"].float(), d["edge_index"]
if len(sys.argv) > 3:
    x = torch.randn(x.size(0), int(sys.argv[3]), device=dev).to(dtype)
h = 64
layer = (EGNNLayer if kind == "egnn" else MPNNLayerSum)(x.size(1), ea.size(1), h).to(dev).to(dtype)
lib = _lib.load()
lib.mpnn_island_prof_buffer.restype = ctypes.c_int32
lib.mpnn_island_prof_buffer.argtypes = [ctypes.c_void_p]
buf = torch.zeros((1024, 16), dtype=torch.int64, device=dev)   # 4 waves x 256 blocks at most
xg = x.clone().requires_grad_(True)
pg = pos.clone().requires_grad_(True)


def run():
    out = layer(xg, pg, ei, ea) if kind == "egnn" else layer(xg, ei, ea)
    return out


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
    o = run()
    (o[0] if kind == "egnn" else o).sum().backward()
torch.cuda.synchronize()
lib.mpnn_island_prof_buffer(buf.data_ptr())
buf.zero_()
with torch.no_grad():
    run()
torch.cuda.synchronize()
report("forward", 5)
buf.zero_()
lib.mpnn_island_prof_buffer(None)
o = run()
lib.mpnn_island_prof_buffer(buf.data_ptr())
(o[0].sum() + o[1].sum() if kind == "egnn" else o.sum()).backward()
torch.cuda.synchronize()
report("backward", 12)
lib.mpnn_island_prof_buffer(None)
