"""Where the waves of k_gemm_nt_ws spend their time (library built with -DMPNN_WS_PROF, named by MPNN_AMD_LIB).
usage: ws_prof.py N K M"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import _lib
from mpnn_amd.dense import gemm_nt
n, k, m = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (89_250, 500, 128)
dev = torch.device("cuda:0")
lib = _lib.load()
lib.mpnn_ws_prof_buffer.restype = ctypes.c_int32
lib.mpnn_ws_prof_buffer.argtypes = [ctypes.c_void_p]
buf = torch.zeros((512 * 8, 4), dtype=torch.int64, device=dev)
x = torch.randn(n, k, device=dev)
w = torch.randn(m, k, device=dev)
for _ in range(3):
    gemm_nt(x, w)
torch.cuda.synchronize()
lib.mpnn_ws_prof_buffer(buf.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
gemm_nt(x, w)
b.record()
torch.cuda.synchronize()
lib.mpnn_ws_prof_buffer(None)
t = buf.cpu().double().view(-1, 8, 4)
live = t[:, 0, 0] > 0
t = t[live]
print(f"N={n} K={k} M={m}: {a.elapsed_time(b) * 1e3:.1f} us for both launches; {int(live.sum())} blocks")
for name, sl in (("consumers", slice(0, 4)), ("producers", slice(4, 8))):
    tot, wait, epi, chunks = (t[:, sl, i] for i in range(4))
    print(f"  {name}: cycles per wave mean {float(tot.mean()):9.0f} max {float(tot.max()):9.0f} | at barriers "
          f"{float(wait.mean()):9.0f} ({float((wait / tot).mean()) * 100:4.1f} %) | epilogue {float(epi.mean()):8.0f} | "
          f"chunks {float(chunks.mean()):.1f} -> {float((tot / chunks).mean()):.0f} cycles per chunk, "
          f"{float(((tot - wait - epi) / chunks).mean()):.0f} busy")
