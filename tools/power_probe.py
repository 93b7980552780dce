"""Samples rocm-smi (shader clock, socket power) while one kernel runs in a loop.
Usage: python tools/power_probe.py [nt|tn|agg|idle] [seconds]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt, gemm_tn

kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
n = 4_000_000
if kind == "nt":
    x = torch.randn(n, 256, device="cuda"); w = torch.randn(128, 256, device="cuda")
    fn = lambda: gemm_nt(x, w)
elif kind == "tn":
    g = torch.randn(n, 128, device="cuda"); x = torch.randn(n, 128, device="cuda")
    fn = lambda: gemm_tn(g, x)
elif kind == "ntbf16":
    from mpnn_amd.dense import gemm_nt_bf16
    x = torch.randn(n, 256, device="cuda").bfloat16(); w = torch.randn(128, 256, device="cuda").bfloat16()
    fn = lambda: gemm_nt_bf16(x, w)
elif kind == "copy":
    x = torch.randn(n, 256, device="cuda"); y = torch.empty_like(x)
    fn = lambda: y.copy_(x)
elif kind == "mm":
    x = torch.randn(n, 256, device="cuda"); w = torch.randn(128, 256, device="cuda")
    fn = lambda: torch.mm(x, w.t())
else:
    fn = lambda: time.sleep(0.01)
stop = False
def sample():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        keep = [l.split(":", 1)[-1].strip() for l in out.splitlines() if "sclk" in l or "(W)" in l]
        print(" | ".join(keep), flush=True)
        time.sleep(0.7)
fn(); torch.cuda.synchronize()
t = threading.Thread(target=sample); t.start()
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    torch.cuda.synchronize(); it += 50
el = time.time() - t0
stop = True; t.join()
print(f"{kind}: {el / it * 1e3:.3f} ms per call over {it} calls")
