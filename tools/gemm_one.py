"""One GEMM shape, a few launches (for counter collection): gemm_one.py [nt|tn|tnmask]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt, gemm_tn
kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
n = 4_000_000
if kind == "nt":
    x = torch.randn(n, 256, device="cuda")
    w = torch.randn(128, 256, device="cuda")
    fn = lambda: gemm_nt(x, w)
else:
    g = torch.randn(n, 128, device="cuda")
    x = torch.randn(n, 128, device="cuda")
    mask = torch.randn(n, 128, device="cuda") if kind == "tnmask" else None
    fn = lambda: gemm_tn(g, x, mask=mask)
for _ in range(4):
    fn()
torch.cuda.synchronize()
