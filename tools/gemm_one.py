"""One NT GEMM shape, a few launches (for counter collection)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt
n, k, m = 4_000_000, 256, 128
x = torch.randn(n, k, device="cuda")
w = torch.randn(m, k, device="cuda")
for _ in range(4):
    gemm_nt(x, w)
torch.cuda.synchronize()
