"""One GEMM shape, a few launches (for counter collection):
gemm_one.py [nt|tn|tnmask|flickr_nt|flickr_nt192|flickr_gx|flickr_tn|flickr_tn192]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt, gemm_tn
kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
n = 4_000_000
nf = 89_250                      # Flickr's node count: 500 input features, 64 hidden channels
if kind == "nt":
    x = torch.randn(n, 256, device="cuda")
    w = torch.randn(128, 256, device="cuda")
    fn = lambda: gemm_nt(x, w)
elif kind in ("flickr_nt", "flickr_nt192"):
    m = 192 if kind.endswith("192") else 128
    x = torch.randn(nf, 500, device="cuda")
    w = torch.randn(m, 500, device="cuda")
    fn = lambda: gemm_nt(x, w)
elif kind == "flickr_gx":
    g = torch.randn(nf, 192, device="cuda")
    w = torch.randn(500, 192, device="cuda")
    fn = lambda: gemm_nt(g, w)
elif kind in ("flickr_tn", "flickr_tn192"):
    m = 192 if kind.endswith("192") else 128
    g = torch.randn(nf, m, device="cuda")
    x = torch.randn(nf, 500, device="cuda")
    fn = lambda: gemm_tn(g, x)
else:
    g = torch.randn(n, 128, device="cuda")
    x = torch.randn(n, 128, device="cuda")
    mask = torch.randn(n, 128, device="cuda") if kind == "tnmask" else None
    fn = lambda: gemm_tn(g, x, mask=mask)
for _ in range(4):
    fn()
torch.cuda.synchronize()
