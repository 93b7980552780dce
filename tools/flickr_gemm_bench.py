"""Times the dense products of a Flickr-shaped layer step (89,250 rows, 500 input features, 64 hidden channels;
reference experiments/flickr_exp.py) one by one as HIP-graph replays.  Usage: python tools/flickr_gemm_bench.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt, gemm_tn
from tools.gemm_small_bench import timeit

dev = torch.device("cuda:0")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 89_250
    f, h = 500, 64
    x = torch.randn(n, f, device=dev)
    rows = []
    for m in (64, 128, 192):
        w = torch.randn(m, f, device=dev)
        t = timeit(lambda: gemm_nt(x, w))
        rows.append((f"NT x W^T        K={f} M={m}", t, n * (f + m) * 4))
    for k in (64, 128, 192):
        g = torch.randn(n, k, device=dev)
        w = torch.randn(f, k, device=dev)
        t = timeit(lambda: gemm_nt(g, w))
        rows.append((f"NT g W  (gx)    K={k} M={f}", t, n * (f + k) * 4))
    a = torch.randn(n, h, device=dev)
    w = torch.randn(h, h, device=dev)
    t = timeit(lambda: gemm_nt(a, w, relu=True))
    rows.append((f"NT aggr Wa^T    K={h} M={h}", t, n * 2 * h * 4))
    for m in (64, 128, 192):
        g = torch.randn(n, m, device=dev)
        t = timeit(lambda: gemm_tn(g, x))
        rows.append((f"TN g^T x        M={m} K={f}", t, n * (f + m) * 4))
    g = torch.randn(n, h, device=dev)
    y = torch.randn(n, h, device=dev)
    gz = torch.empty_like(g)
    t = timeit(lambda: gemm_tn(g, a, mask=y, g_out=gz))
    rows.append((f"TN (g*m)^T aggr M={h} K={h} + gz", t, n * 4 * h * 4))
    for name, t, b in rows:
        print(f"{name:34s} {t:8.1f} us   {b / t / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
