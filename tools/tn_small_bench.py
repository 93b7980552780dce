"""Weight-gradient GEMM at small-graph shapes: gemm_tn(gy, x, mask=y, x2=aggr, g_out=gz) and gemm_tn(g_ab, x)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_tn
dev = "cuda"
def timeit(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, n, f, h in [("flickr L1", 89250, 500, 64), ("flickr L2", 89250, 64, 64), ("modelnet L2", 32768, 64, 64),
                      ("qm9 L2", 18432, 64, 64), ("mid", 1_000_000, 128, 128)]:
    gy, y, x, aggr = (torch.randn(n, h, device=dev), torch.randn(n, h, device=dev), torch.randn(n, f, device=dev),
                      torch.randn(n, h, device=dev))
    gab = torch.randn(n, 2 * h, device=dev)
    gz = torch.empty_like(gy)
    t1 = timeit(lambda: gemm_tn(gy, x, mask=y, x2=aggr, g_out=gz))
    t2 = timeit(lambda: gemm_tn(gab, x))
    print(f"{name:12s} N={n} F={f} H={h}: gW_out {t1:7.1f} us   gW_n {t2:7.1f} us   [{os.environ.get('MPNN_GEMM_TN','pairs')}]")
