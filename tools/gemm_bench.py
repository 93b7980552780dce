"""Times the MFMA GEMM entry points at the layer's shapes and prints TFLOP/s next to torch (rocBLAS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt, gemm_tn

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    shapes_nt = [(128, 0, 256, False, False), (128, 128, 128, True, False), (128, 0, 128, False, True),
                 (256, 0, 128, False, False), (500, 0, 128, False, False), (500, 64, 64, True, False)]
    for k1, k2, m, relu, masked in shapes_nt:
        nn = n if k1 < 400 else n // 8
        x1 = torch.randn(nn, k1, device=dev)
        x2 = torch.randn(nn, k2, device=dev) if k2 else None
        w = torch.randn(m, k1 + k2, device=dev)
        mask = torch.randn(nn, k1, device=dev) if masked else None
        t = timeit(lambda: gemm_nt(x1, w[:, :k1], x2, w[:, k1:] if k2 else None, mask1=mask, relu=relu))
        t2 = timeit(lambda: torch.mm(x1, w[:, :k1].t()))
        fl = 2.0 * nn * (k1 + k2) * m
        print(f"NT N={nn} K={k1}+{k2} M={m} relu={relu} mask={masked}: {t*1e3:8.3f} ms {fl/t/1e12:6.1f} TF | torch.mm(first seg) {t2*1e3:8.3f} ms")
    for m, k, masked in [(256, 128, False), (128, 128, True), (128, 500, False), (64, 500, True)]:
        nn = n if k < 400 else n // 8
        g, x = torch.randn(nn, m, device=dev), torch.randn(nn, k, device=dev)
        mask = torch.randn(nn, m, device=dev) if masked else None
        t = timeit(lambda: gemm_tn(g, x, mask=mask))
        t2 = timeit(lambda: torch.mm(g.t(), x))
        fl = 2.0 * nn * m * k
        print(f"TN N={nn} M={m} K={k} mask={masked}: {t*1e3:8.3f} ms {fl/t/1e12:6.1f} TF | torch.mm {t2*1e3:8.3f} ms")


if __name__ == "__main__":
    main()
