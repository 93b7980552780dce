"""Times the GEMM entry points at the sizes of a mini-batch layer (tens of thousands of rows, widths 3..128) next to
torch.mm.  Usage: python tools/gemm_small_bench.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt, gemm_tn

dev = torch.device("cuda:0")


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    for k1, k2, m, relu, masked in [(3, 0, 128, False, False), (4, 0, 128, False, False), (3, 64, 64, True, False),
                                    (4, 64, 64, True, False), (64, 128, 3, False, True), (64, 128, 4, False, True),
                                    (64, 0, 67, False, False),
                                    (128, 0, 3, False, True), (64, 0, 128, False, False), (64, 64, 64, True, False),
                                    (64, 0, 128, False, True), (128, 0, 64, False, True), (11, 0, 128, False, False),
                                    (500, 0, 128, False, False)]:
        x1 = torch.randn(n, k1, device=dev)
        x2 = torch.randn(n, k2, device=dev) if k2 else None
        w = torch.randn(m, k1 + k2, device=dev)
        w1, w2 = w[:, :k1].contiguous(), (w[:, k1:].contiguous() if k2 else None)
        mask = torch.randn(n, k1, device=dev) if masked else None
        print(f"NT N={n} K={k1}+{k2} M={m} relu={relu} mask={masked}: ", end="", flush=True)
        t = timeit(lambda: gemm_nt(x1, w1, x2, w2, mask1=mask, relu=relu))
        print(f"{t:7.1f} us | torch.mm(first seg) ", end="", flush=True)
        t2 = timeit(lambda: torch.mm(x1, w1.t()))
        print(f"{t2:7.1f} us", flush=True)
    for m, k, k2, masked in [(128, 3, 0, False), (128, 4, 0, False), (64, 3, 64, True), (64, 4, 64, True), (128, 64, 0, False),
                             (64, 64, 64, True), (128, 11, 0, False),
                             (128, 500, 0, False)]:
        g, x = torch.randn(n, m, device=dev), torch.randn(n, k, device=dev)
        x2 = torch.randn(n, k2, device=dev) if k2 else None
        mask = torch.randn(n, m, device=dev) if masked else None
        print(f"TN N={n} M={m} K={k}+{k2} mask={masked}: ", end="", flush=True)
        t = timeit(lambda: gemm_tn(g, x, mask=mask, x2=x2))
        print(f"{t:7.1f} us", flush=True)   # (torch.mm(g.t(), x) does not return under graph capture at these sizes)


if __name__ == "__main__":
    main()
