"""Time of the k-NN graph construction (32 clouds x 1024 points, k = 20 by default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.knn import knn_graph

for clouds, points, k in [(32, 1024, 20), (32, 1024, 8), (8, 4096, 20), (128, 256, 16)]:
    pos = (torch.rand(clouds * points, 3, device="cuda") * 2 - 1)
    for _ in range(3):
        knn_graph(pos, clouds, k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        knn_graph(pos, clouds, k)
    e.record(); torch.cuda.synchronize()
    print(f"{clouds} x {points}, k={k}: {s.elapsed_time(e) / 20:.3f} ms")
