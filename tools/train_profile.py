"""cProfile of the per-batch training loop (host side) on QM9-shaped batches."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from mpnn_amd import MPNN_QM9, synth
from mpnn_amd.data import GraphStore

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
q = GraphStore.from_batched(synth.qm9(dev, 0, 4096))
model = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
rng = np.random.default_rng(0)
ids_list = [rng.permutation(q.num_graphs)[:128] for _ in range(220)]
fixed = q.collate(ids_list[0])


def loop(lst):
    for ids in lst:
        b = q.collate(ids) if mode == "fused" else fixed
        opt.zero_grad()
        out = model(b.x, b.edge_index, b.batch, b.edge_attr)
        loss = F.mse_loss(out[0], b.y[0])
        loss.backward()
        opt.step()
    torch.cuda.synchronize()


loop(ids_list[:20])
pr = cProfile.Profile()
pr.enable()
loop(ids_list[20:])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
