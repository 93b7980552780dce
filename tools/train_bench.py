"""Per-batch training step on the batched workloads (QM9-shaped molecules, ModelNet-shaped clouds):
assemble the mini-batch, run the 2-layer model forward + loss + backward, and step Adam.

  general   the batch arrives as fresh tensors (what a DataLoader hands over): the index structure is rebuilt
            from edge_index inside the first layer call
  fused     GraphStore.collate writes the batch and its index structure in one library call
  cached    the same batch every step (structure cached) - the floor without any batching cost
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from mpnn_amd import MPNN_QM9, MPNN_ModelNet, synth
from mpnn_amd.data import GraphStore
from mpnn_amd.graph import clear_cache

dev = torch.device("cuda:0")


def run(name, store, model, loss_fn, batch_size, steps=60, warmup=15):
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(0)
    ids_list = [rng.permutation(store.num_graphs)[:batch_size] for _ in range(steps + warmup)]

    def train(b):
        opt.zero_grad()
        out = model(b.x, b.edge_index, b.batch, b.edge_attr)
        loss = loss_fn(out, b.y)
        loss.backward()
        opt.step()

    res = {}
    for mode in ("general", "fused", "cached"):
        clear_cache()
        fixed = store.collate(ids_list[0])
        for i, ids in enumerate(ids_list):
            if i == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if mode == "cached":
                b = fixed
            else:
                b = store.collate(ids)
                if mode == "general":
                    clear_cache()          # drop the registered structure: the layers rebuild it from edge_index
            train(b)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ids in ids_list:
        store.collate(ids)
    torch.cuda.synchronize()
    res["collate only"] = (time.perf_counter() - t0) / len(ids_list) * 1e3
    b = store.collate(ids_list[0])
    print(f"{name} batch={batch_size} N={b.num_nodes} E={b.edge_index.size(1)}: " +
          "  ".join(f"{k} {v:.3f} ms/step" for k, v in res.items()), flush=True)


torch.manual_seed(0)
q = GraphStore.from_batched(synth.qm9(dev, 0, 4096))
mq = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").to(dev)
mse0 = lambda out, y: F.mse_loss(out[0], y[0])     # the reference's loss on graph 0 (training/train.py:22)
run("qm9", q, mq, mse0, 128)
run("qm9", q, mq, mse0, 1024)
m = GraphStore.from_batched(synth.modelnet(dev, 0, 128))
mm = MPNN_ModelNet(3, 1, 64, 10, num_layers=2, mode="sum").to(dev)
ce = lambda out, y: F.cross_entropy(out, y)
run("modelnet", m, mm, ce, 32)
