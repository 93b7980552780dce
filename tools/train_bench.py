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
if "caller" in sys.argv[1:]:       # every backward on the calling thread (mpnn_amd.train.backward_on_caller, here for the whole run)
    torch.autograd.set_multithreading_enabled(False)
    print("backward on the calling thread", flush=True)


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

# ------------------------------------------------------------------------------------------------
# full-graph transductive step (Flickr shape): the reference's step literally, the fused loss/optimizer,
# and the same step replayed from a HIP graph
# ------------------------------------------------------------------------------------------------
import types
from mpnn_amd import MPNN
from mpnn_amd.loss import masked_cross_entropy
from mpnn_amd.optim import FlatAdam
from mpnn_amd.train import CapturedStep

fd = types.SimpleNamespace(**synth.flickr(dev, 0))


def timeit(fn, steps=30, warmup=8):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for mode in ("sum", "cat"):
    torch.manual_seed(0)
    m1 = MPNN(500, 64, 7, num_layers=2, mode=mode).to(dev)
    o1 = torch.optim.Adam(m1.parameters(), lr=1e-3)

    def literal():
        o1.zero_grad()
        out = m1(fd.x, fd.edge_index, fd.edge_attr)
        loss = F.cross_entropy(out[fd.train_mask], fd.y[fd.train_mask])
        loss.backward()
        o1.step()
        return loss

    m2 = MPNN(500, 64, 7, num_layers=2, mode=mode).to(dev)
    o2 = FlatAdam(m2.parameters(), lr=1e-3)

    def fused():
        o2.zero_grad()
        loss = masked_cross_entropy(m2(fd.x, fd.edge_index, fd.edge_attr), fd.y, fd.train_mask)
        loss.backward()
        o2.step()
        return loss

    t_lit, t_fused = timeit(literal), timeit(fused)
    cap = CapturedStep(fused)
    t_cap = timeit(cap.replay)
    # the reference's own form: compiled_model = torch.compile(model) (experiments/flickr_exp.py:30)
    m3 = torch.compile(m1)

    def literal_compiled():
        o1.zero_grad()
        out = m3(fd.x, fd.edge_index, fd.edge_attr)
        loss = F.cross_entropy(out[fd.train_mask], fd.y[fd.train_mask])
        loss.backward()
        o1.step()
        return loss

    t_comp = timeit(literal_compiled)
    print(f"flickr full-graph model step ({mode}): reference-literal loss+Adam {t_lit:.3f} ms  the same with "
          f"torch.compile(model) {t_comp:.3f} ms  fused loss+FlatAdam {t_fused:.3f} ms  HIP-graph replay {t_cap:.3f} ms",
          flush=True)

# ------------------------------------------------------------------------------------------------
# fixed-size point-cloud batches: every batch is written over the same buffers and the training step is
# replayed from a HIP graph
# ------------------------------------------------------------------------------------------------
mm2 = MPNN_ModelNet(3, 1, 64, 10, num_layers=2, mode="sum").to(dev)
om2 = FlatAdam(mm2.parameters(), lr=1e-3)
rng = np.random.default_rng(1)
fixed = m.collate(rng.permutation(m.num_graphs)[:32])


def cloud_step():
    om2.zero_grad()
    loss = F.cross_entropy(mm2(fixed.pos, fixed.edge_index, fixed.batch, fixed.edge_attr), fixed.y)
    loss.backward()
    om2.step()
    return loss


cap = CapturedStep(cloud_step)


def batch_and_replay():
    m.collate(rng.permutation(m.num_graphs)[:32], out=fixed)
    cap.replay()


print(f"modelnet batch=32: assemble into fixed buffers + HIP-graph replay of the model step "
      f"{timeit(batch_and_replay, 60, 15):.3f} ms/step  (replay alone {timeit(cap.replay, 60, 15):.3f} ms)", flush=True)

# ------------------------------------------------------------------------------------------------
# molecule batches of varying size: one padded batch of fixed capacity, the step recorded once
# ------------------------------------------------------------------------------------------------
from mpnn_amd.train import CapturedBatchStep

for bs in (128, 1024):
    mq2 = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").to(dev)
    fast = CapturedBatchStep(mq2, FlatAdam(mq2.parameters(), lr=1e-3), F.mse_loss, q, batch_size=bs)
    rq = np.random.default_rng(2)
    t = timeit(lambda: fast.train(rq.permutation(q.num_graphs)[:bs]), 60, 15)
    print(f"qm9 batch={bs}: padded batch (capacity {fast.capacity[0]} nodes / {fast.capacity[1]} edges) + HIP-graph "
          f"replay of the model step {t:.3f} ms/step", flush=True)


# ------------------------------------------------------------------------------------------------
# QM9 batch of 128, the reference's step literally, eager and with torch.compile(model) (experiments/qm9_exp.py:34)
# ------------------------------------------------------------------------------------------------
mq3 = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").to(dev)
oq3 = torch.optim.Adam(mq3.parameters(), lr=0.003)
bq = q.collate(np.arange(128))
for name, mdl in (("eager", mq3), ("torch.compile(model)", torch.compile(mq3))):
    def qstep():
        oq3.zero_grad()
        out = mdl(bq.x, bq.edge_index, bq.batch, bq.edge_attr)
        loss = F.mse_loss(out[0], bq.y[0])
        loss.backward()
        oq3.step()
    print(f"qm9 batch=128 reference-literal step, {name}: {timeit(qstep, 60, 15):.3f} ms", flush=True)
