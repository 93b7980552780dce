"""Generates tests/golden/*.npz from the CPU oracle (oracle/reference_oracle.py).

Run from the repository root:  python oracle/make_golden.py
The vectors are inputs plus the oracle's float32 outputs and gradients, with a float64
recomputation of the same quantities to budget tolerances.  The reference itself cannot be
executed in this image (its layers import torch_geometric, which is absent), so these
vectors pin the restatement against regressions; they are not outputs of the reference.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reference_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def linear_weight(gen, out_f, in_f):
    bound = 1.0 / (in_f ** 0.5)  # torch.nn.Linear default: U(-1/sqrt(in), 1/sqrt(in))
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound


def make_graph(gen, kind):
    if kind == "tiny":
        # 12 nodes; node 3 and 11 isolated, duplicate edges, self loops, unsorted order
        src = [5, 0, 0, 7, 2, 2, 9, 1, 4, 4, 8, 6, 10, 0, 7, 2]
        dst = [0, 5, 5, 7, 1, 1, 2, 9, 4, 6, 6, 8, 0, 10, 0, 2]
        return 12, torch.tensor([src, dst], dtype=torch.int64)
    if kind == "empty":
        return 5, torch.zeros((2, 0), dtype=torch.int64)
    if kind == "qm9":
        # 5 molecules of 14..22 atoms, ring + random chords, both directions, shuffled
        edges, n = [], 0
        for size in (18, 14, 22, 17, 19):
            ids = torch.arange(n, n + size)
            ring = torch.stack([ids, torch.roll(ids, -1)])
            extra = torch.stack([ids[torch.randint(0, size, (3,), generator=gen)],
                                 ids[torch.randint(0, size, (3,), generator=gen)]])
            und = torch.cat([ring, extra], dim=1)
            edges.append(torch.cat([und, und.flip(0)], dim=1))
            n += size
        ei = torch.cat(edges, dim=1)
        return n, ei[:, torch.randperm(ei.size(1), generator=gen)]
    if kind == "modelnet":
        # 2 clouds of 48 points, 6 nearest neighbours, edges j -> i
        n, edges = 0, []
        pos_all = []
        for _ in range(2):
            pos = torch.rand(48, 3, generator=gen) * 2 - 1
            d = torch.cdist(pos, pos)
            d.fill_diagonal_(float("inf"))
            nbr = d.topk(6, largest=False).indices
            dst = torch.arange(48).view(-1, 1).expand(-1, 6)
            edges.append(torch.stack([nbr.reshape(-1) + n, dst.reshape(-1) + n]))
            pos_all.append(pos)
            n += 48
        return n, torch.cat(edges, dim=1)
    if kind == "hub":
        # 40 nodes; node 7 receives 300 edges, node 21 sends 200 edges, plus random edges
        n = 40
        a = torch.stack([torch.randint(0, n, (300,), generator=gen), torch.full((300,), 7)])
        b = torch.stack([torch.full((200,), 21), torch.randint(0, n, (200,), generator=gen)])
        c = torch.randint(0, n, (2, 120), generator=gen)
        ei = torch.cat([a, b, c], dim=1)
        return n, ei[:, torch.randperm(ei.size(1), generator=gen)]
    raise ValueError(kind)


def layer_case(kind, F_in, Fe, H, seed):
    gen = torch.Generator().manual_seed(seed)
    n, ei = make_graph(gen, kind)
    e = ei.size(1)
    x = torch.rand(n, F_in, generator=gen)
    ea = torch.rand(e, Fe, generator=gen)
    r = torch.randn(n, H, generator=gen)
    w = {
        "cat_linear": linear_weight(gen, H, 2 * F_in + Fe),
        "cat_linear_out": linear_weight(gen, H, H + F_in),
        "sum_fused_linear": linear_weight(gen, 2 * H, F_in),
        "sum_linear_3": linear_weight(gen, H, Fe),
        "sum_linear_out": linear_weight(gen, H, H + F_in),
    }
    rec = {"num_nodes": np.int64(n), "edge_index": ei.numpy(), "x": x.numpy(), "edge_attr": ea.numpy(),
           "r": r.numpy()}
    for k, v in w.items():
        rec["w_" + k] = v.numpy()

    for dt, tag in ((torch.float32, ""), (torch.float64, "64")):
        xx = x.to(dt).requires_grad_(True)
        ww = {k: v.to(dt).requires_grad_(True) for k, v in w.items()}
        ea_d, r_d = ea.to(dt), r.to(dt)
        out, aggr = O.layer_cat(xx, ei, ea_d, ww["cat_linear"], ww["cat_linear_out"], return_aggr=True)
        gx, g1, g2 = torch.autograd.grad((out * r_d).sum(), [xx, ww["cat_linear"], ww["cat_linear_out"]])
        rec.update({f"cat_out{tag}": out.detach().numpy(), f"cat_aggr{tag}": aggr.detach().numpy(),
                    f"cat_gx{tag}": gx.numpy(), f"cat_g_linear{tag}": g1.numpy(),
                    f"cat_g_linear_out{tag}": g2.numpy()})
        out, aggr = O.layer_sum(xx, ei, ea_d, ww["sum_fused_linear"], ww["sum_linear_3"], ww["sum_linear_out"],
                                return_aggr=True)
        gx, g1, g2, g3 = torch.autograd.grad(
            (out * r_d).sum(), [xx, ww["sum_fused_linear"], ww["sum_linear_3"], ww["sum_linear_out"]])
        rec.update({f"sum_out{tag}": out.detach().numpy(), f"sum_aggr{tag}": aggr.detach().numpy(),
                    f"sum_gx{tag}": gx.numpy(), f"sum_g_fused_linear{tag}": g1.numpy(),
                    f"sum_g_linear_3{tag}": g2.numpy(), f"sum_g_linear_out{tag}": g3.numpy()})

    csr = O.csr_build(ei, n)
    for k, v in csr.items():
        rec["csr_" + k] = v.numpy()
    items, split, parts = O.work_items(csr["rowptr"], 64, 256)
    rec["items"], rec["split"], rec["n_parts"] = items.numpy(), split.numpy(), np.int64(parts)
    return rec


def model_case(seed):
    """2-layer MPNN_QM9-shaped model: logits, loss gradient, one Adam step (lr from configs/qm9.yaml:4)."""
    gen = torch.Generator().manual_seed(seed)
    n, ei = make_graph(gen, "qm9")
    sizes = (18, 14, 22, 17, 19)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    F_in, Fe, H, out_c = 11, 4, 64, 19
    x = torch.rand(n, F_in, generator=gen)
    x = x / x.sum(dim=1, keepdim=True)
    ea = torch.nn.functional.one_hot(torch.randint(0, 4, (ei.size(1),), generator=gen), 4).float()
    y = torch.randn(len(sizes), out_c, generator=gen)
    rec = {"num_nodes": np.int64(n), "edge_index": ei.numpy(), "batch": batch.numpy(), "x": x.numpy(),
           "edge_attr": ea.numpy(), "y": y.numpy()}
    for mode in ("cat", "sum"):
        params = {}
        for k, fin in enumerate((F_in, H)):
            if mode == "cat":
                params[f"layers.{k}.linear.weight"] = linear_weight(gen, H, 2 * fin + Fe)
            else:
                params[f"layers.{k}.fused_linear.weight"] = linear_weight(gen, 2 * H, fin)
                params[f"layers.{k}.linear_3.weight"] = linear_weight(gen, H, Fe)
            params[f"layers.{k}.linear_out.weight"] = linear_weight(gen, H, H + fin)
        params["output_layer.weight"] = linear_weight(gen, out_c, H)
        params["output_layer.bias"] = (torch.rand(out_c, generator=gen) * 2 - 1) / (H ** 0.5)
        leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        opt = torch.optim.Adam(list(leaves.values()), lr=0.003)
        logits = O.model_forward(leaves, mode, x, ei, ea, batch)
        loss = torch.nn.functional.mse_loss(logits, y)
        loss.backward()
        for k, v in params.items():
            rec[f"{mode}.param.{k}"] = v.numpy()
            rec[f"{mode}.grad.{k}"] = leaves[k].grad.numpy().copy()
        opt.step()
        for k in params:
            rec[f"{mode}.after_adam.{k}"] = leaves[k].detach().numpy().copy()
        rec[f"{mode}.logits"] = logits.detach().numpy()
        rec[f"{mode}.loss"] = np.float32(loss.item())
        node_logits = O.model_forward(params, mode, x, ei, ea, None)  # MPNN (no pooling) with the same weights
        rec[f"{mode}.node_logits"] = node_logits.numpy()
    return rec


def pool_case(seed):
    gen = torch.Generator().manual_seed(seed)
    batch_sorted = torch.repeat_interleave(torch.arange(6), torch.tensor([3, 1, 0, 7, 2, 5]))
    x = torch.randn(batch_sorted.numel(), 20, generator=gen)
    perm = torch.randperm(batch_sorted.numel(), generator=gen)
    return {"x": x.numpy(), "batch_sorted": batch_sorted.numpy(), "perm": perm.numpy(),
            "out_sorted": O.global_mean_pool(x, batch_sorted, 6).numpy(),
            "out_unsorted": O.global_mean_pool(x, batch_sorted[perm], 6).numpy()}


def egnn_case(seed):
    gen = torch.Generator().manual_seed(seed)
    n, ei = make_graph(gen, "modelnet")
    H, Fe = 16, 1
    a = torch.randn(n, H, generator=gen)
    b = torch.randn(n, H, generator=gen)
    ea = torch.rand(ei.size(1), Fe, generator=gen)
    we = linear_weight(gen, H, Fe)
    pos = torch.rand(n, 3, generator=gen) * 2 - 1
    wp = linear_weight(gen, 1, H).view(-1)
    aggr, pos2 = O.egnn_layer(a, b, ea, we, pos, wp, ei)
    return {"num_nodes": np.int64(n), "edge_index": ei.numpy(), "a": a.numpy(), "b": b.numpy(),
            "edge_attr": ea.numpy(), "w_edge": we.numpy(), "pos": pos.numpy(), "w_pos": wp.numpy(),
            "aggr": aggr.numpy(), "pos_out": pos2.numpy()}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    cases = {
        "layer_tiny": layer_case("tiny", 16, 4, 8, 1),
        "layer_empty": layer_case("empty", 6, 4, 8, 2),
        "layer_qm9": layer_case("qm9", 11, 4, 64, 3),
        "layer_modelnet": layer_case("modelnet", 3, 1, 64, 4),
        "layer_hub": layer_case("hub", 10, 4, 32, 5),
        "layer_odd": layer_case("tiny", 5, 3, 7, 6),
        "model_qm9": model_case(7),
        "pool": pool_case(8),
        "egnn": egnn_case(9),
    }
    total = 0
    for name, rec in cases.items():
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **rec)
        total += os.path.getsize(path)
        print(f"{name}: {os.path.getsize(path)} bytes")
    print("total", total)


if __name__ == "__main__":
    main()
