"""Randomised sweep of the device-side mini-batch assembly: random datasets of small graphs, random id lists (with
repeats), exact and padded mode, every output and both sorted views bit for bit against the oracle.
Usage: python tests/fuzz_collate.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd import synth
from mpnn_amd.data import GraphStore
from mpnn_amd.graph import clear_cache, graph_for
from oracle import reference_oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
R = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))

for c in range(cases):
    kind = "qm9" if R(0, 3) else "modelnet"
    n_graphs = R(1, 300) if kind == "qm9" else R(1, 6)
    d = synth.qm9(dev, seed * 100 + c, n_graphs) if kind == "qm9" else synth.modelnet(dev, seed * 100 + c, n_graphs, R(8, 100), R(1, 7))
    store = GraphStore.from_batched(d)
    cpu = dict(x=store.x.cpu(), edge_index=torch.stack([store.src, store.dst]).cpu(), edge_attr=store.edge_attr.cpu(),
               y=store.y.cpu(), pos=store.pos.cpu() if store.pos is not None else None)
    for rep in range(3):
        b = R(1, min(64, 4 * n_graphs))
        ids = [R(0, n_graphs - 1) for _ in range(b)] if rep == 0 else torch.randperm(n_graphs, generator=g)[:min(b, n_graphs)].tolist()
        b = len(ids)
        want = O.collate(cpu["x"], cpu["edge_index"], cpu["edge_attr"], store.node_ptr, store.edge_ptr, ids, y=cpu["y"], pos=cpu["pos"])
        n, e = want["num_nodes"], want["edge_index"].size(1)
        ref = O.csr_build(want["edge_index"], n)
        tag = f"case {c}.{rep} {kind} graphs={n_graphs} batch={b}"
        exact = store.collate(ids)
        for k in ("x", "edge_index", "edge_attr", "batch", "y", "pos"):
            if want[k] is not None:
                assert torch.equal(getattr(exact, k).cpu(), want[k]), f"{tag}: {k}"
        gr = graph_for(exact.edge_index, n)
        for view, names in ((gr.by_dst, ("rowptr", "col", "eid")), (gr.by_src, ("rowptr_t", "col_t", "eid_t"))):
            for attr, nm in zip(("rowptr", "col", "eid"), names):
                assert torch.equal(getattr(view, attr).cpu(), ref[nm]), f"{tag}: {nm}"
        inv = torch.empty(e, dtype=torch.int64); inv[ref["eid"].long()] = torch.arange(e)
        assert torch.equal(gr.tpos.cpu().long(), inv[ref["eid_t"].long()]), f"{tag}: tpos"
        # padded mode into fresh buffers and refilled in place; capacity_for() promises room for distinct graphs
        if len(set(ids)) != b:
            continue
        cap = store.capacity_for(b)
        pad = store.collate(ids, capacity=cap)
        if rep:
            other = torch.randperm(n_graphs, generator=g)[:b].tolist()
            pad = store.collate(other, capacity=cap)
            pad = store.collate(ids, out=pad)
        assert torch.equal(pad.x[:n].cpu(), want["x"]) and float(pad.x[n:].abs().sum()) == 0.0, f"{tag}: padded x"
        assert torch.equal(pad.batch[:n].cpu(), want["batch"]) and bool((pad.batch[n:] == b).all()), f"{tag}: padded batch"
        assert torch.equal(pad.edge_index[:, :e].cpu(), want["edge_index"]) and bool((pad.edge_index[:, e:] == cap[0] - 1).all()), f"{tag}: padded edge_index"
        assert torch.equal(pad.edge_attr[:e].cpu(), want["edge_attr"]), f"{tag}: padded edge_attr"
        gp = graph_for(pad.edge_index, cap[0])
        for view, rp, col in ((gp.by_dst, "rowptr", "col"), (gp.by_src, "rowptr_t", "col_t")):
            assert torch.equal(view.rowptr[:n + 1].cpu(), ref[rp]) and bool((view.rowptr[n:] == e).all()), f"{tag}: padded {rp}"
            assert torch.equal(view.col[:e].cpu(), ref[col]), f"{tag}: padded {col}"
        assert torch.equal(gp.tpos[:e].cpu().long(), inv[ref["eid_t"].long()]), f"{tag}: padded tpos"
    clear_cache()
    if c % 10 == 9:
        print(f"{c + 1} datasets ok", flush=True)
print("fuzz ok")
