"""Mini-batch assembly: host-side slot table and the oracle's collation (CPU), the device assembler against the
oracle and against the general graph builder (GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import reference_oracle as ro
from conftest import assert_close


def tiny_dataset():
    # three graphs: 3 nodes / 4 edges, 1 node / 0 edges, 2 nodes / 2 edges; dataset-global ids
    node_ptr = np.array([0, 3, 4, 6])
    edge_ptr = np.array([0, 4, 4, 6])
    ei = torch.tensor([[0, 1, 2, 1, 4, 5], [1, 0, 1, 2, 5, 4]])
    x = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    ea = torch.arange(6, dtype=torch.float32).reshape(6, 1) + 100
    y = torch.tensor([[10.0], [20.0], [30.0]])
    return x, ei, ea, node_ptr, edge_ptr, y


def test_oracle_collate_known_answer():
    x, ei, ea, node_ptr, edge_ptr, y = tiny_dataset()
    b = ro.collate(x, ei, ea, node_ptr, edge_ptr, [2, 0, 1], y=y)
    assert b["batch"].tolist() == [0, 0, 1, 1, 1, 2]
    assert b["edge_index"].tolist() == [[0, 1, 2, 3, 4, 3], [1, 0, 3, 2, 3, 4]]
    assert b["x"][:, 0].tolist() == [8.0, 10.0, 0.0, 2.0, 4.0, 6.0]
    assert b["edge_attr"].flatten().tolist() == [104.0, 105.0, 100.0, 101.0, 102.0, 103.0]
    assert b["y"].flatten().tolist() == [30.0, 10.0, 20.0]
    # the same graph may appear twice in a batch
    b2 = ro.collate(x, ei, ea, node_ptr, edge_ptr, [2, 2], y=y)
    assert b2["edge_index"].tolist() == [[0, 1, 2, 3], [1, 0, 3, 2]]


def test_slot_table_matches_the_oracle_offsets():
    from mpnn_amd.data import slot_table
    _, _, _, node_ptr, edge_ptr, _ = tiny_dataset()
    t = slot_table(node_ptr, edge_ptr, [2, 0, 1])
    assert t.dtype == np.int32 and t.shape == (5, 4)
    assert t[0].tolist() == [0, 2, 5, 6] and t[1].tolist() == [0, 2, 6, 6]
    assert t[2, :3].tolist() == [0 - 4, 2 - 0, 5 - 3] and t[3, :3].tolist() == [0 - 4, 2 - 0, 6 - 4]
    assert t[4, :3].tolist() == [2, 0, 1]
    with pytest.raises(IndexError):
        slot_table(node_ptr, edge_ptr, [3])
    with pytest.raises(ValueError):
        slot_table(node_ptr, edge_ptr, [])


def _store_and_cpu(kind, n_graphs):
    from mpnn_amd import synth
    from mpnn_amd.data import GraphStore
    dev = torch.device("cuda:0")
    d = synth.qm9(dev, 3, n_graphs) if kind == "qm9" else synth.modelnet(dev, 3, n_graphs, 128, 8)
    store = GraphStore.from_batched(d)
    cpu = dict(x=store.x.cpu(), edge_index=torch.stack([store.src, store.dst]).cpu(), edge_attr=store.edge_attr.cpu(),
               y=store.y.cpu(), pos=store.pos.cpu() if store.pos is not None else None)
    return store, cpu


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n_graphs,ids", [
    ("qm9", 40, [5, 0, 39, 17, 17, 3]), ("qm9", 40, [7]), ("qm9", 300, list(range(299, -1, -3))),
    ("modelnet", 6, [4, 1, 5])])
def test_device_collate_is_bit_exact(kind, n_graphs, ids):
    from mpnn_amd.graph import CHUNK, MAX_PARTS, MPGraph, clear_cache, graph_for
    store, cpu = _store_and_cpu(kind, n_graphs)
    assert store.fused
    got = store.collate(ids)
    want = ro.collate(cpu["x"], cpu["edge_index"], cpu["edge_attr"], store.node_ptr, store.edge_ptr, ids,
                      y=cpu["y"], pos=cpu["pos"])
    for k in ("x", "edge_index", "edge_attr", "batch", "y", "pos"):
        w = want[k]
        if w is None:
            continue
        assert torch.equal(getattr(got, k).cpu(), w), k
    n = want["num_nodes"]
    assert got.num_nodes == n and got.num_graphs == len(ids)
    # index structure: against the oracle's stable sorts and against the general builder on the same edge list
    g = graph_for(got.edge_index, n)            # the registered one
    ref = ro.csr_build(want["edge_index"], n)
    assert torch.equal(g.by_dst.rowptr.cpu(), ref["rowptr"]) and torch.equal(g.by_dst.col.cpu(), ref["col"])
    assert torch.equal(g.by_dst.eid.cpu(), ref["eid"])
    assert torch.equal(g.by_src.rowptr.cpu(), ref["rowptr_t"]) and torch.equal(g.by_src.col.cpu(), ref["col_t"])
    assert torch.equal(g.by_src.eid.cpu(), ref["eid_t"])
    items, split, parts = ro.work_items(ref["rowptr"], CHUNK, MAX_PARTS)
    assert parts == 0 and split.numel() == 0
    as_set = lambda t: sorted(map(tuple, t.tolist()))
    assert as_set(g.by_dst.items[:g.by_dst.n_items].cpu()) == as_set(items)
    general = MPGraph(got.edge_index, n)
    assert torch.equal(general.tpos, g.tpos)
    ea_d, ea_s = general.sorted_edge_attr(got.edge_attr, got.edge_attr.dtype)
    gd, gs = g.sorted_edge_attr(got.edge_attr, got.edge_attr.dtype)
    assert torch.equal(ea_d, gd) and torch.equal(ea_s, gs)
    # pooling segments seeded by the assembler: same result as deriving them from the batch vector
    from mpnn_amd import ops
    feats = torch.randn(n, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    seeded = ops.global_mean_pool(feats, got.batch)
    ops._POOL_CACHE.clear()
    derived = ops.global_mean_pool(feats, got.batch)
    assert torch.equal(seeded, derived)
    assert_close(seeded, ro.global_mean_pool(feats.cpu(), want["batch"]), 1e-6, what="pool")
    clear_cache()


@pytest.mark.gpu
def test_model_on_assembled_batch_equals_general_path():
    from mpnn_amd import MPNN_QM9
    from mpnn_amd.graph import clear_cache
    store, _ = _store_and_cpu("qm9", 64)
    torch.manual_seed(0)
    model = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").cuda()
    outs = []
    for registered in (True, False):
        b = store.collate(list(range(10, 42)))
        if not registered:
            clear_cache()                      # forces the general builder on the same edge list
        model.zero_grad()
        out = model(b.x, b.edge_index, b.batch, b.edge_attr)
        (out * out).sum().backward()
        outs.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        if "linear_3" in k:
            # the edge-weight gradient is a sum over work items in item order, and the two builders order the
            # items differently (row order here, descending degree there): same terms, different rounding
            assert_close(outs[0][1][k], outs[1][1][k], 1e-5, what=k)
        else:
            assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    clear_cache()


@pytest.mark.gpu
def test_loader_covers_every_graph_once_per_epoch():
    from mpnn_amd.data import GraphLoader
    store, _ = _store_and_cpu("qm9", 70)
    loader = GraphLoader(store, batch_size=32, shuffle=True, seed=1)
    assert len(loader) == 3
    seen, sizes = [], []
    for data in loader:
        data = data.to("cuda")
        sizes.append(data.num_graphs)
        assert data.y.shape == (data.num_graphs, 19) and int(data.batch.max()) == data.num_graphs - 1
        seen.append(data.y[:, 0].cpu())
    assert sizes == [32, 32, 6]
    assert torch.equal(torch.sort(torch.cat(seen)).values, torch.sort(store.y[:, 0].cpu()).values)


@pytest.mark.gpu
def test_collate_into_the_same_buffers_feeds_a_captured_step():
    """Fixed-size clouds: a new batch is written over the old one and a captured training step, replayed,
    computes what an eager step on a freshly allocated batch computes."""
    from mpnn_amd import MPNN_ModelNet
    from mpnn_amd.optim import FlatAdam
    from mpnn_amd.train import CapturedStep
    from mpnn_amd.graph import clear_cache
    store, _ = _store_and_cpu("modelnet", 12)
    models, opts = [], []
    for _ in range(2):
        torch.manual_seed(7)
        m = MPNN_ModelNet(3, 1, 32, 10, num_layers=2, mode="sum").cuda()
        models.append(m)
        opts.append(FlatAdam(m.parameters(), lr=1e-2))

    def step_on(m, o, b):
        o.zero_grad()
        loss = F.cross_entropy(m(b.pos, b.edge_index, b.batch, b.edge_attr), b.y)
        loss.backward()
        o.step()
        return loss

    batches = [[0, 1, 2, 3], [7, 5, 11, 2], [4, 4, 9, 10]]
    fixed = store.collate(batches[0])
    cap = CapturedStep(lambda: step_on(models[1], opts[1], fixed), warmup=2)
    for _ in range(2):
        step_on(models[0], opts[0], store.collate(batches[0]))
    for ids in batches[1:] + batches[:1]:
        le = step_on(models[0], opts[0], store.collate(ids))
        store.collate(ids, out=fixed)
        lc = cap.replay()
        assert torch.equal(le, lc)
    for pa, pb in zip(models[0].parameters(), models[1].parameters()):
        assert torch.equal(pa, pb)
    with pytest.raises(ValueError):
        store.collate([0, 1], out=fixed)
    clear_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("ids", [[5, 0, 39, 17, 17, 3], [7, 8, 9, 10, 11, 12]])
def test_padded_collate_is_the_exact_batch_plus_padding(ids):
    """capacity=(N, E): the first real rows / edge entries are bit for bit those of the exact batch, the rest is the
    documented padding, and refilling the same buffers with other graphs gives what a fresh padded batch holds."""
    from mpnn_amd.graph import clear_cache, graph_for
    store, _ = _store_and_cpu("qm9", 40)
    cap = store.capacity_for(len(ids))
    exact = store.collate(ids)
    n, e, b = exact.num_nodes, exact.edge_index.size(1), len(ids)
    assert n < cap[0] and e <= cap[1]
    ge = graph_for(exact.edge_index, n)
    keep = {k: getattr(exact, k).clone() for k in ("x", "edge_index", "edge_attr", "batch", "y", "pos")}
    keep.update(rowptr=ge.by_dst.rowptr.clone(), col=ge.by_dst.col.clone(), rowptr_t=ge.by_src.rowptr.clone(),
                col_t=ge.by_src.col.clone(), tpos=ge.tpos.clone())
    first = store.collate([1, 2, 3, 4, 6, 30], capacity=cap)       # other graphs first, then refilled in place
    pad = store.collate(ids, out=first)
    assert pad is first and pad.capacity == cap and pad.num_nodes == cap[0] and pad.edge_index.shape == (2, cap[1])
    assert torch.equal(pad.x[:n], keep["x"]) and float(pad.x[n:].abs().sum()) == 0.0
    assert torch.equal(pad.pos[:n], keep["pos"]) and float(pad.pos[n:].abs().sum()) == 0.0
    assert torch.equal(pad.batch[:n], keep["batch"]) and bool((pad.batch[n:] == b).all())
    assert torch.equal(pad.y, keep["y"])
    assert torch.equal(pad.edge_index[:, :e], keep["edge_index"]) and bool((pad.edge_index[:, e:] == cap[0] - 1).all())
    assert torch.equal(pad.edge_attr[:e], keep["edge_attr"]) and float(pad.edge_attr[e:].abs().sum()) == 0.0
    g = graph_for(pad.edge_index, cap[0])
    assert g.by_dst.n_items == cap[0] and g.by_dst.n_parts == 0
    for view, rp, col in ((g.by_dst, "rowptr", "col"), (g.by_src, "rowptr_t", "col_t")):
        assert torch.equal(view.rowptr[:n + 1], keep[rp]) and bool((view.rowptr[n:] == e).all())
        assert torch.equal(view.col[:e], keep[col])
        items = view.items[:cap[0]].cpu()
        assert torch.equal(items[:, 0], torch.arange(cap[0], dtype=torch.int32))
        assert torch.equal(items[:, 1], view.rowptr[:-1].cpu()) and torch.equal(items[:, 2], view.rowptr[1:].cpu())
    assert torch.equal(g.tpos[:e], keep["tpos"])
    seg = pad._parts["pool_rowptr"].cpu().tolist()
    assert seg[:b + 1] == pad._parts["slots"][0].cpu().tolist() and seg[b] == n and seg[b + 1] == cap[0]
    with pytest.raises(ValueError):
        store.collate(ids, capacity=(n, e))                        # no room for the padding node
    clear_cache()


@pytest.mark.gpu
def test_model_on_a_padded_batch_equals_the_exact_batch():
    from mpnn_amd import MPNN_QM9
    from mpnn_amd.graph import clear_cache
    store, _ = _store_and_cpu("qm9", 64)
    ids = list(range(10, 42))
    torch.manual_seed(0)
    model = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").cuda()
    res = []
    for cap in (None, store.capacity_for(len(ids))):
        b = store.collate(ids, capacity=cap)
        model.zero_grad()
        out = model(b.x, b.edge_index, b.batch, b.edge_attr)
        assert out.size(0) == len(ids) + (cap is not None)
        out = out[:len(ids)]
        (out * out).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])                       # real rows do not see the padding
    for k in res[0][1]:                                            # sums over rows: other grouping, same terms + zeros
        assert_close(res[1][1][k], res[0][1][k].double().cpu(), 1e-5, what=k)
    clear_cache()


@pytest.mark.gpu
def test_pool_of_a_padded_batch_leaves_the_padding_graph_out():
    """global_mean_pool on a padded batch: the real graphs' rows are those of the exact batch bit for bit, the spare
    graph's row is zero whatever its (thousands of) padding rows hold, and those rows take a zero gradient."""
    from mpnn_amd import global_mean_pool
    from mpnn_amd.graph import clear_cache
    store, _ = _store_and_cpu("qm9", 64)
    ids = list(range(5, 37))
    exact = store.collate(ids)
    n = exact.num_nodes
    gen = torch.Generator().manual_seed(4)
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(store.capacity_for(len(ids))[0], 64, generator=gen).cuda().to(dtype)
        r = torch.randn(len(ids) + 1, 64, generator=gen).cuda().to(dtype)
        xe = x[:n].clone().requires_grad_(True)
        want = global_mean_pool(xe, exact.batch)
        want.backward(r[:-1])
        pad = store.collate(ids, capacity=store.capacity_for(len(ids)))
        xp = x.clone().requires_grad_(True)
        got = global_mean_pool(xp, pad.batch)
        got.backward(r)
        assert got.shape == (len(ids) + 1, 64)
        assert torch.equal(got[:-1], want) and float(got[-1].abs().sum()) == 0.0
        assert torch.equal(xp.grad[:n], xe.grad) and float(xp.grad[n:].abs().sum()) == 0.0
    clear_cache()


@pytest.mark.gpu
def test_captured_batch_step_trains_like_the_eager_loop():
    """Molecule batches of varying size through one recorded step: same losses and weights as the eager loop
    (to float32 rounding: the padded weight-gradient sums group their rows differently)."""
    from mpnn_amd import MPNN_QM9
    from mpnn_amd.optim import FlatAdam
    from mpnn_amd.train import CapturedBatchStep
    from mpnn_amd.graph import clear_cache
    store, _ = _store_and_cpu("qm9", 50)
    models, opts = [], []
    for _ in range(2):
        torch.manual_seed(7)
        m = MPNN_QM9(11, 4, 32, 19, num_layers=2, mode="sum").cuda()
        models.append(m)
        opts.append(FlatAdam(m.parameters(), lr=1e-2))
    loss_fn = F.mse_loss
    fast = CapturedBatchStep(models[1], opts[1], loss_fn, store, batch_size=8)
    batches = [list(range(0, 8)), list(range(20, 28)), [49, 3, 17, 5, 33, 8, 1, 40], [9, 9], list(range(30, 38))]
    for ids in batches:
        b = store.collate(ids)
        opts[0].zero_grad()
        le = loss_fn(models[0](b.x, b.edge_index, b.batch, b.edge_attr)[0], b.y[0])
        le.backward()
        opts[0].step()
        lc = fast.train(ids)
        assert_close(lc.reshape(1), le.detach().reshape(1).double().cpu(), 1e-4, what=f"loss of batch {ids[:2]}..")
    assert opts[0].steps_taken == opts[1].steps_taken == len(batches)
    for pa, pb in zip(models[0].parameters(), models[1].parameters()):
        assert_close(pb, pa.detach().double().cpu(), 1e-3, what="weights")
    clear_cache()


@pytest.mark.gpu
def test_captured_batch_step_with_the_egnn_model():
    """The coordinate-threading model through the recorded step (call= says how it takes a batch): same losses and
    weights as the eager loop on the exact batches."""
    from mpnn_amd import EGNN
    from mpnn_amd.optim import FlatAdam
    from mpnn_amd.train import CapturedBatchStep
    from mpnn_amd.graph import clear_cache
    store, _ = _store_and_cpu("qm9", 40)
    assert store.pos is not None
    call = lambda m, b: m(b.x, b.pos, b.edge_index, b.batch, b.edge_attr)
    models, opts = [], []
    for _ in range(2):
        torch.manual_seed(3)
        m = EGNN(11, 4, 64, 19, num_layers=2).cuda()
        models.append(m)
        opts.append(FlatAdam(m.parameters(), lr=1e-2))
    loss_of = lambda out, b: F.mse_loss(out[:b.num_graphs], b.y)
    fast = CapturedBatchStep(models[1], opts[1], F.mse_loss, store, batch_size=8, loss_of=loss_of, call=call)
    for ids in (list(range(0, 8)), list(range(16, 24)), [39, 3, 17, 5, 33, 8, 1, 30]):
        b = store.collate(ids)
        opts[0].zero_grad()
        le = loss_of(call(models[0], b), b)
        le.backward()
        opts[0].step()
        lc = fast.train(ids)
        assert_close(lc.reshape(1), le.detach().reshape(1).double().cpu(), 1e-4, what=f"loss of batch {ids[:2]}..")
    for pa, pb in zip(models[0].parameters(), models[1].parameters()):
        assert_close(pb, pa.detach().double().cpu(), 1e-3, what="weights")
    clear_cache()


def _hub_store(dev):
    """Graphs of identical size whose node 0 receives an edge from every other node of the graph and a second one
    from most of them (in-degree 70 + 40 > 64, so rows are split and the store cannot register its own structure)."""
    from mpnn_amd.data import GraphStore
    gen = torch.Generator().manual_seed(11)
    n_graphs, per = 12, 71
    srcs, dsts = [], []
    for g in range(n_graphs):
        base = g * per
        others = torch.arange(1, per) + base
        extra = others[torch.randperm(per - 1, generator=gen)[:40]]
        back = others[torch.randperm(per - 1, generator=gen)[:30]]
        srcs += [others, extra, torch.full((30,), base)]
        dsts += [torch.full((per - 1,), base), torch.full((40,), base), back]
    ei = torch.stack([torch.cat(srcs), torch.cat(dsts)])
    e_per = (per - 1) + 40 + 30
    x = torch.rand(n_graphs * per, 11, generator=gen)
    ea = torch.rand(ei.size(1), 4, generator=gen)
    y = torch.randn(n_graphs, 19, generator=gen)
    node_ptr = torch.arange(n_graphs + 1) * per
    edge_ptr = torch.arange(n_graphs + 1) * e_per
    return GraphStore(x.to(dev), ei.to(dev), ea.to(dev), node_ptr, edge_ptr, y=y.to(dev))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["hub-store (general graph builder)", "fused store after clear_cache", "fused store, many batches between"])
def test_refilled_batch_uses_the_structure_of_its_new_contents(kind):
    """collate(ids, out=batch) rewrites a batch through raw device pointers (no version bump): the layers must
    never run on an index structure or sorted edge attributes derived from the earlier contents."""
    from mpnn_amd import MPNN_QM9
    from mpnn_amd.graph import clear_cache
    dev = torch.device("cuda:0")
    clear_cache()
    if kind.startswith("hub"):
        store = _hub_store(dev)
        assert not store.fused
        first, second = [0, 1, 2, 3], [7, 5, 11, 9]
    else:
        store, _ = _store_and_cpu("modelnet", 48)        # equal-size clouds: any ids refill the same buffers
        assert store.fused
        first, second = [0, 1, 2, 3], [40, 7, 33, 21]
    torch.manual_seed(0)
    model = MPNN_QM9(store.x.size(1), store.edge_attr.size(1), 64, 19, num_layers=2, mode="sum").to(dev)

    def run(b):
        model.zero_grad()
        out = model(b.x, b.edge_index, b.batch, b.edge_attr)
        (out * out).sum().backward()
        return out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}

    batch = store.collate(first)
    run(batch)                                            # builds / finds the structure of the first contents
    if kind == "fused store after clear_cache":
        clear_cache()
    if kind == "fused store, many batches between":
        for k in range(12):                               # more than the graph cache holds
            run(store.collate([k, k + 1, k + 2, k + 3]))
    store.collate(second, out=batch)
    got = run(batch)
    if kind == "fused store after clear_cache":           # ... and once more, now from whatever that run left behind
        store.collate(first, out=batch)
        run(batch)
        store.collate(second, out=batch)
        got = run(batch)
    clear_cache()
    want = run(store.collate(second))
    assert torch.equal(got[0], want[0])
    for k in want[1]:
        assert_close(got[1][k], want[1][k], 1e-6, what=k)
    clear_cache()
