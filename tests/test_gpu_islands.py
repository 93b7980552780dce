"""The one-launch layers for batches of small graphs (mpnn_island_layer_fwd / _bwd) against the CPU oracle.

The islands of a graph - runs of consecutive node ids that no edge leaves - are found on the device and compared
bit for bit with the oracle's; the layers built on them (MPNNLayerSum, MPNNLayerCat, EGNNLayer; float32 and
bfloat16) are checked like the general path: outputs and every gradient against the float64 oracle, 1e-5 / 2e-2,
and against the general path itself.  The reference has no counterpart of the island kernels (they compute
models/layers.py:91-140 for the DataLoader batches of data/dataset.py:94-96); PARITY UNPINNED as for every layer
test: the oracle restates the reference's formulas.
"""
import pytest
import torch

from conftest import assert_close
from oracle import reference_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = {torch.float32: 1e-5, torch.bfloat16: 2e-2}
KINK = {torch.float32: 0.0, torch.bfloat16: 2e-3}


@pytest.fixture
def islands_on(monkeypatch):
    import mpnn_amd.graph as G
    monkeypatch.setattr(G, "ISLANDS_ENABLED", True)
    G.clear_cache()
    yield
    G.clear_cache()


@pytest.fixture
def islands_off(monkeypatch):
    import mpnn_amd.graph as G
    monkeypatch.setattr(G, "ISLANDS_ENABLED", False)
    G.clear_cache()
    yield
    G.clear_cache()


def molecules(seed, count, max_nodes=29, fe=4, f=11, dense=False):
    """A batch of random small graphs: sizes 1 .. max_nodes, some without edges, self loops and repeated edges
    included; node features, edge attributes, coordinates, graph ids."""
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(1, max_nodes + 1, (count,), generator=g)
    sizes[0] = max_nodes
    if count > 2:
        sizes[1] = 1
    src, dst, off = [], [], 0
    for i, nn in enumerate(sizes.tolist()):
        e = 0 if (i % 7 == 3 or nn == 1 and i % 2) else int(torch.randint(1, (nn * nn if dense else 3 * nn) + 1, (1,), generator=g))
        src.append(torch.randint(0, nn, (e,), generator=g) + off)
        dst.append(torch.randint(0, nn, (e,), generator=g) + off)
        off += nn
    ei = torch.stack([torch.cat(src), torch.cat(dst)]).long()
    ei = ei[:, torch.randperm(ei.size(1), generator=g)]
    n = off
    return dict(x=torch.randn(n, f, generator=g), edge_index=ei, edge_attr=torch.randn(ei.size(1), fe, generator=g),
                pos=torch.randn(n, 3, generator=g), batch=torch.repeat_interleave(torch.arange(count), sizes), n=n)


@pytest.mark.parametrize("case", ["molecules", "molecules-3000", "molecules-12000", "chain", "one-island", "no-edges",
                                  "qm9"])
def test_islands_match_the_oracle(islands_on, case):
    from mpnn_amd import synth
    from mpnn_amd.graph import MPGraph
    if case.startswith("molecules"):
        # 3,000 / 12,000 molecules are 45 / 180 thousand nodes: more than 32 rows per thread of the one-block search,
        # which then walks its rows in memory instead of registers
        d = molecules(1, int(case.split("-")[1]) if "-" in case else 300)
        ei, n = d["edge_index"], d["n"]
        assert ("-" not in case) or n > 32768
    elif case == "chain":            # consecutive nodes joined: one island although every edge is short
        n = 5000
        ei = torch.stack([torch.arange(n - 1), torch.arange(1, n)])
    elif case == "one-island":       # a single long edge spans everything in between
        d = molecules(2, 200)
        n = d["n"]
        ei = torch.cat([d["edge_index"], torch.tensor([[n - 1], [0]])], dim=1)
    elif case == "no-edges":
        n, ei = 2500, torch.zeros((2, 0), dtype=torch.int64)
    else:
        d = synth.qm9("cpu", 0, 1024)
        ei, n = d["edge_index"], d["x"].size(0)
    want = O.islands(ei, n)
    g = MPGraph(ei.to(DEV), n)
    isl = g.islands
    if case == "one-island":
        # one row alone spans the whole graph: the graph build stops searching at the first such row ...
        assert isl is None
        # ... and the search itself, asked to finish, lists the single island
        from mpnn_amd import _lib
        lib = _lib.load()
        ptr = torch.empty(n + 1, dtype=torch.int32, device=DEV)
        meta = torch.empty(4, dtype=torch.int32, device=DEV)
        ws = torch.empty(lib.mpnn_islands_workspace_bytes(n), dtype=torch.uint8, device=DEV)
        _lib.check(lib.mpnn_islands_build(_lib.ptr(g.by_dst.rowptr), _lib.ptr(g.by_dst.col), n, 0, _lib.ptr(ptr),
                                          _lib.ptr(meta), _lib.ptr(ws), ws.numel(), _lib.stream_handle(ptr.device)),
                   "mpnn_islands_build")
        assert meta.tolist()[:2] == [1, n] and meta.tolist()[3] == 0 and ptr[:2].tolist() == want == [0, n]
        return
    assert isl is not None and isl.count == len(want) - 1
    assert isl.ptr[:isl.count + 1].cpu().tolist() == want                      # bit-exact boundaries
    sizes = [b - a for a, b in zip(want[:-1], want[1:])]
    assert isl.max_nodes == max(sizes)
    rowptr = g.by_dst.rowptr.cpu()
    assert isl.max_edges == max(int(rowptr[b] - rowptr[a]) for a, b in zip(want[:-1], want[1:]))


def _layer_case(mode, dtype, f, h, fe, seed, count=200, egnn=False):
    from mpnn_amd import EGNNLayer, MPNNLayerCat, MPNNLayerSum
    d = molecules(seed, count, f=f, fe=fe)
    x, ea, pos, ei = d["x"].to(dtype), d["edge_attr"].to(dtype), d["pos"], d["edge_index"]
    torch.manual_seed(seed + 100)
    layer = (EGNNLayer if egnn else (MPNNLayerSum if mode == "sum" else MPNNLayerCat))(f, fe, h)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(p.to(dtype).float())
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    gen = torch.Generator().manual_seed(seed + 200)
    r, rp = torch.randn(d["n"], h, generator=gen), torch.randn(d["n"], 3, generator=gen)
    xc = x.double().requires_grad_(True)
    pc = pos.double().requires_grad_(True)
    if egnn:
        want, want_pos = O.egnn_full_layer(xc, pc, ei, ea.double(), params["fused_linear.weight"], params["linear_3.weight"],
                                           params["linear_out.weight"], params["coord_linear.weight"])
        ((want * r.double()).sum() + (want_pos * rp.double()).sum()).backward()
    elif mode == "sum":
        want = O.layer_sum(xc, ei, ea.double(), params["fused_linear.weight"], params["linear_3.weight"],
                           params["linear_out.weight"])
        (want * r.double()).sum().backward()
        want_pos = None
    else:
        want = O.layer_cat(xc, ei, ea.double(), params["linear.weight"], params["linear_out.weight"])
        (want * r.double()).sum().backward()
        want_pos = None
    return dict(layer=layer, x=x, ea=ea, pos=pos, ei=ei, r=r, rp=rp, params=params, want=want, want_pos=want_pos,
                xc=xc, pc=pc, n=d["n"])


def _run(c, dtype, egnn):
    layer = c["layer"].to(DEV).to(dtype)
    for p in layer.parameters():
        p.grad = None
    xg = c["x"].to(DEV).clone().requires_grad_(True)
    pg = c["pos"].to(DEV).clone().requires_grad_(True)
    ei, ea = c["ei"].to(DEV), c["ea"].to(DEV)
    if egnn:
        got, got_pos = layer(xg, pg, ei, ea)
        ((got.float() * c["r"].to(DEV)).sum() + (got_pos * c["rp"].to(DEV)).sum()).backward()
    else:
        got, got_pos = layer(xg, ei, ea), None
        (got.float() * c["r"].to(DEV)).sum().backward()
    grads = {k: p.grad.clone() for k, p in layer.named_parameters()}
    return got, got_pos, xg.grad, pg.grad, grads, ei


CASES = [
    pytest.param("sum", torch.float32, 11, 64, 4, False, id="sum-f32-11-64"),
    pytest.param("sum", torch.float32, 64, 64, 4, False, id="sum-f32-64-64"),
    pytest.param("sum", torch.float32, 64, 128, 3, False, id="sum-f32-64-128-fe3"),
    pytest.param("sum", torch.float32, 3, 64, 1, False, id="sum-f32-3-64-fe1"),
    pytest.param("sum", torch.float32, 100, 64, 8, False, id="sum-f32-100-64-fe8"),
    pytest.param("cat", torch.float32, 11, 64, 4, False, id="cat-f32-11-64"),
    pytest.param("sum", torch.bfloat16, 11, 64, 4, False, id="sum-bf16-11-64"),
    pytest.param("sum", torch.bfloat16, 64, 64, 4, False, id="sum-bf16-64-64"),
    pytest.param("sum", torch.float32, 11, 64, 4, True, id="egnn-f32-11-64"),
    pytest.param("sum", torch.float32, 64, 128, 4, True, id="egnn-f32-64-128"),
    pytest.param("sum", torch.bfloat16, 64, 64, 4, True, id="egnn-bf16-64-64"),
    pytest.param("sum", torch.bfloat16, 11, 64, 4, True, id="egnn-bf16-11-64"),
    pytest.param("sum", torch.float32, 128, 128, 4, False, id="sum-f32-128-128"),
    pytest.param("sum", torch.bfloat16, 128, 128, 2, True, id="egnn-bf16-128-128"),
    pytest.param("sum", torch.float32, 32, 64, 4, True, id="egnn-f32-32-64"),
]


@pytest.mark.parametrize("mode,dtype,f,h,fe,egnn", CASES)
def test_island_layer_outputs_and_every_gradient(islands_on, mode, dtype, f, h, fe, egnn):
    from mpnn_amd import islands
    from mpnn_amd.graph import graph_for
    c = _layer_case(mode, dtype, f, h, fe, seed=f + h + fe, egnn=egnn)
    got, got_pos, gx, gpos, grads, ei = _run(c, dtype, egnn)
    graph = graph_for(ei, c["n"])
    assert islands.plan_for(graph, f, h, fe, egnn) is not None, "the island kernels did not take this layer"
    rtol, kink = RTOL[dtype], KINK[dtype]
    assert got.dtype == dtype
    assert_close(got.float(), c["want"], rtol, what="y")
    assert_close(gx.float(), c["xc"].grad, rtol, atol_scale=4.0, what="grad x", kink_frac=kink)
    if egnn:
        assert got_pos.dtype == torch.float32
        assert_close(got_pos, c["want_pos"], rtol, what="pos'")
        assert_close(gpos, c["pc"].grad, rtol, atol_scale=4.0, what="grad pos", kink_frac=kink)
    for k, g in grads.items():
        assert g.shape == c["params"][k].shape
        assert_close(g.float(), c["params"][k].grad, rtol, atol_scale=4.0, what=f"grad {k}", kink_frac=kink)
    # the same bits on a second run: every sum has a fixed order
    got2, pos2, gx2, gpos2, grads2, _ = _run(c, dtype, egnn)
    assert torch.equal(got2, got) and torch.equal(gx2, gx)
    assert all(torch.equal(grads2[k], grads[k]) for k in grads)
    if egnn:
        assert torch.equal(pos2, got_pos) and torch.equal(gpos2, gpos)


@pytest.mark.parametrize("egnn", [False, True], ids=["mpnn", "egnn"])
def test_islands_with_many_edges(islands_on, egnn):
    """Dense little graphs (up to n^2 edges, repeated edges and self loops included): the edge walk stages 32 edges
    at a time, so islands of several hundred edges cross many staging rounds and rows end inside them."""
    from mpnn_amd import EGNNLayer, MPNNLayerSum, islands
    from mpnn_amd.graph import graph_for
    d = molecules(11, 60, max_nodes=24, dense=True)
    assert d["edge_index"].size(1) > 60 * 40
    f, h, fe = 11, 64, 4
    x, ea, pos, ei = d["x"], d["edge_attr"], d["pos"], d["edge_index"]
    torch.manual_seed(1)
    layer = (EGNNLayer if egnn else MPNNLayerSum)(f, fe, h)
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    gen = torch.Generator().manual_seed(2)
    r, rp = torch.randn(d["n"], h, generator=gen), torch.randn(d["n"], 3, generator=gen)
    xc, pc = x.double().requires_grad_(True), pos.double().requires_grad_(True)
    if egnn:
        want, want_pos = O.egnn_full_layer(xc, pc, ei, ea.double(), params["fused_linear.weight"], params["linear_3.weight"],
                                           params["linear_out.weight"], params["coord_linear.weight"])
        ((want * r.double()).sum() + (want_pos * rp.double()).sum()).backward()
    else:
        want = O.layer_sum(xc, ei, ea.double(), params["fused_linear.weight"], params["linear_3.weight"],
                           params["linear_out.weight"])
        (want * r.double()).sum().backward()
    c = dict(layer=layer, x=x, ea=ea, pos=pos, ei=ei, r=r, rp=rp)
    got, got_pos, gx, gpos, grads, ei_dev = _run(c, torch.float32, egnn)
    g = graph_for(ei_dev, d["n"])
    assert g.islands.max_edges > 200 and islands.plan_for(g, f, h, fe, egnn) is not None
    assert_close(got, want, 1e-5, what="y")
    assert_close(gx, xc.grad, 1e-5, atol_scale=4.0, what="grad x")
    if egnn:
        assert_close(got_pos, want_pos, 1e-5, what="pos'")
        assert_close(gpos, pc.grad, 1e-5, atol_scale=4.0, what="grad pos")
    for k, gr in grads.items():
        assert_close(gr, params[k].grad, 1e-5, atol_scale=4.0, what=f"grad {k}")


def test_general_path_on_the_same_batch(islands_off):
    """With the island search off the same batch runs through the general kernels: the two paths agree with the
    oracle independently."""
    from mpnn_amd.graph import graph_for
    c = _layer_case("sum", torch.float32, 11, 64, 4, seed=79)
    got, _, gx, _, grads, ei = _run(c, torch.float32, False)
    assert graph_for(ei, c["n"]).islands is None
    assert_close(got, c["want"], 1e-5, what="y")
    assert_close(gx, c["xc"].grad, 1e-5, atol_scale=4.0, what="grad x")
    for k, g in grads.items():
        assert_close(g, c["params"][k].grad, 1e-5, atol_scale=4.0, what=f"grad {k}")


def test_layers_without_input_gradient_and_without_grad(islands_on):
    """First layers get no gradient for x (the kernels then skip that product); inference keeps nothing."""
    c = _layer_case("sum", torch.float32, 11, 64, 4, seed=5)
    layer = c["layer"].to(DEV)
    x, ei, ea = c["x"].to(DEV), c["ei"].to(DEV), c["ea"].to(DEV)
    got = layer(x, ei, ea)
    (got * c["r"].to(DEV)).sum().backward()
    assert_close(got, c["want"], 1e-5, what="y")
    for k, p in layer.named_parameters():
        assert_close(p.grad, c["params"][k].grad, 1e-5, atol_scale=4.0, what=f"grad {k}")
    with torch.no_grad():
        assert torch.equal(layer(x, ei, ea), got)


def test_islands_too_large_or_too_many_use_the_general_path(islands_on, monkeypatch):
    from mpnn_amd import islands
    from mpnn_amd.graph import MPGraph
    d = molecules(3, 50, max_nodes=40)
    g = MPGraph(d["edge_index"].to(DEV), d["n"])
    assert g.islands is None or g.islands.max_nodes == 40      # (None: a single row already spans 33 nodes or more)
    assert islands.plan_for(g, 11, 64, 4, False) is None
    d = molecules(4, 50)
    g = MPGraph(d["edge_index"].to(DEV), d["n"])
    assert islands.plan_for(g, 11, 64, 4, False) is not None
    assert islands.plan_for(g, 11, 96, 4, False) is None and islands.plan_for(g, 11, 64, 9, False) is None
    monkeypatch.setattr(islands, "COUNT_MAX", 10)
    assert islands.plan_for(g, 11, 64, 4, False) is None


def test_a_deferred_source_view_refuses_an_edge_list_that_changed(islands_on):
    """The two-step build sorts the destination view first and the source view on demand; if edge_index was overwritten
    in between, the second step raises instead of pairing views of two different edge lists."""
    from mpnn_amd.graph import MPGraph
    d = molecules(5, 40)
    ei = d["edge_index"].to(DEV).clone()
    g = MPGraph(ei, d["n"])
    assert g.islands is not None and g._deferred is not None
    ei[0, 0] = ei[0, 1]
    with pytest.raises(RuntimeError, match="modified in place"):
        g.by_src
    g2 = MPGraph(ei, d["n"])
    assert g2.by_src.rowptr.numel() == d["n"] + 1


def test_padded_collated_batches_run_island_by_island(islands_on):
    """A fixed-capacity batch of GraphStore.collate: its graphs are the islands, the padding rows an edgeless tail;
    real rows equal the unpadded batch's bit for bit, parameter gradients too."""
    from mpnn_amd import MPNNLayerSum, islands, synth
    from mpnn_amd.data import GraphStore
    from mpnn_amd.graph import graph_for
    d = synth.qm9(DEV, 0, 256)
    store = GraphStore.from_batched(d)
    ids = list(range(0, 256, 3))
    plain = store.collate(ids)
    padded = store.collate(ids, capacity=store.capacity_for(len(ids)))
    torch.manual_seed(0)
    layer = MPNNLayerSum(d["x"].size(1), d["edge_attr"].size(1), 64).to(DEV)
    outs = []
    for b in (plain, padded):
        g = graph_for(b.edge_index, b.x.size(0))
        assert g.islands is not None and islands.plan_for(g, b.x.size(1), 64, b.edge_attr.size(1), False) is not None
        for p in layer.parameters():
            p.grad = None
        x = b.x.clone().requires_grad_(True)
        y = layer(x, b.edge_index, b.edge_attr)
        y[:plain.num_nodes].square().sum().backward()
        outs.append((y.detach(), x.grad, [p.grad.clone() for p in layer.parameters()]))
    n = plain.num_nodes
    assert torch.equal(outs[0][0], outs[1][0][:n]) and torch.equal(outs[0][1], outs[1][1][:n])
    assert float(outs[1][0][n:].abs().max()) == 0.0
    # the weight-gradient partials are summed block by block: the padded batch has the same blocks plus empty ones
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)
