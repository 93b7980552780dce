"""The whole-layer kernels for batches of point clouds (mpnn_region_layer_fwd / _bwd) against the CPU oracle.

A region is an island of up to 2048 nodes whose rows of x are at most 4 wide - the k-NN clouds of
experiments/modelnet_exp.py:58 with ``data.pos`` as x.  Outputs and every gradient of MPNNLayerSum / MPNNLayerCat
(models/layers.py:91-140) are checked against the float64 oracle, 1e-5 (float32) / 2e-2 (bfloat16), and against the
general path.  The reference has no counterpart of these kernels; PARITY UNPINNED as for every layer test: the
oracle restates the reference's formulas.
"""
import pytest
import torch

from conftest import assert_close
from oracle import reference_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = {torch.float32: 1e-5, torch.bfloat16: 2e-2}
KINK = {torch.float32: 0.0, torch.bfloat16: 2e-3}


@pytest.fixture(autouse=True)
def fresh_graphs():
    import mpnn_amd.graph as G
    G.clear_cache()
    yield
    G.clear_cache()


def clouds(seed, count, max_nodes, f, fe, degree=6):
    """Islands of 33 .. max_nodes nodes (the first the largest, one of a tile and a half); edges anywhere inside an
    island, self loops and repeated edges included, plus a chain through its nodes so that it stays one island."""
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(33, max_nodes + 1, (count,), generator=g)
    sizes[0] = max_nodes
    if count > 2:
        sizes[1] = 48
    src, dst, off = [], [], 0
    for i, nn in enumerate(sizes.tolist()):
        e = int(torch.randint(nn, degree * nn + 1, (1,), generator=g))
        src.append(torch.randint(0, nn, (e,), generator=g) + off)
        dst.append(torch.randint(0, nn, (e,), generator=g) + off)
        src.append(torch.arange(nn - 1) + off)
        dst.append(torch.arange(1, nn) + off)
        off += nn
    ei = torch.stack([torch.cat(src), torch.cat(dst)]).long()
    ei = ei[:, torch.randperm(ei.size(1), generator=g)]
    return dict(x=torch.randn(off, f, generator=g), edge_index=ei, edge_attr=torch.randn(ei.size(1), fe, generator=g),
                n=off, sizes=sizes.tolist())


def _case(d, mode, dtype, h, seed):
    from mpnn_amd import MPNNLayerCat, MPNNLayerSum
    f, fe = d["x"].size(1), d["edge_attr"].size(1)
    x, ea, ei = d["x"].to(dtype), d["edge_attr"].to(dtype), d["edge_index"]
    torch.manual_seed(seed)
    layer = (MPNNLayerSum if mode == "sum" else MPNNLayerCat)(f, fe, h)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(p.to(dtype).float())
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    r = torch.randn(d["n"], h, generator=torch.Generator().manual_seed(seed + 1))
    xc = x.double().requires_grad_(True)
    if mode == "sum":
        want = O.layer_sum(xc, ei, ea.double(), params["fused_linear.weight"], params["linear_3.weight"],
                           params["linear_out.weight"])
    else:
        want = O.layer_cat(xc, ei, ea.double(), params["linear.weight"], params["linear_out.weight"])
    (want * r.double()).sum().backward()
    return dict(layer=layer, x=x, ea=ea, ei=ei, r=r, params=params, want=want, xc=xc, n=d["n"])


def _run(c, dtype, need_x=True):
    layer = c["layer"].to(DEV).to(dtype)
    for p in layer.parameters():
        p.grad = None
    xg = c["x"].to(DEV).clone().requires_grad_(need_x)
    ei, ea = c["ei"].to(DEV), c["ea"].to(DEV)
    got = layer(xg, ei, ea)
    (got.float() * c["r"].to(DEV)).sum().backward()
    return got, xg.grad, {k: p.grad.clone() for k, p in layer.named_parameters()}, ei


def _check(c, dtype, got, gx, grads):
    rtol, kink = RTOL[dtype], KINK[dtype]
    assert got.dtype == dtype
    assert_close(got.float(), c["want"], rtol, what="y")
    if gx is not None:
        assert_close(gx.float(), c["xc"].grad, rtol, atol_scale=4.0, what="grad x", kink_frac=kink)
    for k, g in grads.items():
        assert g.shape == c["params"][k].shape
        assert_close(g.float(), c["params"][k].grad, rtol, atol_scale=4.0, what=f"grad {k}", kink_frac=kink)


CASES = [
    pytest.param("sum", torch.float32, 3, 1, 300, id="sum-f32-F3-Fe1"),
    pytest.param("sum", torch.float32, 4, 4, 521, id="sum-f32-F4-Fe4"),
    pytest.param("sum", torch.float32, 1, 8, 129, id="sum-f32-F1-Fe8"),
    pytest.param("cat", torch.float32, 3, 1, 200, id="cat-f32-F3-Fe1"),
    pytest.param("sum", torch.float32, 2, 3, 2048, id="sum-f32-F2-Fe3-2048"),
    pytest.param("sum", torch.bfloat16, 3, 1, 300, id="sum-bf16-F3-Fe1"),
    pytest.param("cat", torch.bfloat16, 4, 2, 260, id="cat-bf16-F4-Fe2"),
]


@pytest.mark.parametrize("mode,dtype,f,fe,max_nodes", CASES)
def test_region_layer_outputs_and_every_gradient(mode, dtype, f, fe, max_nodes):
    from mpnn_amd import islands
    from mpnn_amd.graph import graph_for
    d = clouds(f + fe + max_nodes, 3 if max_nodes > 1000 else 12, max_nodes, f, fe)
    c = _case(d, mode, dtype, 64, seed=7 + f)
    got, gx, grads, ei = _run(c, dtype)
    graph = graph_for(ei, c["n"])
    assert graph.islands is not None and graph.islands.count == len(d["sizes"])
    plan = islands.plan_for(graph, f, 64, fe, False)
    assert plan is not None and plan.region_nodes == max_nodes, "the region kernels did not take this layer"
    _check(c, dtype, got, gx, grads)
    # the same bits on a second run: every sum has a fixed order
    got2, gx2, grads2, _ = _run(c, dtype)
    assert torch.equal(got2, got) and torch.equal(gx2, gx)
    assert all(torch.equal(grads2[k], grads[k]) for k in grads)
    # without the gradient of x (the first layer of a model) the backward is one pass: the source half of the node
    # weight's gradient is then summed edge by edge in destination order instead of row by row in source order -
    # the same value to rounding, checked against the oracle again; everything else keeps its bits
    got3, gx3, grads3, _ = _run(c, dtype, need_x=False)
    assert gx3 is None and torch.equal(got3, got)
    _check(c, dtype, got3, None, grads3)
    for k in grads:
        if "fused_linear" not in k and k != "linear.weight":
            assert torch.equal(grads3[k], grads[k]), k
    got4, _, grads4, _ = _run(c, dtype, need_x=False)
    assert all(torch.equal(grads4[k], grads3[k]) for k in grads)


EGNN_CASES = [
    pytest.param(torch.float32, 3, 1, 300, id="egnn-f32-F3-Fe1"),
    pytest.param(torch.float32, 4, 4, 523, id="egnn-f32-F4-Fe4"),
    pytest.param(torch.float32, 2, 3, 1024, id="egnn-f32-F2-Fe3-1024"),
    pytest.param(torch.bfloat16, 3, 1, 300, id="egnn-bf16-F3-Fe1"),
]


@pytest.mark.parametrize("dtype,f,fe,max_nodes", EGNN_CASES)
def test_region_egnn_layer_outputs_and_every_gradient(dtype, f, fe, max_nodes):
    """The EGNN layer (messages as above plus pos'[n] = pos[n] + sum (pos[n] - pos[src]) (w_pos . m)) on the same
    kernels: y, pos' and the gradients of x, pos and all four weights against the float64 oracle."""
    from mpnn_amd import EGNNLayer, islands
    from mpnn_amd.graph import graph_for
    d = clouds(3 * f + fe + max_nodes, 3 if max_nodes > 1000 else 10, max_nodes, f, fe)
    n, h = d["n"], 64
    gen = torch.Generator().manual_seed(11 + f)
    pos = torch.randn(n, 3, generator=gen)
    x, ea, ei = d["x"].to(dtype), d["edge_attr"].to(dtype), d["edge_index"]
    torch.manual_seed(5 + f)
    layer = EGNNLayer(f, fe, h)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(p.to(dtype).float())
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    r, rp = torch.randn(n, h, generator=gen), torch.randn(n, 3, generator=gen)
    xc, pc = x.double().requires_grad_(True), pos.double().requires_grad_(True)
    want, want_pos = O.egnn_full_layer(xc, pc, ei, ea.double(), params["fused_linear.weight"], params["linear_3.weight"],
                                       params["linear_out.weight"], params["coord_linear.weight"])
    ((want * r.double()).sum() + (want_pos * rp.double()).sum()).backward()

    def run(inputs_need_grad=True):
        lay = layer.to(DEV).to(dtype)
        for p in lay.parameters():
            p.grad = None
        xg = x.to(DEV).clone().requires_grad_(inputs_need_grad)
        pg = pos.to(DEV).clone().requires_grad_(inputs_need_grad)
        y, pos_out = lay(xg, pg, ei.to(DEV), ea.to(DEV))
        ((y.float() * r.to(DEV)).sum() + (pos_out * rp.to(DEV)).sum()).backward()
        return y, pos_out, xg.grad, pg.grad, {k: p.grad.clone() for k, p in lay.named_parameters()}

    y, pos_out, gx, gpos, grads = run()
    graph = graph_for(ei.to(DEV), n)
    plan = islands.plan_for(graph, f, h, fe, True)
    assert plan is not None and plan.region_nodes == max_nodes and plan.egnn, "the region kernels did not take this layer"
    rtol, kink = RTOL[dtype], KINK[dtype]
    assert y.dtype == dtype and pos_out.dtype == torch.float32
    assert_close(y.float(), want, rtol, what="y")
    assert_close(pos_out, want_pos, rtol, what="pos'")
    assert_close(gx.float(), xc.grad, rtol, atol_scale=4.0, what="grad x", kink_frac=kink)
    assert_close(gpos, pc.grad, rtol, atol_scale=4.0, what="grad pos", kink_frac=kink)
    for k, g in grads.items():
        assert g.shape == params[k].shape
        assert_close(g.float(), params[k].grad, rtol, atol_scale=4.0, what=f"grad {k}", kink_frac=kink)
    y2, pos2, gx2, gpos2, grads2 = run()
    assert torch.equal(y2, y) and torch.equal(pos2, pos_out) and torch.equal(gx2, gx) and torch.equal(gpos2, gpos)
    assert all(torch.equal(grads2[k], grads[k]) for k in grads)
    # a model's first layer: neither x nor pos wants a gradient, and the backward is one pass shorter
    y3, pos3, gx3, gpos3, grads3 = run(inputs_need_grad=False)
    assert gx3 is None and gpos3 is None and torch.equal(y3, y) and torch.equal(pos3, pos_out)
    for k, g in grads3.items():
        assert_close(g.float(), params[k].grad, rtol, atol_scale=4.0, what=f"first layer: grad {k}", kink_frac=kink)


def test_knn_clouds_egnn_layer_agrees_with_the_general_path(monkeypatch):
    """BASELINE's point-cloud EGNN shape, smaller: k-NN clouds, x = pos, one edge feature."""
    from mpnn_amd import EGNNLayer, islands, synth
    from mpnn_amd.graph import clear_cache, graph_for
    d = synth.modelnet(DEV, 5, 3)
    x, ei, ea = d["x"].clone(), d["edge_index"], d["edge_attr"]
    pos = d["pos"].float().clone()
    n = x.size(0)
    torch.manual_seed(0)
    layer = EGNNLayer(3, 1, 64).to(DEV)
    r, rp = torch.randn(n, 64, device=DEV), torch.randn(n, 3, device=DEV)

    def run():
        for p in layer.parameters():
            p.grad = None
        xg, pg = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        y, po = layer(xg, pg, ei, ea)
        ((y * r).sum() + (po * rp).sum()).backward()
        return [y, po, xg.grad, pg.grad] + [p.grad.clone() for p in layer.parameters()]

    got = run()
    plan = islands.plan_for(graph_for(ei, n), 3, 64, 1, True)
    assert plan is not None and plan.region_nodes == 1024 and plan.egnn
    monkeypatch.setattr(islands, "REGIONS_ENABLED", False)
    clear_cache()
    assert islands.plan_for(graph_for(ei, n), 3, 64, 1, True) is None
    want = run()
    for i, (g, w) in enumerate(zip(got, want)):
        # (two float32 evaluations: a relu decision at zero may differ and moves the two end rows' gradients)
        assert_close(g, w.double(), 1e-5, atol_scale=1.0 if i < 2 else 4.0, what=f"output {i}", kink_frac=2e-4 if i in (2, 3) else 1e-5)


def test_knn_clouds_take_the_region_kernels_and_agree_with_the_general_path(monkeypatch):
    """The benchmark's own shape, smaller: k-NN clouds built by the library, x = pos, one edge feature."""
    from mpnn_amd import MPNNLayerSum, islands, synth
    from mpnn_amd.graph import clear_cache, graph_for
    d = synth.modelnet(DEV, 3, 4)
    x, ei, ea = d["x"], d["edge_index"], d["edge_attr"]
    n = x.size(0)
    torch.manual_seed(0)
    layer = MPNNLayerSum(3, 1, 64).to(DEV)
    r = torch.randn(n, 64, device=DEV)

    def run():
        for p in layer.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        y = layer(xg, ei, ea)
        (y * r).sum().backward()
        return y, xg.grad, {k: p.grad.clone() for k, p in layer.named_parameters()}

    y, gx, grads = run()
    plan = islands.plan_for(graph_for(ei, n), 3, 64, 1, False)
    assert plan is not None and plan.region_nodes == 1024
    monkeypatch.setattr(islands, "REGIONS_ENABLED", False)
    clear_cache()
    assert islands.plan_for(graph_for(ei, n), 3, 64, 1, False) is None
    y0, gx0, grads0 = run()
    assert_close(y, y0.double(), 1e-5, what="y")
    assert_close(gx, gx0.double(), 1e-5, atol_scale=4.0, what="grad x")
    for k in grads:
        assert_close(grads[k], grads0[k].double(), 1e-5, atol_scale=4.0, what=f"grad {k}")
    # and against the oracle
    params = {k: v.detach().cpu().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    xc = x.cpu().double().requires_grad_(True)
    want = O.layer_sum(xc, ei.cpu(), ea.cpu().double(), params["fused_linear.weight"], params["linear_3.weight"],
                       params["linear_out.weight"])
    (want * r.cpu().double()).sum().backward()
    assert_close(y.cpu(), want, 1e-5, what="y vs oracle")
    assert_close(gx.cpu(), xc.grad, 1e-5, atol_scale=4.0, what="grad x vs oracle")
    for k in grads:
        assert_close(grads[k].cpu(), params[k].grad, 1e-5, atol_scale=4.0, what=f"grad {k} vs oracle")


def test_backward_without_the_kept_sums_recomputes_them():
    """The backward takes the sums over incoming edges from the forward; handed none, it walks the edges once more:
    the same gradients, bit for bit."""
    from mpnn_amd import islands
    from mpnn_amd.graph import graph_for
    d = clouds(5, 6, 400, 3, 1)
    g = torch.Generator().manual_seed(1)
    n, h = d["n"], 64
    x, ea, ei = d["x"].to(DEV), d["edge_attr"].to(DEV), d["edge_index"].to(DEV)
    wn, we, wo = (torch.randn(s, generator=g).mul(0.3).to(DEV) for s in ((2 * h, 3), (h, 1), (h, 3 + h)))
    gy = torch.randn(n, h, generator=g).to(DEV)
    graph = graph_for(ei, n)
    plan = islands.plan_for(graph, 3, h, 1, False)
    assert plan is not None and plan.region_nodes == 400
    y, _, saved = islands.forward(plan, graph, x, ea, wn, we, wo)
    assert saved[5] is not None and saved[5].dtype == torch.float32
    kept = [t.clone() for t in islands.backward(plan, graph, gy, y, saved, (True, True, True, True))[:4]]
    again = islands.backward(plan, graph, gy, y, saved[:5] + (None, saved[6]), (True, True, True, True))[:4]
    for a, b in zip(kept, again):
        assert torch.equal(a, b)


def test_region_entry_points_reject_what_they_do_not_take():
    import ctypes
    from mpnn_amd import _lib
    lib = _lib.load()
    assert lib.mpnn_region_layer_fits(3, 64, 1, 1024, 0) == 1
    assert lib.mpnn_region_layer_fits(3, 64, 1, 2048, 0) == 1
    assert lib.mpnn_region_layer_fits(3, 64, 1, 1024, 1) == 1
    assert lib.mpnn_region_layer_fits(3, 64, 1, 2048, 1) == 0       # with coordinates: three tables in the LDS
    assert lib.mpnn_region_layer_fits(5, 64, 1, 1024, 0) == 0       # x too wide
    assert lib.mpnn_region_layer_fits(3, 128, 1, 1024, 0) == 0      # H
    assert lib.mpnn_region_layer_fits(3, 64, 1, 2049, 0) == 0       # island too large
    assert lib.mpnn_region_layer_partial_floats(4, 1024, 3, 64, 1, 0) > 0
    assert lib.mpnn_region_layer_partial_floats(4, 1024, 3, 64, 1, 1) > lib.mpnn_region_layer_partial_floats(4, 1024, 3, 64, 1, 0)
    assert lib.mpnn_region_layer_partial_floats(4, 4096, 3, 64, 1, 0) == -1
    d = _lib.IslandLayerDesc()
    d.F, d.H, d.Fe, d.dtype, d.rows, d.n_islands, d.num_nodes = 3, 64, 1, 0, 32, 1, 64
    assert lib.mpnn_region_layer_fwd(ctypes.byref(d), 64, None, None) != 0     # null pointers
    assert b"null pointer" in lib.mpnn_last_error()
    d.F = 7
    assert lib.mpnn_region_layer_fwd(ctypes.byref(d), 64, None, None) != 0
    assert b"1 <= F <= 4" in lib.mpnn_last_error()
