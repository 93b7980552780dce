"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Tolerances (fp32): rel 1e-5 with an absolute floor of 1e-5 * max|expected| (outputs pass through a
bias-free relu, so many entries are ~0).  bf16: rel 2e-2 with the same kind of floor.
Integer structures (CSR, transpose, degrees, work items) must match bit for bit.
"""
import pytest
import torch

from conftest import assert_close, load_golden
from mpnn_amd.graph import MAX_PARTS, default_chunk
from oracle import reference_oracle as O

pytestmark = pytest.mark.gpu

RTOL32 = 1e-5
RTOL16 = 2e-2
LAYER_CASES = ["layer_tiny", "layer_empty", "layer_qm9", "layer_modelnet", "layer_hub", "layer_odd"]


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _graph(g, dev):
    from mpnn_amd.graph import MPGraph
    return MPGraph(g["edge_index"].to(dev), int(g["num_nodes"]))


# --------------------------------------------------------------------------- integer structures
@pytest.mark.parametrize("case", LAYER_CASES)
def test_csr_bit_exact_golden(case, dev):
    g = load_golden(case)
    graph = _graph(g, dev)
    for name, got in (("rowptr", graph.by_dst.rowptr), ("col", graph.by_dst.col), ("eid", graph.by_dst.eid),
                      ("rowptr_t", graph.by_src.rowptr), ("col_t", graph.by_src.col), ("eid_t", graph.by_src.eid),
                      ("deg_in", graph.by_dst.degree()), ("deg_out", graph.by_src.degree())):
        assert torch.equal(got.cpu(), g["csr_" + name]), f"{case}: {name} differs"
    items = graph.by_dst.items[:graph.by_dst.n_items].cpu()
    split = graph.by_dst.split[:graph.by_dst.n_split].cpu()
    assert torch.equal(items, g["items"].reshape(-1, 4)), f"{case}: work items differ"
    assert torch.equal(split, g["split"].reshape(-1, 4)), f"{case}: split rows differ"
    assert graph.by_dst.n_parts == int(g["n_parts"])


def test_csr_bit_exact_random_large(dev):
    gen = torch.Generator().manual_seed(11)
    n, e = 5000, 200_000
    # skewed destinations so that some rows are split into several pieces
    dst = (torch.rand(e, generator=gen) ** 4 * n).long().clamp_(max=n - 1)
    src = torch.randint(0, n, (e,), generator=gen)
    ei = torch.stack([src, dst])
    want = O.csr_build(ei, n)
    graph = _graph({"edge_index": ei, "num_nodes": n}, dev)
    assert torch.equal(graph.by_dst.rowptr.cpu(), want["rowptr"])
    assert torch.equal(graph.by_dst.col.cpu(), want["col"])
    assert torch.equal(graph.by_dst.eid.cpu(), want["eid"])
    assert torch.equal(graph.by_src.rowptr.cpu(), want["rowptr_t"])
    assert torch.equal(graph.by_src.col.cpu(), want["col_t"])
    assert torch.equal(graph.by_src.eid.cpu(), want["eid_t"])
    items, split, parts = O.work_items(want["rowptr"], default_chunk(int(want["rowptr"][-1])), MAX_PARTS)
    assert torch.equal(graph.by_dst.items[:graph.by_dst.n_items].cpu(), items)
    assert torch.equal(graph.by_dst.split[:graph.by_dst.n_split].cpu(), split)
    assert graph.by_dst.n_parts == parts and parts > 0


def test_work_items_of_a_large_edge_list_use_longer_pieces(dev):
    """From 2^23 edges on the split pieces are at least 128 edges long (default_chunk); the items stay bit-exact."""
    gen = torch.Generator().manual_seed(12)
    n, e = 20_000, 9_000_000
    dst = (torch.rand(e, generator=gen) ** 3 * n).long().clamp_(max=n - 1)
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), dst])
    assert default_chunk(e) == 128
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0)
    items, split, parts = O.work_items(rowptr.int(), 128, MAX_PARTS)
    graph = _graph({"edge_index": ei, "num_nodes": n}, dev)
    assert torch.equal(graph.by_dst.rowptr.cpu().long(), rowptr)
    assert torch.equal(graph.by_dst.items[:graph.by_dst.n_items].cpu(), items)
    assert torch.equal(graph.by_dst.split[:graph.by_dst.n_split].cpu(), split)
    assert graph.by_dst.n_parts == parts and parts > 0


def test_rows_in_work_item_order_give_identical_results(dev):
    """row_order="items" only moves whole rows inside the sorted views: the work items become one ascending run
    over the edge arrays, and the fused operator returns bit-identical outputs and gradients."""
    from mpnn_amd import synth
    from mpnn_amd.graph import MPGraph
    from mpnn_amd.ops import aggregate_fused
    n, e, H = 20000, 600000, 64
    d = synth.powerlaw(dev, 3, n, e, 8, 4)
    ei = d["edge_index"]
    g_nat = MPGraph(ei, n, chunk=64)
    g_it = MPGraph(ei, n, chunk=64, row_order="items")
    for a, b in ((g_nat.by_dst, g_it.by_dst), (g_nat.by_src, g_it.by_src)):
        assert b.n_items == a.n_items and b.n_split == a.n_split
        assert torch.equal(a.degree(), b.degree())
        items = b.items[:b.n_items].cpu()
        assert int(items[0, 1]) == 0 and int(items[-1, 2]) == e and torch.equal(items[1:, 1], items[:-1, 2])
        # same rows, same pieces
        listing = lambda it: sorted(zip(it[:, 0].tolist(), (it[:, 2] - it[:, 1]).tolist(), it[:, 3].tolist()))
        assert listing(items) == listing(a.items[:a.n_items].cpu())
        # every row holds the same (neighbour, edge id) sequence
        order = b.order.cpu()
        for j in (0, 1, n // 2, n - 1):
            row = int(order[j])
            got = slice(int(b.rowptr[j]), int(b.rowptr[j + 1]))
            want = slice(int(a.rowptr[row]), int(a.rowptr[row + 1]))
            assert torch.equal(b.col[got], a.col[want]) and torch.equal(b.eid[got], a.eid[want])
    assert g_it.by_dst.n_split > 0      # the hubs of this graph are cut into pieces
    gen = torch.Generator(device=dev).manual_seed(1)
    ab = torch.randn(n, 2 * H, generator=gen, device=dev)
    we = torch.randn(H, 4, generator=gen, device=dev)
    r = torch.randn(n, H, generator=gen, device=dev)
    res = []
    for g in (g_nat, g_it):
        ab_, we_ = ab.clone().requires_grad_(True), we.clone().requires_grad_(True)
        out = aggregate_fused(ab_, d["edge_attr"], we_, g)
        out.backward(r)
        res.append((out.detach(), ab_.grad, we_.grad))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    # the EGNN edge passes walk the same structures
    from mpnn_amd import egnn_aggregate
    pos = torch.rand(n, 3, generator=gen, device=dev) * 2 - 1
    wp = torch.randn(H, generator=gen, device=dev) / H
    rp = torch.randn(n, 3, generator=gen, device=dev)
    res = []
    for g in (g_nat, g_it):
        leaves = [t.clone().requires_grad_(True) for t in (ab[:, :H].contiguous(), ab[:, H:].contiguous(), we, pos, wp)]
        aggr, pos_out = egnn_aggregate(leaves[0], leaves[1], d["edge_attr"], leaves[2], leaves[3], leaves[4], g)
        ((aggr * r).sum() + (pos_out * rp).sum()).backward()
        res.append([aggr.detach(), pos_out.detach()] + [t.grad for t in leaves])
    for x, y in zip(*res):
        assert torch.equal(x, y)
    bad = ei.clone()
    bad[1, 5] = n
    with pytest.raises(IndexError):
        MPGraph(bad, n, row_order="items")


def test_out_of_range_edge_raises(dev):
    from mpnn_amd.graph import MPGraph
    ei = torch.tensor([[0, 1, 9], [1, 2, 0]], dtype=torch.int64, device=dev)
    with pytest.raises(IndexError):
        MPGraph(ei, 3)


# --------------------------------------------------------------------------- aggregate operator
def _aggregate_case(dev, n, e, H, Fe, seed, skew=False, dtype=torch.float32):
    from mpnn_amd.graph import MPGraph
    from mpnn_amd.ops import aggregate_fused
    gen = torch.Generator().manual_seed(seed)
    if skew:
        dst = (torch.rand(e, generator=gen) ** 6 * n).long().clamp_(max=n - 1)
    else:
        dst = torch.randint(0, n, (e,), generator=gen)
    src = torch.randint(0, n, (e,), generator=gen)
    ei = torch.stack([src, dst])
    ab = torch.randn(n, 2 * H, generator=gen)
    ea = torch.rand(e, Fe, generator=gen)
    we = torch.randn(H, Fe, generator=gen) * 0.5
    r = torch.randn(n, H, generator=gen)
    if dtype != torch.float32:
        ab, ea, we = (t.to(dtype).float() for t in (ab, ea, we))  # oracle sees the rounded inputs

    ab_c = ab.clone().requires_grad_(True)
    we_c = we.clone().requires_grad_(True)
    want = O.aggregate_sum_form(ab_c[:, :H], ab_c[:, H:], ea, we_c, ei)
    (want * r).sum().backward()

    graph = MPGraph(ei.to(dev), n)
    ab_g = ab.to(dev, dtype).requires_grad_(True)
    we_g = we.to(dev, dtype).requires_grad_(True)
    got = aggregate_fused(ab_g, ea.to(dev, dtype), we_g, graph)
    (got.float() * r.to(dev)).sum().backward()
    return (got, ab_g.grad, we_g.grad), (want, ab_c.grad, we_c.grad), graph


@pytest.mark.parametrize("n,e,H,Fe,skew", [
    (300, 3000, 64, 4, False),     # QM9/Flickr layer shape
    (300, 6000, 128, 4, True),     # hidden 128, skewed: split rows in both CSRs
    (257, 2000, 64, 1, False),     # ModelNet edge width
    (100, 900, 8, 4, False),       # 2 lanes per row
    (100, 900, 20, 3, False),      # partial last lane group, generic edge width
    (64, 500, 7, 2, False),        # scalar path (H % 4 != 0)
    (50, 400, 320, 4, False),      # two column tiles
])
def test_aggregate_fwd_bwd_fp32(n, e, H, Fe, skew, dev):
    got, want, graph = _aggregate_case(dev, n, e, H, Fe, seed=H + Fe, skew=skew)
    if skew:
        assert graph.by_dst.n_parts > 0
    assert_close(got[0], want[0], RTOL32, what="aggregate out")
    assert_close(got[1], want[1], RTOL32, what="grad AB")
    assert_close(got[2], want[2], RTOL32, what="grad We")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_counts_of_rows_longer_than_a_byte_can_count(dtype, dev):
    """The forward stores the per-column count of positive messages as one byte; rows of more than 255 edges (split
    rows, and unsplit rows when the pieces are longer than 255 edges) keep exact counts in a side table that
    mpnn_count_fixup applies after the GEMM epilogue multiplied by the bytes.  Messages are made positive almost
    everywhere so that the counts really pass 255."""
    from mpnn_amd import MPNNLayerSum
    from mpnn_amd.graph import MPGraph, clear_cache, register_graph
    from mpnn_amd.ops import aggregate_fused
    gen = torch.Generator().manual_seed(77)
    n, H, Fe = 400, 64, 4
    degs = [2000, 700, 511, 400, 300, 257, 256, 255, 254] + [7] * 200       # chunk 512: 2 split rows, 5 long unsplit rows
    dst = torch.cat([torch.full((d,), i) for i, d in enumerate(degs)])
    dst = dst[torch.randperm(dst.numel(), generator=gen)]
    e = dst.numel()
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), dst])
    graph = MPGraph(ei.to(dev), n, chunk=512)
    assert graph.by_dst.n_split == 2 and graph.by_dst.n_big == 5
    rtol = RTOL32 if dtype == torch.float32 else RTOL16
    rnd = (lambda *s: torch.randn(*s, generator=gen).to(dtype).float())

    # the bare operator
    ab = rnd(n, 2 * H) * 0.3 + 1.0                      # positive pre-activations: counts close to the degrees
    ea, we, r = torch.rand(e, Fe, generator=gen).to(dtype).float(), rnd(H, Fe) * 0.1, rnd(n, H)
    # float64 oracle: the float32 CPU sum of 2,000 equal-signed terms is itself 2e-5 off, the GPU's count * g is exact
    ab_c, we_c = ab.double().requires_grad_(True), we.double().requires_grad_(True)
    want = O.aggregate_sum_form(ab_c[:, :H], ab_c[:, H:], ea.double(), we_c, ei)
    (want * r.double()).sum().backward()
    ab_g, we_g = ab.to(dev, dtype).requires_grad_(True), we.to(dev, dtype).requires_grad_(True)
    got = aggregate_fused(ab_g, ea.to(dev, dtype), we_g, graph)
    (got.float() * r.to(dev)).sum().backward()
    assert_close(got.float(), want, rtol, what="out")
    assert_close(ab_g.grad.float(), ab_c.grad, rtol, what="grad AB (counts up to 2000)")
    assert float(ab_c.grad[:7, :H].abs().max()) > 0

    # the whole layer: the byte counts ride on the epilogue of the GEMM, the fix-up follows it
    clear_cache()
    ei_d = ei.to(dev)
    register_graph(ei_d, n, graph)
    torch.manual_seed(5)
    layer = MPNNLayerSum(16, Fe, H)
    with torch.no_grad():
        layer.fused_linear.weight.mul_(0.05).add_(0.08)       # mostly positive projections
        for p in layer.parameters():
            p.copy_(p.to(dtype).float())
    x = (torch.rand(n, 16, generator=gen) + 0.5).to(dtype).float()
    params = [p.detach().double().requires_grad_(True) for p in layer.parameters()]
    xc = x.double().requires_grad_(True)
    want = O.layer_sum(xc, ei, ea.double(), *params)
    (want * r.double()).sum().backward()
    layer = layer.to(dev, dtype)

    def run():
        for p in layer.parameters():
            p.grad = None
        xg = x.to(dev, dtype).requires_grad_(True)
        out = layer(xg, ei_d, ea.to(dev, dtype))
        (out.float() * r.to(dev)).sum().backward()
        return [out.detach().float(), xg.grad.float()] + [p.grad.float().clone() for p in layer.parameters()]

    got = run()
    assert_close(got[0], want, rtol, what="layer out")
    if dtype == torch.float32:
        for g, w in zip(got[1:], [xc.grad] + [p.grad for p in params]):
            assert_close(g, w, rtol, atol_scale=2.0, what="layer gradient")
    else:
        # aggregates of 2,000 positive messages next to rows of 7: in bfloat16 storage the weight gradients of this
        # construction lose digits to cancellation whatever computes them, so the yardstick is the same layer over
        # a structure in which every long row is a split row (pieces of 64 edges: no byte count is ever used for it)
        clear_cache()
        plain = MPGraph(ei_d, n, chunk=64)
        assert plain.by_dst.n_big == 0 and plain.by_dst.n_split >= 7
        register_graph(ei_d, n, plain)
        for g, w in zip(got[1:], run()[1:]):
            assert_close(g, w, rtol, atol_scale=2.0, what="layer gradient against the all-split structure")
    clear_cache()


# H <= 128: packed 16-bit mask words; H = 256: 32 lanes per row, one 32-bit word per column group; H = 8: one lane
@pytest.mark.parametrize("n,e,H,Fe", [(300, 3000, 64, 4), (200, 4000, 128, 4), (100, 700, 24, 2), (150, 2500, 256, 4),
                                      (120, 1500, 8, 1)])
def test_aggregate_fwd_bwd_bf16(n, e, H, Fe, dev):
    got, want, _ = _aggregate_case(dev, n, e, H, Fe, seed=3, dtype=torch.bfloat16)
    assert_close(got[0].float(), want[0], RTOL16, what="aggregate out (bf16)")
    assert_close(got[1].float(), want[1], RTOL16, what="grad AB (bf16)")
    assert_close(got[2].float(), want[2], RTOL16, what="grad We (bf16)")


def test_aggregate_deterministic(dev):
    a, _, _ = _aggregate_case(dev, 400, 20000, 64, 4, seed=5, skew=True)
    b, _, _ = _aggregate_case(dev, 400, 20000, 64, 4, seed=5, skew=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y), "two runs on the same input differ bitwise"


def test_aggregate_separate_tables(dev):
    """A [Nd,H] and B [Ns,H] given separately, Nd != Ns (a rank-local piece of a partitioned graph)."""
    from mpnn_amd.graph import MPGraph
    from mpnn_amd.ops import aggregate
    gen = torch.Generator().manual_seed(21)
    nd, ns, e, H, Fe = 70, 190, 1500, 64, 4
    ei = torch.stack([torch.randint(0, ns, (e,), generator=gen), torch.randint(0, nd, (e,), generator=gen)])
    a = torch.randn(nd, H, generator=gen)
    b = torch.randn(ns, H, generator=gen)
    ea = torch.rand(e, Fe, generator=gen)
    we = torch.randn(H, Fe, generator=gen)
    r = torch.randn(nd, H, generator=gen)
    ac, bc, wc = (t.clone().requires_grad_(True) for t in (a, b, we))
    msg = torch.relu((ac[ei[1]] + bc[ei[0]]) + ea @ wc.t())
    want = torch.zeros(nd, H).index_add_(0, ei[1], msg)
    (want * r).sum().backward()
    graph = MPGraph(ei.to(dev), nd, ns)
    ag, bg, wg = (t.to(dev).requires_grad_(True) for t in (a, b, we))
    got = aggregate(ag, bg, ea.to(dev), wg, graph)
    (got * r.to(dev)).sum().backward()
    assert_close(got, want, RTOL32, what="out")
    assert_close(ag.grad, ac.grad, RTOL32, what="gA")
    assert_close(bg.grad, bc.grad, RTOL32, what="gB")
    assert_close(wg.grad, wc.grad, RTOL32, what="gWe")


# --------------------------------------------------------------------------- layers against golden vectors
@pytest.mark.parametrize("case", LAYER_CASES)
@pytest.mark.parametrize("mode", ["cat", "sum"])
def test_layer_golden(case, mode, dev):
    from mpnn_amd import MPNNLayerCat, MPNNLayerSum
    g = load_golden(case)
    n = int(g["num_nodes"])
    F_in, Fe = g["x"].size(1), g["edge_attr"].size(1)
    H = g["r"].size(1)
    if mode == "cat":
        layer = MPNNLayerCat(F_in, Fe, H)
        names = {"linear": "w_cat_linear", "linear_out": "w_cat_linear_out"}
    else:
        layer = MPNNLayerSum(F_in, Fe, H)
        names = {"fused_linear": "w_sum_fused_linear", "linear_3": "w_sum_linear_3", "linear_out": "w_sum_linear_out"}
    with torch.no_grad():
        for attr, key in names.items():
            getattr(layer, attr).weight.copy_(g[key])
    layer = layer.to(dev)
    x = g["x"].to(dev).requires_grad_(True)
    out = layer(x, g["edge_index"].to(dev), g["edge_attr"].to(dev))
    assert out.shape == (n, H)
    (out * g["r"].to(dev)).sum().backward()
    assert_close(out, g[f"{mode}_out"], RTOL32, what=f"{case}/{mode} out")
    assert_close(x.grad, g[f"{mode}_gx"], RTOL32, what=f"{case}/{mode} grad x")
    for attr in names:
        assert_close(getattr(layer, attr).weight.grad, g[f"{mode}_g_{attr}"], RTOL32,
                     what=f"{case}/{mode} grad {attr}")
    # float64 recomputation of the oracle bounds how much of the budget is rounding in the oracle itself
    assert_close(out, g[f"{mode}_out64"], 4 * RTOL32, what=f"{case}/{mode} out vs float64")


# --------------------------------------------------------------------------- pooling and models
@pytest.mark.parametrize("mode", ["sum", "cat"])
@pytest.mark.parametrize("n,e,f,h,fe", [(300, 2000, 1, 16, 3), (1, 4, 5, 8, 2), (2, 0, 1, 4, 1)])
def test_layer_degenerate_widths(mode, n, e, f, h, fe, dev):
    """One input feature, a single node, no edges: matrices with a single row or column keep whatever stride torch
    left them, which must not reach the kernels as a leading dimension."""
    from mpnn_amd import MPNNLayerCat, MPNNLayerSum
    gen = torch.Generator().manual_seed(n + e)
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)])
    x, ea, r = torch.randn(n, f, generator=gen), torch.rand(e, fe, generator=gen), torch.randn(n, h, generator=gen)
    torch.manual_seed(3)
    layer = (MPNNLayerSum if mode == "sum" else MPNNLayerCat)(f, fe, h)
    ws = [p.detach().clone().requires_grad_(True) for p in layer.parameters()]
    xw = x.clone().requires_grad_(True)
    want = (O.layer_sum if mode == "sum" else O.layer_cat)(xw, ei, ea, *ws)
    (want * r).sum().backward()
    layer = layer.to(dev)
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg, ei.to(dev), ea.to(dev))
    (out * r.to(dev)).sum().backward()
    assert_close(out, want.detach(), RTOL32, what="out")
    assert_close(xg.grad, xw.grad, RTOL32, what="gx")
    for p, w in zip(layer.parameters(), ws):
        assert_close(p.grad, w.grad, RTOL32, what="gW")


@pytest.mark.parametrize("mode", ["sum", "cat"])
def test_strided_views_of_the_inputs(mode, dev):
    """x as a column slice or every second row of a larger array, edge_index as the transposed view of an [E, 2]
    array, edge_attr as a column slice: same bits as with contiguous inputs."""
    from mpnn_amd import MPNNLayerCat, MPNNLayerSum
    gen = torch.Generator().manual_seed(0)
    n, e, f, h, fe = 500, 4000, 12, 32, 3
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)]).to(dev)
    x, ea, r = torch.randn(n, f, generator=gen).to(dev), torch.rand(e, fe, generator=gen).to(dev), torch.randn(n, h, generator=gen).to(dev)
    torch.manual_seed(1)
    layer = (MPNNLayerSum if mode == "sum" else MPNNLayerCat)(f, fe, h).to(dev)

    def run(x_, ei_, ea_):
        for p in layer.parameters():
            p.grad = None
        out = layer(x_, ei_, ea_)
        (out * r).sum().backward()
        return [out.detach()] + [p.grad.clone() for p in layer.parameters()]

    base = run(x, ei, ea)
    wide = torch.randn(n, f + 5, device=dev)
    wide[:, 2:2 + f] = x
    tall = torch.empty(2 * n, f, device=dev)
    tall[::2] = x
    ea_wide = torch.zeros(e, fe + 2, device=dev)
    ea_wide[:, 1:1 + fe] = ea
    for args in ((wide[:, 2:2 + f], ei, ea), (tall[::2], ei, ea), (x, ei.t().contiguous().t(), ea),
                 (x, ei, ea_wide[:, 1:1 + fe])):
        for got, want in zip(run(*args), base):
            assert torch.equal(got, want)


def test_global_mean_pool_golden(dev):
    from mpnn_amd import global_mean_pool
    g = load_golden("pool")
    x = g["x"].to(dev).requires_grad_(True)
    out = global_mean_pool(x, g["batch_sorted"].to(dev), 6)
    assert_close(out, g["out_sorted"], RTOL32, what="pool sorted")
    out.sum().backward()
    cnt = torch.bincount(g["batch_sorted"], minlength=6).clamp(min=1).float()
    assert_close(x.grad, (1.0 / cnt)[g["batch_sorted"]].view(-1, 1).expand(-1, 20), RTOL32, what="pool grad")
    out2 = global_mean_pool(g["x"].to(dev), g["batch_sorted"][g["perm"]].to(dev), 6)
    assert_close(out2, g["out_unsorted"], RTOL32, what="pool unsorted")


@pytest.mark.parametrize("mode", ["cat", "sum"])
def test_model_qm9_golden_with_adam_step(mode, dev):
    from mpnn_amd import MPNN, MPNN_QM9
    g = load_golden("model_qm9")
    model = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode=mode)
    state = {k[len(mode) + 7:]: v for k, v in g.items() if k.startswith(f"{mode}.param.")}
    model.load_state_dict(state, strict=True)   # same state_dict keys as the reference
    model = model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.003)
    x, ei, batch, ea = (g[k].to(dev) for k in ("x", "edge_index", "batch", "edge_attr"))
    logits = model(x, ei, batch, ea)           # reference argument order: batch before edge_attr
    loss = torch.nn.functional.mse_loss(logits, g["y"].to(dev))
    loss.backward()
    assert_close(logits, g[f"{mode}.logits"], RTOL32, what="logits")
    for k, p in model.named_parameters():
        assert_close(p.grad, g[f"{mode}.grad.{k}"], RTOL32, atol_scale=2.0, what=f"grad {k}")
    opt.step()
    for k, p in model.named_parameters():
        assert_close(p, g[f"{mode}.after_adam.{k}"], 1e-4, what=f"after Adam {k}")

    node_model = MPNN(11, 64, 19, num_layers=2, mode=mode)
    node_model.load_state_dict(state, strict=True)
    node_logits = node_model.to(dev)(x, ei, ea)
    assert_close(node_logits, g[f"{mode}.node_logits"], RTOL32, what="node logits")


def test_graph_cache_reuses_and_invalidates(dev):
    from mpnn_amd.graph import graph_for, clear_cache
    clear_cache()
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]], dtype=torch.int64, device=dev)
    g1 = graph_for(ei, 3)
    assert graph_for(ei, 3) is g1
    ei[0, 0] = 2                      # in-place edit bumps the version counter
    g2 = graph_for(ei, 3)
    assert g2 is not g1
    assert g2.by_dst.col.cpu().tolist() == [2, 2, 1]


def test_cpu_tensor_is_rejected_loudly():
    from mpnn_amd import MPNNLayerSum
    from mpnn_amd._lib import MpnnError
    layer = MPNNLayerSum(4, 4, 8)
    with pytest.raises(MpnnError):
        layer(torch.zeros(3, 4), torch.zeros((2, 2), dtype=torch.int64), torch.zeros(2, 4))


# --------------------------------------------------------------------------- EGNN coordinate update
# No reference counterpart: parity is against the build's own oracle of the formula
# pos_i += sum_j (pos_i - pos_j) * phi_x(m_ij)  (oracle/reference_oracle.py::egnn_layer) -- parity unpinned.
def test_egnn_golden(dev):
    from mpnn_amd import egnn_aggregate
    g = load_golden("egnn")
    graph = _graph(g, dev)
    aggr, pos = egnn_aggregate(g["a"].to(dev), g["b"].to(dev), g["edge_attr"].to(dev), g["w_edge"].to(dev),
                               g["pos"].to(dev), g["w_pos"].to(dev), graph)
    assert_close(aggr, g["aggr"], RTOL32, what="egnn aggr")
    assert_close(pos, g["pos_out"], RTOL32, what="egnn pos")


@pytest.mark.parametrize("n,e,H,Fe,skew,dtype", [
    (300, 3000, 64, 4, False, torch.float32),
    (200, 6000, 128, 1, True, torch.float32),
    (150, 1500, 8, 3, False, torch.float32),
    (300, 3000, 64, 4, False, torch.bfloat16),
])
def test_egnn_fwd_bwd(n, e, H, Fe, skew, dtype, dev):
    from mpnn_amd import egnn_aggregate
    from mpnn_amd.graph import MPGraph
    gen = torch.Generator().manual_seed(31 + H)
    dst = ((torch.rand(e, generator=gen) ** 6 * n).long().clamp_(max=n - 1) if skew
           else torch.randint(0, n, (e,), generator=gen))
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), dst])
    a, b = torch.randn(n, H, generator=gen), torch.randn(n, H, generator=gen)
    ea = torch.rand(e, Fe, generator=gen)
    we = torch.randn(H, Fe, generator=gen) * 0.5
    wp = torch.randn(H, generator=gen) / H
    pos = torch.rand(n, 3, generator=gen) * 2 - 1
    r, rp = torch.randn(n, H, generator=gen), torch.randn(n, 3, generator=gen)
    if dtype != torch.float32:
        a, b, ea, we, wp = (t.to(dtype).float() for t in (a, b, ea, we, wp))
    leaves = [t.clone().requires_grad_(True) for t in (a, b, we, pos, wp)]
    aggr_w, pos_w = O.egnn_layer(leaves[0], leaves[1], ea, leaves[2], leaves[3], leaves[4], ei)
    ((aggr_w * r).sum() + (pos_w * rp).sum()).backward()

    graph = MPGraph(ei.to(dev), n)
    if skew:
        assert graph.by_dst.n_parts > 0
    gl = [t.to(dev, dtype if i != 3 else torch.float32).requires_grad_(True) for i, t in enumerate((a, b, we, pos, wp))]
    aggr_g, pos_g = egnn_aggregate(gl[0], gl[1], ea.to(dev, dtype), gl[2], gl[3], gl[4], graph)
    ((aggr_g.float() * r.to(dev)).sum() + (pos_g * rp.to(dev)).sum()).backward()
    rtol = RTOL32 if dtype == torch.float32 else RTOL16
    assert_close(aggr_g.float(), aggr_w, rtol, what="aggr")
    assert_close(pos_g, pos_w, rtol, what="pos'")
    for name, got, want in zip(("gA", "gB", "gWe", "gpos", "gwpos"), gl, leaves):
        assert_close(got.grad.float(), want.grad, rtol, atol_scale=4.0, what=name)


def test_egnn_layer_module(dev):
    from mpnn_amd import EGNNLayer
    g = load_golden("layer_modelnet")
    torch.manual_seed(0)
    layer = EGNNLayer(3, 1, 64).to(dev)
    x = g["x"].to(dev)
    h, pos = layer(x, x, g["edge_index"].to(dev), g["edge_attr"].to(dev))
    want_aggr, want_pos = O.egnn_layer(*(t.detach().cpu() for t in (
        layer.fused_linear(x)[:, :64], layer.fused_linear(x)[:, 64:])), g["edge_attr"],
        layer.linear_3.weight.detach().cpu(), g["x"], layer.coord_linear.weight.detach().cpu().view(-1),
        g["edge_index"])
    assert_close(pos, want_pos, RTOL32, what="layer pos'")
    assert_close(h, O.update(g["x"], want_aggr, layer.linear_out.weight.detach().cpu()), RTOL32, what="layer h")
    (h.sum() + pos.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())


# --------------------------------------------------------------------------- dense MFMA GEMMs (fp32)
# Floating-point kernels: checked against torch float64 matmul of the same operands.
@pytest.mark.parametrize("n,k1,k2,m,relu", [
    (1000, 128, 0, 256, False),    # fused projection, hidden 128
    (777, 500, 64, 64, True),      # Flickr layer-1 update: x | aggr
    (300, 11, 64, 64, True),       # QM9 layer-1 update (unaligned K)
    (129, 3, 0, 128, False),       # ModelNet projection (K = 3)
    (260, 64, 0, 500, False),      # input gradient towards 500 features (two column slices)
    (64, 20, 12, 7, True),
])
def test_gemm_nt(n, k1, k2, m, relu, dev):
    from mpnn_amd.dense import gemm_nt
    gen = torch.Generator().manual_seed(n + m)
    x1, w1 = torch.randn(n, k1, generator=gen), torch.randn(m, k1 + k2, generator=gen)
    x2 = torch.randn(n, k2, generator=gen) if k2 else None
    mask = torch.randn(n, k1, generator=gen)
    want = (x1.double() * (mask > 0)) @ w1[:, :k1].double().t()
    if k2:
        want = want + x2.double() @ w1[:, k1:].double().t()
    if relu:
        want = want.relu()
    wd = w1.to(dev)
    got = gemm_nt(x1.to(dev), wd[:, :k1], x2.to(dev) if k2 else None, wd[:, k1:] if k2 else None,
                  mask1=mask.to(dev), relu=relu)
    assert_close(got, want, RTOL32, what="gemm_nt")


@pytest.mark.parametrize("n,m,k", [(5000, 256, 128), (3001, 64, 500), (100, 64, 11), (33, 7, 3), (4096, 128, 130),
                                   (600_000, 256, 64)])   # the last: two row tiles per block
def test_gemm_tn(n, m, k, dev):
    from mpnn_amd.dense import gemm_tn
    gen = torch.Generator().manual_seed(n + k)
    g, x, mask = torch.randn(n, m, generator=gen), torch.randn(n, k, generator=gen), torch.randn(n, m, generator=gen)
    want = (g.double() * (mask > 0)).t() @ x.double()
    wide = torch.zeros(m, k + 5, device=dev)
    got = gemm_tn(g.to(dev), x.to(dev), mask=mask.to(dev), out=wide[:, 5:])
    assert_close(got, want, RTOL32, what="gemm_tn")
    assert float(wide[:, :5].abs().sum()) == 0.0
    again = gemm_tn(g.to(dev), x.to(dev), mask=mask.to(dev))
    assert torch.equal(again, got.contiguous()), "weight gradient is not bitwise reproducible"


@pytest.mark.parametrize("n,m,k,k2", [(89_250, 192, 500, 0), (20_011, 192, 128, 64), (3_000, 132, 36, 0), (500, 160, 500, 4)])
def test_gemm_tn_with_129_to_192_rows(n, m, k, k2, dev, monkeypatch):
    """Weight gradients of 129 .. 192 rows ([gA | gB | gz]^T x of a 64-channel layer) take one 192-row output tile:
    float64 product at float32 rounding level, reproducible, equal to rounding with the 128-row tiles."""
    from mpnn_amd.dense import gemm_tn
    gen = torch.Generator().manual_seed(n + m)
    g, x = torch.randn(n, m, generator=gen).to(dev), torch.randn(n, k, generator=gen).to(dev)
    x2 = torch.randn(n, k2, generator=gen).to(dev) if k2 else None
    want = g.double().cpu().t() @ (torch.cat([x, x2], 1) if k2 else x).double().cpu()
    wide = torch.zeros(m, k + k2 + 4, device=dev)
    got = gemm_tn(g, x, x2=x2, out=wide[:, :k + k2])
    assert_close(got, want, RTOL32, what="gemm_tn 192-row tile")
    assert float(wide[:, k + k2:].abs().sum()) == 0.0
    assert torch.equal(gemm_tn(g, x, x2=x2), got.contiguous()), "not reproducible"
    monkeypatch.setenv("MPNN_GEMM_TN", "single")
    assert_close(gemm_tn(g, x, x2=x2), want, RTOL32, what="gemm_tn 128-row tiles")


# the last two are long enough (>= 2^19 rows) for the kernel that computes two tiles per block, one with an odd tile count
@pytest.mark.parametrize("n,m,k,k2", [(5000, 128, 128, 128), (3001, 64, 500, 64), (700, 16, 11, 16), (2048, 256, 132, 4),
                                      (530_000, 128, 128, 128), (524_295, 160, 132, 4)])
def test_gemm_tn_two_right_hand_matrices(n, m, k, k2, dev):
    """dW = (g * [mask > 0])^T [x | x2] in one pass against float64 and against the two separate products (which
    may sum their row slabs in another grouping, hence not bit for bit)."""
    from mpnn_amd.dense import gemm_tn
    gen = torch.Generator().manual_seed(n + k2)
    g, mask = torch.randn(n, m, generator=gen).to(dev), torch.randn(n, m, generator=gen).to(dev)
    x, x2 = torch.randn(n, k, generator=gen).to(dev), torch.randn(n, k2, generator=gen).to(dev)
    both = gemm_tn(g, x, mask=mask, x2=x2)
    want = (g.double().cpu() * (mask.cpu() > 0)).t() @ torch.cat([x, x2], 1).double().cpu()
    assert_close(both, want, RTOL32, what="gemm_tn two matrices")
    assert_close(both[:, :k], gemm_tn(g, x, mask=mask).double().cpu(), RTOL32, what="left part")
    assert_close(both[:, k:], gemm_tn(g, x2, mask=mask).double().cpu(), RTOL32, what="right part")
    assert torch.equal(both, gemm_tn(g, x, mask=mask, x2=x2)), "not reproducible"


# one case per layout of the stored mask: words of 32 lanes (fp32, H = 128) and of 16 (H = 64), 16-bit words
# (bf16, H = 128), a byte per lane (bf16 rows of 8 lanes: H = 64, and H = 56 with an idle lane), Fe = 1 and generic
@pytest.mark.parametrize("H,Fe,dtype", [(128, 4, torch.float32), (64, 4, torch.float32), (128, 4, torch.bfloat16),
                                        (64, 4, torch.bfloat16), (64, 1, torch.bfloat16), (64, 3, torch.bfloat16),
                                        (56, 4, torch.bfloat16)])
def test_recomputing_backward_matches_mask_bit_backward(H, Fe, dtype, dev):
    """The two backward forms (stored relu mask bits vs. recomputed pre-activation) give the same gradients bit for
    bit, and the forward the same sums whether or not it keeps the bits (rows of every length from 0 to thousands of
    edges: chunks with 1 to 8 edges, split rows)."""
    from mpnn_amd import ops
    lib = ops._lib.load()
    assert lib.mpnn_mask_words(H, ops._lib.dtype_code(dtype)) == (2 if dtype == torch.bfloat16 and H <= 64 else 4)
    a, want, _ = _aggregate_case(dev, 400, 20000, H, Fe, seed=9, skew=True, dtype=dtype)
    ops.USE_MASKBITS = False
    try:
        b, _, _ = _aggregate_case(dev, 400, 20000, H, Fe, seed=9, skew=True, dtype=dtype)
    finally:
        ops.USE_MASKBITS = True
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert_close(a[0], want[0], RTOL32 if dtype == torch.float32 else RTOL16, what="sums")


@pytest.mark.gpu
@pytest.mark.parametrize("n,k1,k2,m", [(5000, 128, 0, 128), (3000, 128, 256, 128), (2000, 500, 64, 64), (4096, 64, 0, 256)])
def test_split_bf16_gemm_is_as_accurate_as_the_float32_matrix_cores(n, k1, k2, m, monkeypatch):
    """The default NT GEMM multiplies on the bfloat16 matrix cores with every float32 operand split into three
    bfloat16 pieces; against a float64 product its error must not exceed the float32-MFMA form's by more than a
    small factor, and both must sit at float32 rounding level (far inside the 1e-5 parity bar)."""
    from mpnn_amd.dense import gemm_nt
    g = torch.Generator().manual_seed(n + m)
    x1 = torch.randn(n, k1, generator=g)
    w1 = torch.randn(m, k1, generator=g) / k1 ** 0.5
    x2 = torch.randn(n, k2, generator=g) if k2 else None
    w2 = torch.randn(m, k2, generator=g) / k2 ** 0.5 if k2 else None
    ref = x1.double() @ w1.double().t()
    if k2:
        ref = ref + x2.double() @ w2.double().t()
    cu = lambda t: t.cuda() if t is not None else None
    err = {}
    for mode in ("split", "f32"):
        monkeypatch.setenv("MPNN_GEMM", mode)
        y = gemm_nt(cu(x1), cu(w1), cu(x2), cu(w2)).cpu().double()
        err[mode] = float((y - ref).abs().max() / ref.abs().max())
    assert err["f32"] < 2e-6 and err["split"] < 2e-6, err
    assert err["split"] <= 2.0 * err["f32"] + 1e-7, err


@pytest.mark.gpu
@pytest.mark.parametrize("k1,k2,m,relu,masked,second", [
    (128, 0, 256, False, False, False), (128, 128, 128, True, False, False), (128, 0, 128, False, True, True),
    (256, 0, 128, False, True, False), (72, 40, 64, True, False, True), (128, 0, 100, False, False, False),
    (20, 0, 36, False, False, True)])
def test_gemm_nt_on_big_matrices(k1, k2, m, relu, masked, second, monkeypatch):
    """Matrices of >= 2^17 rows take the pre-split weight images: float32 rounding level of the float64 product, bit
    for bit the same as with the weights split inside the kernel, with a ragged last row step, a mask, relu and the
    second output (byte-sized multiplier)."""
    from mpnn_amd.dense import gemm_nt
    n = 131_072 + 77
    g = torch.Generator().manual_seed(k1 + m)
    x1, x2 = torch.randn(n, k1, generator=g), (torch.randn(n, k2, generator=g) if k2 else None)
    w = torch.randn(m, k1 + k2, generator=g) / (k1 + k2) ** 0.5
    mask = torch.randn(n, k1, generator=g) if masked else None
    emul = torch.randint(0, 200, (n, m), generator=g, dtype=torch.uint8) if second else None
    ref = (x1.double() * (mask > 0) if masked else x1.double()) @ w[:, :k1].double().t()
    if k2:
        ref = ref + x2.double() @ w[:, k1:].double().t()
    if relu:
        ref = ref.relu()
    cu = lambda t: t.cuda() if t is not None else None
    wd = w.cuda()
    out = {}
    for form in ("default", "nosplit"):
        if form != "default":
            monkeypatch.setenv("MPNN_GEMM_NT", form)
        wide = torch.zeros(n, m + 4, device="cuda") if second else None
        y = gemm_nt(cu(x1), wd[:, :k1], cu(x2), wd[:, k1:] if k2 else None, mask1=cu(mask), relu=relu, emul=cu(emul),
                    out2=wide[:, :m] if second else None)
        assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6, form
        if second:
            assert_close(wide[:, :m], ref * emul.double(), 2e-6, what=f"Y2 ({form})")
            assert float(wide[:, m:].abs().sum()) == 0.0
        out[form] = y
    assert torch.equal(out["default"], out["nosplit"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,k1,k2,m,relu,masked,add", [
    (89_250, 500, 0, 192, False, False, False),   # Flickr layer 1: projection with the update's x-part as extra columns
    (89_250, 128, 64, 500, False, False, False),  # its input gradient: two segments, two column blocks
    (20_001, 500, 0, 128, False, False, False),   # ragged last row step, ragged last k-chunk
    (9_000, 64, 0, 64, True, False, True),        # update from the projected part: relu(P + aggr Wa^T), narrow kernel
    (9_000, 64, 0, 128, True, False, True),       # the same through the wide kernel
    (30_000, 128, 128, 256, True, True, False),   # mask on the first operand
    (5_000, 36, 0, 320, False, True, True),       # two column blocks of 192, 20 columns past the end
    (777, 16, 0, 100, True, False, True),         # less than one block per CU, columns past the end
    (100, 500, 64, 65, False, False, False),
])
def test_gemm_nt_wide_products(n, k1, k2, m, relu, masked, add, monkeypatch):
    """Products of more than 64 columns can run on the kernel that divides the work between producer and consumer
    waves (k_gemm_nt_ws; by default those of 129 .. 192 columns do): float32 rounding level of the float64 product, bit for bit what k_gemm_nt gives (same piece
    products in the same order), with an addend, a relu, an operand mask and ragged edges in rows, columns and k."""
    from mpnn_amd.dense import gemm_nt
    g = torch.Generator().manual_seed(n + m)
    x1, x2 = torch.randn(n, k1, generator=g), (torch.randn(n, k2, generator=g) if k2 else None)
    w = torch.randn(m, k1 + k2, generator=g) / (k1 + k2) ** 0.5
    mask = torch.randn(n, k1, generator=g) if masked else None
    c = torch.randn(n, m + 3, generator=g)[:, :m] if add else None     # a view: leading dimension above M
    ref = (x1.double() * (mask > 0) if masked else x1.double()) @ w[:, :k1].double().t()
    if k2:
        ref = ref + x2.double() @ w[:, k1:].double().t()
    if add:
        ref = ref + c.double()
    if relu:
        ref = ref.relu()
    cu = lambda t: t.cuda() if t is not None else None
    wd = w.cuda()
    out = {}
    for form in ("ws", "old"):
        monkeypatch.setenv("MPNN_GEMM_NT", form)
        wide = torch.full((n, m + 5), 7.0, device="cuda")
        y = gemm_nt(cu(x1), wd[:, :k1], cu(x2), wd[:, k1:] if k2 else None, mask1=cu(mask), relu=relu, add=cu(c),
                    out=wide[:, :m])
        assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6, form
        assert bool((wide[:, m:] == 7.0).all()), "columns beside the output were written"
        out[form] = y.clone()
    assert torch.equal(out["ws"], out["old"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,m", [(5000, 64, 7), (89_250, 64, 7), (4100, 128, 19)])
def test_node_level_head_matches_torch_linear(n, k, m):
    """dense.head on a tall input: output, input gradient, weight gradient and bias gradient (the column sums that come
    out of the same product as the weight gradient) against float64; short inputs stay with the module."""
    from mpnn_amd import dense
    torch.manual_seed(n)
    lin = torch.nn.Linear(k, m).cuda()
    gen = torch.Generator().manual_seed(m)
    x, r = torch.randn(n, k, generator=gen), torch.randn(n, m, generator=gen)
    xd = x.double().requires_grad_(True)
    wd, bd = lin.weight.detach().double().cpu().requires_grad_(True), lin.bias.detach().double().cpu().requires_grad_(True)
    (torch.nn.functional.linear(xd, wd, bd) * r.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = dense.head(xg, lin)
    assert y.grad_fn is not None and "Head" in type(y.grad_fn).__name__
    (y * r.cuda()).sum().backward()
    assert_close(y, torch.nn.functional.linear(xd, wd, bd), RTOL32, what="head output")
    assert_close(xg.grad, xd.grad, RTOL32, what="grad x")
    assert_close(lin.weight.grad, wd.grad, RTOL32, what="grad weight")
    assert_close(lin.bias.grad, bd.grad, RTOL32, what="grad bias")
    short = dense.head(x[:100].cuda(), lin)
    assert "Head" not in type(short.grad_fn).__name__


# --------------------------------------------------------------------------- bfloat16 layer node
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sum", "cat"])
@pytest.mark.parametrize("n,e,f,h,fe", [(400, 5000, 16, 64, 4), (300, 9000, 128, 128, 4), (200, 1500, 8, 8, 1)])
def test_bf16_layer_node_against_the_oracle(mode, n, e, f, h, fe):
    """bfloat16 models whose widths are multiples of 8 run the layer as one node on mpnn_gemm_nt_bf16 + the edge
    kernels; against the float32 oracle on the same (bf16-rounded) inputs the bar is the bfloat16 one."""
    from mpnn_amd import MPNNLayerCat, MPNNLayerSum, layer_fn
    gen = torch.Generator().manual_seed(n + h)
    dst = (torch.rand(e, generator=gen) ** 3 * n).long().clamp_(max=n - 1)     # skewed: some split rows
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), dst])
    rnd = lambda *s: torch.randn(*s, generator=gen).bfloat16().float()
    x, ea, r = rnd(n, f), torch.rand(e, fe, generator=gen).bfloat16().float(), rnd(n, h)
    torch.manual_seed(7)
    layer = (MPNNLayerSum if mode == "sum" else MPNNLayerCat)(f, fe, h)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(p.bfloat16().float())
    params = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    xc = x.clone().requires_grad_(True)
    if mode == "sum":
        want = O.layer_sum(xc, ei, ea, params["fused_linear.weight"], params["linear_3.weight"], params["linear_out.weight"])
    else:
        want = O.layer_cat(xc, ei, ea, params["linear.weight"], params["linear_out.weight"])
    (want * r).sum().backward()

    layer = layer.cuda().bfloat16()
    xg = x.cuda().bfloat16().requires_grad_(True)
    assert layer_fn.supported(xg, ea.cuda().bfloat16(), *[p for p in layer.parameters()])
    got = layer(xg, ei.cuda(), ea.cuda().bfloat16())
    assert got.dtype == torch.bfloat16
    (got.float() * r.cuda()).sum().backward()
    assert_close(got.float(), want, RTOL16, what="out")
    assert_close(xg.grad.float(), xc.grad, RTOL16, atol_scale=2.0, what="grad x")
    for k, p in layer.named_parameters():
        assert_close(p.grad.float(), params[k].grad, RTOL16, atol_scale=2.0, what=f"grad {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("n,k1,k2,m,relu", [(5000, 128, 0, 256, False), (3001, 128, 128, 128, True), (77, 8, 24, 40, True)])
def test_gemm_nt_bf16(n, k1, k2, m, relu):
    """bf16 operands, float32 accumulation: the float64 product of the same bf16 values, rounded once to bf16."""
    from mpnn_amd.dense import gemm_nt_bf16
    gen = torch.Generator().manual_seed(n + m)
    bf = lambda *s: torch.randn(*s, generator=gen).bfloat16()
    x1, w = bf(n, k1), bf(m, k1 + k2)
    x2 = bf(n, k2) if k2 else None
    emul = torch.rand(n, m, generator=gen)
    want = x1.double() @ w[:, :k1].double().t()
    if k2:
        want = want + x2.double() @ w[:, k1:].double().t()
    if relu:
        want = want.relu()
    wd = w.cuda()
    out2 = torch.zeros(n, m + 8, dtype=torch.bfloat16, device="cuda")
    got = gemm_nt_bf16(x1.cuda(), wd[:, :k1].contiguous(), x2.cuda() if k2 else None,
                       wd[:, k1:].contiguous() if k2 else None, relu=relu, emul=emul.cuda(), out2=out2[:, :m])
    # one bf16 rounding of an (almost) exact sum: half an ulp = 2^-9 relative, plus the float32 accumulation noise
    assert_close(got.float(), want, 2.0 ** -8, what="gemm_nt_bf16")
    assert_close(out2[:, :m].float(), got.float().cpu().double() * emul.double(), 2.0 ** -8, what="gemm_nt_bf16 second output")
    assert float(out2[:, m:].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,k", [(5000, 256, 128), (3001, 64, 504), (100, 64, 16), (33, 8, 8), (4096, 128, 136)])
def test_gemm_tn_bf16(n, m, k):
    from mpnn_amd.dense import gemm_tn_bf16
    gen = torch.Generator().manual_seed(n + k)
    bf = lambda *s: torch.randn(*s, generator=gen).bfloat16()
    g, x, mask = bf(n, m), bf(n, k), bf(n, m).relu()          # a relu output as the mask: zeros and positives
    mask[0, 0] = -0.0
    gz = g.double() * (mask.double() > 0)
    want = gz.t() @ x.double()
    wide = torch.zeros(m, k + 8, dtype=torch.bfloat16, device="cuda")
    g_out = torch.empty(n, m, dtype=torch.bfloat16, device="cuda")
    got = gemm_tn_bf16(g.cuda(), x.cuda(), mask=mask.cuda(), out=wide[:, 8:], g_out=g_out)
    assert_close(got.float(), want, 2.0 ** -8, what="gemm_tn_bf16")
    assert float(wide[:, :8].abs().sum()) == 0.0
    assert torch.equal(g_out.cpu().double(), gz), "masked gradient by-product"
    again = gemm_tn_bf16(g.cuda(), x.cuda(), mask=mask.cuda())
    assert torch.equal(again, got.contiguous()), "weight gradient is not bitwise reproducible"
    plain = gemm_tn_bf16(g.cuda(), x.cuda())
    assert_close(plain.float(), g.double().t() @ x.double(), 2.0 ** -8, what="gemm_tn_bf16 unmasked")
    x2 = bf(n, 24)
    both = gemm_tn_bf16(g.cuda(), x.cuda(), mask=mask.cuda(), x2=x2.cuda())
    assert_close(both.float(), gz.t() @ torch.cat([x, x2], 1).double(), 2.0 ** -8, what="gemm_tn_bf16 two matrices")


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,m", [(1000, 64, 32), (3001, 128, 64), (5000, 128, 128), (2049, 36, 256), (700, 11, 16)])
def test_gemm_nt_second_output(n, k, m, dev):
    """Y and Y2 = Y * emul from one call (the epilogue staged through LDS for every tile width, and the plain
    epilogue for unaligned operands), with relu, into a column slice of a wider matrix."""
    from mpnn_amd.dense import gemm_nt
    gen = torch.Generator().manual_seed(n + m)
    x, w = torch.randn(n, k, generator=gen), torch.randn(m, k, generator=gen)
    emul = torch.randint(0, 50, (n, m), generator=gen).float()
    want = (x.double() @ w.double().t()).relu()
    wide = torch.zeros(n, 2 * m, device=dev)
    got = gemm_nt(x.to(dev), w.to(dev), relu=True, emul=emul.to(dev), out2=wide[:, :m])
    assert_close(got, want, RTOL32, what="Y")
    assert_close(wide[:, :m], want * emul.double(), RTOL32, what="Y2")
    assert float(wide[:, m:].abs().sum()) == 0.0


@pytest.mark.gpu
def test_gemm_variants_agree(monkeypatch, dev):
    """The forms kept for comparison (weights split inside the kernel, one output tile per block) agree with the
    defaults: NT bit for bit (same arithmetic, other data path), TN to float32 rounding (other slab grouping)."""
    from mpnn_amd.dense import gemm_nt, gemm_tn
    gen = torch.Generator().manual_seed(3)
    n = 140_000   # long enough for the pre-split weight images (>= 2^17 rows)
    x, x2, w = (torch.randn(n, 128, generator=gen).to(dev), torch.randn(n, 64, generator=gen).to(dev),
                torch.randn(128, 192, generator=gen).to(dev))
    g, mask = torch.randn(n, 256, generator=gen).to(dev), torch.randn(n, 256, generator=gen).to(dev)
    y = gemm_nt(x, w[:, :128], x2, w[:, 128:], relu=True)
    dw = gemm_tn(g, x, mask=mask, x2=x2)
    monkeypatch.setenv("MPNN_GEMM_NT", "nosplit")
    assert torch.equal(gemm_nt(x, w[:, :128], x2, w[:, 128:], relu=True), y)
    monkeypatch.delenv("MPNN_GEMM_NT")
    for form in ("single", "regs"):
        monkeypatch.setenv("MPNN_GEMM_TN", form)
        assert_close(gemm_tn(g, x, mask=mask, x2=x2), dw.double().cpu(), RTOL32, what=f"TN form {form}")


@pytest.mark.gpu
def test_bf16_weight_gradient_mask_on_special_values(dev):
    """The relu mask inside the bfloat16 weight-gradient product keeps g exactly where torch's `y <= 0 ? 0 : g` does:
    +0 / -0 / -inf / negative numbers drop, denormals, +inf and NaN of either sign keep (bit for bit, through the
    masked gradient the kernel writes out)."""
    from mpnn_amd.dense import gemm_tn_bf16
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(4096, 128, generator=gen).to(dev).bfloat16()
    x = torch.randn(4096, 128, generator=gen).to(dev).bfloat16()
    y = torch.randn(4096, 128, generator=gen).to(dev).bfloat16()
    special = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1e-38, -1e-38, 3.0, -3.0]).to(dev).bfloat16()
    neg_nan = torch.tensor([-1], dtype=torch.int16, device=dev).view(torch.bfloat16)      # 0xffff: NaN with the sign bit
    flat = y.view(-1)
    flat[:9] = special
    flat[9] = neg_nan[0]
    flat[4096:4105] = special          # the other half of a packed pair, another row
    flat[4105] = neg_nan[0]
    gz = torch.empty_like(g)
    out = torch.empty(128, 128, device=dev, dtype=torch.bfloat16)
    gemm_tn_bf16(g, x, mask=y, out=out, g_out=gz)
    want = torch.where(y <= 0, torch.zeros_like(g), g)
    assert torch.equal(gz.view(torch.int16), want.view(torch.int16))
    assert_close(out, want.double().cpu().t() @ x.double().cpu(), RTOL16, what="masked weight gradient")


@pytest.mark.gpu
@pytest.mark.parametrize("H", [64, 128])
def test_bf16_mask_bytes_at_the_chunk_boundaries(H, dev):
    """Rows with 0, 1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 64 and 65 in-edges: the lane-owned mask bytes of the
    bfloat16 forward are collected per word of 4 edges and per chunk of 8 (H = 64) or 16 (H = 128) edges, and a
    finished chunk's bytes are written while the next one is under way - every boundary of that scheme once.  Sums
    against the float64 oracle, and the backward that reads the bytes against the one that recomputes, bit for bit."""
    from mpnn_amd import ops
    from mpnn_amd.graph import MPGraph
    from mpnn_amd.ops import aggregate_fused
    degrees = [0, 1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 64, 65] * 3
    n, fe = len(degrees), 4
    gen = torch.Generator().manual_seed(H)
    dst = torch.repeat_interleave(torch.arange(n), torch.tensor(degrees))
    e = dst.numel()
    dst = dst[torch.randperm(e, generator=gen)]
    src = torch.randint(0, n, (e,), generator=gen)
    ei = torch.stack([src, dst])
    ab = torch.randn(n, 2 * H, generator=gen).bfloat16().float()
    ea = torch.rand(e, fe, generator=gen).bfloat16().float()
    we = (torch.randn(H, fe, generator=gen) * 0.5).bfloat16().float()
    r = torch.randn(n, H, generator=gen)
    want = O.aggregate_sum_form(ab[:, :H], ab[:, H:], ea, we, ei)

    def run():
        graph = MPGraph(ei.to(dev), n)
        ab_g = ab.to(dev, torch.bfloat16).requires_grad_(True)
        we_g = we.to(dev, torch.bfloat16).requires_grad_(True)
        got = aggregate_fused(ab_g, ea.to(dev, torch.bfloat16), we_g, graph)
        (got.float() * r.to(dev)).sum().backward()
        return got, ab_g.grad, we_g.grad

    with_bytes = run()
    ops.USE_MASKBITS = False
    try:
        recomputed = run()
    finally:
        ops.USE_MASKBITS = True
    for a, b in zip(with_bytes, recomputed):
        assert torch.equal(a, b)
    assert_close(with_bytes[0], want, RTOL16, what="sums")
