"""CPU tests of the oracle itself: it must reproduce the committed golden vectors, agree with
brute-force definitions written independently of it, and satisfy the identities the HIP engine
relies on (cat-form == split-weight sum-form)."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden
from oracle import reference_oracle as O

LAYER_CASES = ["layer_tiny", "layer_empty", "layer_qm9", "layer_modelnet", "layer_hub", "layer_odd"]


@pytest.mark.parametrize("case", LAYER_CASES)
def test_oracle_reproduces_golden_layers(case):
    torch.set_num_threads(1)
    g = load_golden(case)
    ei, x, ea = g["edge_index"], g["x"], g["edge_attr"]
    out, aggr = O.layer_cat(x, ei, ea, g["w_cat_linear"], g["w_cat_linear_out"], return_aggr=True)
    assert_close(out, g["cat_out"], 1e-6, what="cat out")
    assert_close(aggr, g["cat_aggr"], 1e-6, what="cat aggr")
    out, aggr = O.layer_sum(x, ei, ea, g["w_sum_fused_linear"], g["w_sum_linear_3"], g["w_sum_linear_out"],
                            return_aggr=True)
    assert_close(out, g["sum_out"], 1e-6, what="sum out")
    assert_close(aggr, g["sum_aggr"], 1e-6, what="sum aggr")
    # float32 oracle against its own float64 recomputation: the rounding budget of the parity bar
    assert_close(g["sum_out"], g["sum_out64"], 1e-5, what="sum fp32 vs fp64")
    assert_close(g["cat_out"], g["cat_out64"], 1e-5, what="cat fp32 vs fp64")


def brute_force_layer(x, ei, ea, message):
    """Edge-by-edge Python loop, written without the oracle's gather/scatter helpers."""
    n = x.size(0)
    out = None
    for e in range(ei.size(1)):
        j, i = int(ei[0, e]), int(ei[1, e])
        m = message(i, j, e)
        if out is None:
            out = torch.zeros(n, m.numel(), dtype=m.dtype)
        out[i] += m
    return out


def test_oracle_matches_edge_loop_definition():
    g = load_golden("layer_tiny")
    x, ei, ea = g["x"].double(), g["edge_index"], g["edge_attr"].double()
    w = g["w_cat_linear"].double()
    want = brute_force_layer(x, ei, ea, lambda i, j, e: torch.relu(w @ torch.cat([x[i], x[j], ea[e]])))
    got = O.scatter_sum(O.cat_message(x, ei, ea, w), ei, x.size(0))
    assert_close(got, want, 1e-12, what="cat aggregate vs loop")
    wf, w3 = g["w_sum_fused_linear"].double(), g["w_sum_linear_3"].double()
    fused = x @ wf.t()
    h = w3.size(0)
    want = brute_force_layer(x, ei, ea, lambda i, j, e: torch.relu((fused[i, :h] + fused[j, h:]) + w3 @ ea[e]))
    got = O.scatter_sum(O.sum_message(x, ei, ea, wf, w3), ei, x.size(0))
    assert_close(got, want, 1e-12, what="sum aggregate vs loop")
    # node 3 and node 11 have no incoming edge: their aggregate is exactly zero
    assert float(got[3].abs().sum()) == 0.0 and float(got[11].abs().sum()) == 0.0


@pytest.mark.parametrize("case", ["layer_qm9", "layer_modelnet", "layer_hub"])
def test_cat_equals_split_weight_sum_form(case):
    """relu(W [x_i | x_j | e]) == relu((Wi x)_i + (Wj x)_j + We e): the identity that lets one edge kernel serve
    both modes (reference models/layers.py:22,56 against :104-110,128)."""
    g = load_golden(case)
    x, ei, ea, w = g["x"], g["edge_index"], g["edge_attr"], g["w_cat_linear"]
    f = x.size(1)
    a = x @ w[:, :f].t()
    b = x @ w[:, f:2 * f].t()
    got = O.aggregate_sum_form(a, b, ea, w[:, 2 * f:], ei)
    assert_close(got, g["cat_aggr"], 1e-5, what="split-weight form")


@pytest.mark.parametrize("case", LAYER_CASES)
def test_csr_oracle_against_numpy_stable_sort(case):
    g = load_golden(case)
    ei = g["edge_index"].numpy()
    n = int(g["num_nodes"])
    for key_row, other_row, names in ((1, 0, ("rowptr", "col", "eid")), (0, 1, ("rowptr_t", "col_t", "eid_t"))):
        perm = np.argsort(ei[key_row], kind="stable")
        rowptr = np.searchsorted(ei[key_row][perm], np.arange(n + 1), side="left")
        assert np.array_equal(g["csr_" + names[0]].numpy(), rowptr.astype(np.int32))
        assert np.array_equal(g["csr_" + names[1]].numpy(), ei[other_row][perm].astype(np.int32))
        assert np.array_equal(g["csr_" + names[2]].numpy(), perm.astype(np.int32))
    got = O.csr_build(g["edge_index"], n)
    for k, v in got.items():
        assert torch.equal(v, g["csr_" + k])
    assert int(g["csr_deg_in"].sum()) == ei.shape[1] == int(g["csr_deg_out"].sum())


def test_work_items_cover_every_edge_once():
    gen = torch.Generator().manual_seed(0)
    n, e = 300, 20000
    dst = (torch.rand(e, generator=gen) ** 5 * n).long()
    rowptr = O.csr_by_key(dst, torch.zeros(e, dtype=torch.int64), n)[0]
    items, split, parts = O.work_items(rowptr, 64, 8)
    covered = torch.zeros(e, dtype=torch.int32)
    for row, e0, e1, slot in items.tolist():
        assert rowptr[row] <= e0 <= e1 <= rowptr[row + 1]
        covered[e0:e1] += 1
        deg = int(rowptr[row + 1] - rowptr[row])
        assert (slot == -1) == (deg <= 64)
    assert bool((covered == 1).all())
    assert parts == sum(p for _, _, p, _ in split.tolist())
    for row, first, pieces, first_item in split.tolist():
        assert 2 <= pieces <= 8
        assert items[first_item:first_item + pieces, 0].eq(row).all()
        assert items[first_item:first_item + pieces, 3].tolist() == list(range(first, first + pieces))
    golden = load_golden("layer_hub")
    assert int(golden["n_parts"]) > 0  # the committed hub case exercises split rows


def test_global_mean_pool_oracle():
    g = load_golden("pool")
    x, batch = g["x"], g["batch_sorted"]
    for k in range(6):
        rows = x[batch == k]
        want = rows.mean(0) if rows.size(0) else torch.zeros(x.size(1))
        assert_close(g["out_sorted"][k], want, 1e-6, what=f"graph {k}")
    assert_close(O.global_mean_pool(x, batch, 6), g["out_sorted"], 1e-6)
    assert_close(g["out_unsorted"], O.global_mean_pool(x, batch[g["perm"]], 6), 1e-6)


def test_model_golden_reproduced():
    torch.set_num_threads(1)
    g = load_golden("model_qm9")
    for mode in ("cat", "sum"):
        params = {k[len(mode) + 7:]: v for k, v in g.items() if k.startswith(f"{mode}.param.")}
        logits = O.model_forward(params, mode, g["x"], g["edge_index"], g["edge_attr"], g["batch"])
        assert_close(logits, g[f"{mode}.logits"], 1e-6, what=f"{mode} logits")


def test_dst_range_partition_balances_edges():
    gen = torch.Generator().manual_seed(3)
    n, e = 1000, 50000
    dst = (torch.rand(e, generator=gen) ** 3 * n).long()
    rowptr = O.csr_by_key(dst, torch.zeros(e, dtype=torch.int64), n)[0]
    for world in (1, 2, 4, 8):
        b = O.dst_range_partition(rowptr, world)
        assert b[0] == 0 and b[-1] == n and len(b) == world + 1 and b == sorted(b)
        counts = [int(rowptr[b[r + 1]] - rowptr[b[r]]) for r in range(world)]
        assert sum(counts) == e
        assert max(counts) <= e / world + int((rowptr[1:] - rowptr[:-1]).max())


def test_default_chunk_rule():
    """Shortest split piece as a function of the edge count: 64 up to 2^23 - 1 edges, then the largest power of two
    <= E / 65536, capped at 1024."""
    from mpnn_amd.graph import CHUNK, CHUNK_MAX, default_chunk
    assert (CHUNK, CHUNK_MAX) == (64, 1024)
    for e, want in [(0, 64), (899_756, 64), (2 ** 23 - 1, 64), (2 ** 23, 128), (9_000_000, 128), (2 ** 24, 256),
                    (2 ** 26 - 1, 512), (2 ** 26, 1024), (100_000_000, 1024), (2 ** 31 - 1, 1024)]:
        assert default_chunk(e) == want, (e, default_chunk(e), want)


def test_rows_in_item_order_oracle_lists_rows_as_the_work_items_do():
    """The oracle's item-order view against its own work_items: rows appear in the order of their first item, every
    row keeps its edges, and the work items of the moved view are one ascending run over the edge arrays."""
    gen = torch.Generator().manual_seed(4)
    n, e, chunk = 300, 9000, 16
    dst = (torch.rand(e, generator=gen) ** 4 * n).long().clamp_(max=n - 1)     # skewed: long rows that get split
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), dst])
    ref = O.csr_build(ei, n)
    order, rowptr, col, eid = O.rows_in_item_order(ref["rowptr"], ref["col"], ref["eid"], chunk)
    items, split, _ = O.work_items(ref["rowptr"], chunk, 256)
    assert order.tolist() == list(dict.fromkeys(items[:, 0].tolist())) and split.numel() > 0
    assert int(rowptr[0]) == 0 and int(rowptr[-1]) == e
    for j in (0, 1, 7, n // 2, n - 1):
        row = int(order[j])
        got = slice(int(rowptr[j]), int(rowptr[j + 1]))
        want = slice(int(ref["rowptr"][row]), int(ref["rowptr"][row + 1]))
        assert torch.equal(col[got], ref["col"][want]) and torch.equal(eid[got], ref["eid"][want])
    moved, _, _ = O.work_items(rowptr, chunk, 256)
    assert torch.equal(moved[1:, 1], moved[:-1, 2]) and int(moved[0, 1]) == 0 and int(moved[-1, 2]) == e


@pytest.mark.gpu
@pytest.mark.parametrize("n,e,chunk,power", [(300, 9000, 16, 4), (5000, 200_000, 64, 6), (1, 50, 8, 1), (777, 0, 64, 1),
                                             (40_000, 3_000_000, 256, 8)])
def test_rows_in_item_order_on_the_device_are_bit_exact(n, e, chunk, power):
    """mpnn_csr_item_order against the oracle: order, row pointers, neighbour and edge-id lists, bit for bit."""
    from mpnn_amd.graph import MPGraph
    gen = torch.Generator().manual_seed(n + chunk)
    dst = (torch.rand(e, generator=gen) ** power * n).long().clamp_(max=n - 1)
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), dst])
    ref = O.csr_build(ei, n)
    g = MPGraph(ei.cuda(), n, chunk=chunk, row_order="items")
    for view, names in ((g.by_dst, ("rowptr", "col", "eid")), (g.by_src, ("rowptr_t", "col_t", "eid_t"))):
        if e == 0:
            assert view.order is None and int(view.rowptr.abs().sum()) == 0
            continue
        order, rowptr, col, eid = O.rows_in_item_order(*(ref[k] for k in names), chunk)
        assert torch.equal(view.order.cpu(), order)
        assert torch.equal(view.rowptr.cpu(), rowptr)
        assert torch.equal(view.col.cpu(), col) and torch.equal(view.eid.cpu(), eid)
        assert torch.equal(view.degree().cpu(), (ref[names[0]][1:] - ref[names[0]][:-1]))
    with pytest.raises(IndexError):
        bad = ei.clone().cuda()
        if e:
            bad[1, 0] = n
            MPGraph(bad, n, chunk=chunk, row_order="items")
        else:
            raise IndexError


def test_islands_oracle_against_brute_force():
    """O.islands (the specification the device search is compared with) against the definition itself: a cut after
    node r stands iff no edge has its ends on both sides."""
    g = torch.Generator().manual_seed(0)
    for _ in range(60):
        n = int(torch.randint(1, 40, (1,), generator=g))
        e = int(torch.randint(0, 30, (1,), generator=g))
        far = torch.randint(0, n, (2, e), generator=g)
        near = far.clone()
        near[1] = (near[0] + torch.randint(-2, 3, (e,), generator=g)).clamp(0, n - 1)
        for ei in (far, near):
            want = [0] + [r + 1 for r in range(n) if not any(min(a, b) <= r < max(a, b) for a, b in ei.t().tolist())]
            assert O.islands(ei, n) == want
