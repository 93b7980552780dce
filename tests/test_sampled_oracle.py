"""oracle/sampled.py (the checker of the full-size GPU runs) against the whole-graph oracle on a graph small enough
to evaluate in full: the rows it reports must equal the rows of the full evaluation."""
import torch

from conftest import assert_close
from oracle import reference_oracle as O
from oracle import sampled as S


def _problem(seed=3, n=400, e=6000, f=12, fe=4, h=16):
    gen = torch.Generator().manual_seed(seed)
    dst = (torch.rand(e, generator=gen) ** 3 * n).long().clamp_(max=n - 1)
    src = torch.randint(0, n, (e,), generator=gen)
    ei = torch.stack([src, dst])
    x = torch.randn(n, f, generator=gen, dtype=torch.float64)
    ea = torch.rand(e, fe, generator=gen, dtype=torch.float64)
    ws = [torch.randn(2 * h, f, generator=gen, dtype=torch.float64) * 0.3,
          torch.randn(h, fe, generator=gen, dtype=torch.float64) * 0.3,
          torch.randn(h, h + f, generator=gen, dtype=torch.float64) * 0.3]
    return n, ei, x, ea, ws, h, gen


def test_sparse_upstream_subgraph_equals_whole_graph():
    n, ei, x, ea, ws, h, gen = _problem()
    rows = S.sample_rows(ei, n, 1, 40, hubs=2, seed=1)
    assert int(torch.bincount(ei[1], minlength=n).argmax()) in rows.tolist()
    gy_rows = torch.randn(rows.numel(), h, generator=gen, dtype=torch.float64)
    got = S.layer_sum_sparse_upstream(x, ei, ea, ws[0], ws[1], ws[2], rows, gy_rows)

    xg = x.clone().requires_grad_(True)
    wg = [w.clone().requires_grad_(True) for w in ws]
    y = O.layer_sum(xg, ei, ea, *wg)
    gy = torch.zeros(n, h, dtype=torch.float64)
    gy[rows] = gy_rows
    y.backward(gy)
    assert_close(got["y_rows"], y[rows], 1e-12, what="y rows")
    assert_close(got["gx"], xg.grad[got["nodes"]], 1e-12, what="grad x")
    untouched = torch.ones(n, dtype=torch.bool)
    untouched[got["nodes"]] = False
    assert float(xg.grad[untouched].abs().max()) == 0.0
    for k, w in zip(("gw_fused", "gw_edge", "gw_out"), wg):
        assert_close(got[k], w.grad, 1e-12, what=k)
    assert got["num_edges"] == int(torch.isin(ei[1], rows).sum())


def test_edge_operator_rows_equal_whole_graph():
    n, ei, x, ea, ws, h, gen = _problem(seed=4)
    a = torch.randn(n, h, generator=gen, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(n, h, generator=gen, dtype=torch.float64).requires_grad_(True)
    g = torch.randn(n, h, generator=gen, dtype=torch.float64)
    out = O.aggregate_sum_form(a, b, ea, ws[1], ei)
    out.backward(g)
    dst_rows = S.sample_rows(ei, n, 1, 30, hubs=2, seed=2)
    src_rows = S.sample_rows(ei, n, 0, 30, hubs=2, seed=3)
    got = S.aggregate_rows(a.detach(), b.detach(), ea, ws[1], ei, g, dst_rows, src_rows)
    assert_close(got["out_rows"], out[dst_rows], 1e-12, what="out rows")
    assert_close(got["ga_rows"], a.grad[dst_rows], 1e-12, what="grad a rows")
    assert_close(got["gb_rows"], b.grad[src_rows], 1e-12, what="grad b rows")


def test_error_report_flags_a_wrong_tensor_and_tolerates_isolated_kinks():
    want = torch.linspace(-1, 1, 200_000, dtype=torch.float64)
    ok = S.error_report([("same", want.clone(), want)], 1e-5)
    assert ok["ok"] and ok["outside"] == 0 and ok["max_rel_err"] == 0.0
    kink = want.clone()
    kink[5] += 0.5                                    # one entry of 200,000: a relu kink, tolerated and counted
    rep = S.error_report([("kink", kink, want)], 1e-5)
    assert rep["ok"] and rep["outside"] == 1
    off = want * (1 + 1e-3)
    assert not S.error_report([("scaled", off, want)], 1e-5)["ok"]
    assert not S.error_report([("shape", want[:-1], want)], 1e-5)["ok"]
    bad = want.clone()
    bad[7] = float("nan")
    assert not S.error_report([("nan", bad, want)], 1e-5)["ok"]
