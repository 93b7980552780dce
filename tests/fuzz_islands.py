"""Randomised sweep of the one-launch layers for batches of small graphs: random numbers and sizes of graphs (up to 32
nodes, some without edges, some dense with repeated edges and self loops), random input / edge widths, hidden 64 or
128, MPNNLayerSum / MPNNLayerCat / EGNNLayer, float32 and bfloat16, with and without the input gradient, against the
float64 oracle.  Usage: python tests/fuzz_islands.py [seed] [cases] [only this case]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from conftest import assert_close
from test_gpu_islands import _layer_case, _run, RTOL, KINK
from mpnn_amd import islands
from mpnn_amd.graph import graph_for, clear_cache
import test_gpu_islands as T

import itertools
import torch.nn.functional as F


def replay(case, kind, flip_msg=(), flip_out=()):
    """The layer in float64 with chosen relu decisions taken the other way: flip_msg = [(edge, column)] of the message
    relu (models/layers.py:56,128), flip_out = [(node, column)] of the update's (models/layers.py:69,140).  Returns the
    gradients {name: tensor} with 'x' and 'pos' among them, and the candidates for flipping: the decisions whose
    pre-activation is zero to float32 rounding (relative to the size of its summands)."""
    P = {k: v.detach().clone().requires_grad_(True) for k, v in case["params"].items()}
    x = case["xc"].detach().clone().requires_grad_(True)
    pos = case["pc"].detach().clone().requires_grad_(True)
    ea, (src, dst) = case["ea"].double(), case["ei"]
    n = x.size(0)
    if kind == "cat":
        w, f = P["linear.weight"], x.size(1)
        parts = (F.linear(x, w[:, :f])[dst], F.linear(x, w[:, f:2 * f])[src], F.linear(ea, w[:, 2 * f:]))
    else:
        xs, xt = F.linear(x, P["fused_linear.weight"]).chunk(2, dim=1)
        parts = (xs[dst], xt[src], F.linear(ea, P["linear_3.weight"]))
    pre = parts[0] + parts[1] + parts[2]
    mask = pre > 0
    for e, c in flip_msg:
        mask[e, c] = ~mask[e, c]
    m = pre * mask
    h = m.size(1)
    aggr = torch.zeros(n, h, dtype=torch.float64).index_add(0, dst, m)
    cat = torch.cat([x, aggr], 1)
    z = F.linear(cat, P["linear_out.weight"])
    zmask = z > 0
    for r, c in flip_out:
        zmask[r, c] = ~zmask[r, c]
    loss = (z * zmask * case["r"].double()).sum()
    if kind == "egnn":
        sc = (m * P["coord_linear.weight"].view(1, -1)).sum(1, keepdim=True)
        pos_out = pos + torch.zeros(n, 3, dtype=torch.float64).index_add(0, dst, (pos[dst] - pos[src]) * sc)
        loss = loss + (pos_out * case["rp"].double()).sum()
    loss.backward()
    grads = {k: v.grad for k, v in P.items()}
    grads["x"] = x.grad
    if kind == "egnn":
        grads["pos"] = pos.grad
    with torch.no_grad():
        mag = parts[0].abs() + parts[1].abs() + parts[2].abs()
        near_msg = torch.nonzero(pre.abs() < 1e-6 * mag).tolist()
        zmag = F.linear(cat.abs(), P["linear_out.weight"].abs())
        near_out = torch.nonzero(z.abs() < 1e-6 * zmag).tolist()
    return grads, near_msg, near_out


def explained_by_flips(case, kind, got: dict, rtol) -> str | None:
    """The float32 result differs from the float64 oracle by more than rounding.  Is it the oracle's result with some
    of the zero-to-rounding relu decisions taken the other way?  Returns the subset that explains it, or None."""
    _, near_msg, near_out = replay(case, kind)
    cands = [("msg", tuple(c)) for c in near_msg] + [("out", tuple(c)) for c in near_out]
    if not cands or len(cands) > 6:
        return None
    for k in range(1, len(cands) + 1):
        for subset in itertools.combinations(cands, k):
            grads, _, _ = replay(case, kind, [c for t, c in subset if t == "msg"], [c for t, c in subset if t == "out"])
            try:
                for name, g in got.items():
                    assert_close(g.float(), grads[name], rtol, atol_scale=4.0, what=name)
                return str(list(subset))
            except AssertionError:
                continue
    return None


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
g = torch.Generator().manual_seed(seed)
R = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
taken = 0
for c in range(cases):
    egnn = R(0, 2) == 0
    mode = "sum" if egnn or R(0, 1) else "cat"
    dtype = torch.bfloat16 if R(0, 2) == 0 else torch.float32
    f = [R(1, 20), 4 * R(1, 16), 32 * R(1, 4)][R(0, 2)]
    h = 64 if R(0, 3) else 128
    fe = R(1, 8)
    count, max_nodes, dense = R(1, 400), R(1, 32), R(0, 3) == 0
    if only is not None and c != only:
        continue
    orig = T.molecules
    T.molecules = lambda s, cnt, max_nodes=max_nodes, fe=4, f=11, dense=dense: orig(s, cnt, max_nodes, fe, f, dense)
    try:
        case = _layer_case(mode, dtype, f, h, fe, seed=1000 * seed + c, count=count, egnn=egnn)
    finally:
        T.molecules = orig
    clear_cache()
    got, got_pos, gx, gpos, grads, ei = _run(case, dtype, egnn)
    graph = graph_for(ei, case["n"])
    used = islands.plan_for(graph, f, h, fe, egnn) is not None
    taken += used
    rtol, kink = RTOL[dtype], KINK[dtype]
    tag = f"case {c}: {'egnn' if egnn else mode} {str(dtype)[6:]} F={f} H={h} Fe={fe} graphs={count} max_nodes={max_nodes} dense={dense} islands={used}"
    note = ""
    try:
        assert_close(got.float(), case["want"], rtol, what="y")
        if egnn:
            assert_close(got_pos, case["want_pos"], rtol, what="pos'")
        try:
            assert_close(gx.float(), case["xc"].grad, rtol, atol_scale=4.0, what="grad x", kink_frac=kink)
            if egnn:
                assert_close(gpos, case["pc"].grad, rtol, atol_scale=4.0, what="grad pos", kink_frac=kink)
            for k, gr in grads.items():
                assert_close(gr.float(), case["params"][k].grad, rtol, atol_scale=4.0, what=f"grad {k}", kink_frac=kink)
        except AssertionError as e:
            # float32: a pre-activation within rounding of zero (about one in 1e7; a dense case has 2e6 of them) can
            # take the other branch of the relu, which moves one row of the weight gradients and two rows of the
            # input gradient by one message's share.  Accept exactly that: the float64 result with those decisions
            # (and only those) taken the other way, at the strict bar.
            if dtype != torch.float32:
                raise
            mine = dict(grads, x=gx)
            if egnn:
                mine["pos"] = gpos
            why = explained_by_flips(case, "egnn" if egnn else mode, mine, rtol)
            if why is None:
                raise
            note = f"  (relu decisions at zero taken the other way: {why})"
    except AssertionError as e:
        print("FAILED", tag, "::", e)
        sys.exit(1)
    print("ok", tag + note, flush=True)
print(f"{cases} cases, {taken} through the island kernels")
