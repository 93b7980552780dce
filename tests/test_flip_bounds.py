"""oracle/flip_bounds.py against the autograd oracle it restates, and what its allowance lets through (CPU)."""
import torch

from oracle import flip_bounds as FB
from oracle import reference_oracle as O


def _graph(seed, n=60, e=400, f=7, fe=3, h=8):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    d = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    return dict(ei=ei, x=d(n, f), ea=d(e, fe), pos=d(n, 3), wn=d(2 * h, f) * 0.4, we=d(h, fe) * 0.4,
                wo=d(h, f + h) * 0.3, wp=d(1, h) * 0.2, r=d(n, h), rp=d(n, 3), n=n, h=h, f=f)


def test_explicit_layer_matches_the_autograd_oracle():
    c = _graph(0)
    leaves = {k: c[k].clone().requires_grad_(True) for k in ("x", "pos", "wn", "we", "wo", "wp")}
    y, pos_out = O.egnn_full_layer(leaves["x"], leaves["pos"], c["ei"], c["ea"], leaves["wn"], leaves["we"], leaves["wo"],
                                   leaves["wp"])
    ((y * c["r"]).sum() + (pos_out * c["rp"]).sum()).backward()
    t = FB.LayerTrace(c["x"], c["ei"], c["ea"], c["wn"], c["we"], c["wo"], 2e-6, c["pos"], c["wp"])
    assert torch.allclose(t.y, y.detach(), rtol=1e-12, atol=1e-12) and torch.allclose(t.pos_out, pos_out.detach(), rtol=1e-12, atol=1e-12)
    out = t.backward(c["r"], gpos=c["rp"])
    for mine, theirs in (("x", "x"), ("pos", "pos"), ("w_nodes", "wn"), ("w_edge", "we"), ("w_out", "wo"), ("w_pos", "wp")):
        g, bound, var = out[mine]
        assert torch.allclose(g.reshape(leaves[theirs].shape), leaves[theirs].grad, rtol=1e-10, atol=1e-10), mine
        assert float(bound.abs().max()) == 0.0 and float(var.abs().max()) == 0.0     # random data: nothing at zero
    # without coordinates: MPNNLayerSum
    x = c["x"].clone().requires_grad_(True)
    (O.layer_sum(x, c["ei"], c["ea"], c["wn"], c["we"], c["wo"]) * c["r"]).sum().backward()
    out = FB.LayerTrace(c["x"], c["ei"], c["ea"], c["wn"], c["we"], c["wo"], 2e-6).backward(c["r"])
    assert torch.allclose(out["x"][0], x.grad, rtol=1e-10, atol=1e-10) and "pos" not in out


def test_explicit_model_matches_the_autograd_oracle():
    c = _graph(1, f=8)
    g = torch.Generator().manual_seed(5)
    h, out_ch = c["h"], 5
    params = {}
    for k, f in enumerate((8, h)):
        params[f"layers.{k}.fused_linear.weight"] = torch.randn(2 * h, f, generator=g, dtype=torch.float64) * 0.4
        params[f"layers.{k}.linear_3.weight"] = c["we"].clone()
        params[f"layers.{k}.linear_out.weight"] = torch.randn(h, f + h, generator=g, dtype=torch.float64) * 0.3
        params[f"layers.{k}.coord_linear.weight"] = c["wp"].clone()
    params["output_layer.weight"] = torch.randn(out_ch, h, generator=g, dtype=torch.float64)
    params["output_layer.bias"] = torch.randn(out_ch, generator=g, dtype=torch.float64)
    batch = torch.sort(torch.randint(0, 6, (c["n"],), generator=g)).values
    b = int(batch.max()) + 1
    r = torch.randn(b, out_ch, generator=g, dtype=torch.float64)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    x, pos = c["x"].clone().requires_grad_(True), c["pos"].clone().requires_grad_(True)
    logits, pos_last = O.egnn_model_forward(leaves, x, pos, c["ei"], c["ea"], batch)
    ((logits * r).sum() + (pos_last * c["rp"]).sum()).backward()
    got_logits, got_pos, grads = FB.egnn_model(params, c["x"], c["pos"], c["ei"], c["ea"], batch, r, c["rp"], 2e-6)
    assert torch.allclose(got_logits, logits.detach(), rtol=1e-12, atol=1e-12)
    assert torch.allclose(got_pos, pos_last.detach(), rtol=1e-12, atol=1e-12)
    for k, v in leaves.items():
        assert torch.allclose(grads[k][0], v.grad, rtol=1e-9, atol=1e-10), k
    assert torch.allclose(grads["x"][0], x.grad, rtol=1e-9, atol=1e-10)
    assert torch.allclose(grads["pos"][0], pos.grad, rtol=1e-9, atol=1e-10)


def test_allowance_covers_a_flipped_decision_and_nothing_else():
    """One message pre-activation placed at zero to rounding: the gradient with that relu taken the other way
    passes, the same gradient with an unrelated entry moved by as much does not."""
    c = _graph(2)
    h, e0, col = c["h"], 17, 3
    ab = c["x"] @ c["wn"].t()
    es = c["ea"] @ c["we"].t()
    pre = ab[c["ei"][1, e0], col] + ab[c["ei"][0, e0], h + col] + es[e0, col]
    # move the edge term so that the pre-activation of (e0, col) becomes +1e-12: shift edge e0's attributes along We[col]
    w = c["we"][col]
    c["ea"][e0] -= w * (pre - 1e-12) / (w @ w)
    t = FB.LayerTrace(c["x"], c["ei"], c["ea"], c["wn"], c["we"], c["wo"], 2e-6)
    assert bool(t.open_msg[e0, col]) and int(t.open_msg.sum()) == 1 and bool(t.mask[e0, col])
    out = t.backward(c["r"])
    t.mask[e0, col] = False                       # the other correct evaluation
    flipped = t.backward(c["r"])
    names = ("x", "w_nodes", "w_edge", "w_out")
    rep = FB.report([(k, flipped[k][0], out[k]) for k in names], 1e-9)
    assert rep["ok"] and rep["through_flips"] > 0, rep
    delta = float((flipped["w_nodes"][0] - out["w_nodes"][0]).abs().max())
    assert delta > 0
    wrong = flipped["w_nodes"][0].clone()
    untouched = torch.nonzero(out["w_nodes"][1] == 0)[0]
    wrong[untouched[0], untouched[1]] += delta    # an entry no undecided relu reaches
    rep = FB.report([("w_nodes", wrong, out["w_nodes"])], 1e-9)
    assert not rep["ok"]
