"""EGNN layer and model on the GPU against the CPU oracle, at the two configurations the benchmark names:
a QM9-shaped batch of 1,024 molecules in bfloat16 and a ModelNet-shaped batch of 32 clouds (k-NN, k = 20) in float32.

PARITY UNPINNED: the reference holds no EGNN code (its models/layers.py has MPNNLayerCat and MPNNLayerSum only);
the oracle is the formula pos_i += sum_j (pos_i - pos_j) * phi_x(m_ij) around the reference's sum-form message
(models/layers.py:104-110,128,140), stated in oracle/reference_oracle.py.

Bars: float32 1e-5, bfloat16 2e-2, both relative with the usual floor (outputs pass a bias-free relu); outputs
AND every gradient: x, pos, fused_linear, linear_3, linear_out, coord_linear (and the head for the model).
The oracle is evaluated in float64 on the values the GPU holds: its float32 evaluation is itself 3e-6 (QM9 batch)
to 6e-4 (point clouds, through relu decisions that fall on the other side of zero) away from the float64 one on
the weight gradients of the two-layer model (tools/egnn_model_debug.py prints both distances), which would leave
no budget to the kernels under test.
"""
import pytest
import torch

from conftest import assert_close
from oracle import reference_oracle as O

pytestmark = pytest.mark.gpu
RTOL = {torch.float32: 1e-5, torch.bfloat16: 2e-2}
# Share of a GRADIENT tensor's entries that may miss the bar.  bfloat16 storage of the projected rows moves
# pre-activations by up to 2^-9 relative, so a few of them change sign against the float32 oracle; a message
# whose relu flips contributes a whole term (not a rounding error) to the few gradient entries it feeds.
# Forward outputs are continuous in the pre-activation and get no such allowance.
KINK = {torch.float32: 0.0, torch.bfloat16: 2e-3}


def _data(workload, dtype):
    from mpnn_amd import synth
    dev = torch.device("cuda:0")
    d = synth.qm9(dev, 0, 1024) if workload == "qm9" else synth.modelnet(dev, 0)
    x = d["x"].to(dtype)
    ea = d["edge_attr"].to(dtype)
    return d, x, ea, d["pos"].float().clone()


def _rounded(layer_or_model, dtype):
    """Parameters rounded to the storage dtype, so that the float32 oracle sees the values the GPU holds."""
    with torch.no_grad():
        for p in layer_or_model.parameters():
            p.copy_(p.to(dtype).float())
    return {k: v.detach().clone().double().requires_grad_(True) for k, v in layer_or_model.state_dict().items()}


CASES = [pytest.param("qm9", torch.bfloat16, id="qm9-1024-bf16"), pytest.param("qm9", torch.float32, id="qm9-1024-f32"),
         pytest.param("modelnet", torch.float32, id="modelnet-32-f32"),
         pytest.param("modelnet", torch.bfloat16, id="modelnet-32-bf16")]


@pytest.mark.parametrize("workload,dtype", CASES)
def test_egnn_layer_outputs_and_every_gradient(workload, dtype):
    from mpnn_amd import EGNNLayer
    d, x, ea, pos = _data(workload, dtype)
    ei = d["edge_index"]
    n, f, fe, h = x.size(0), x.size(1), ea.size(1), 64
    rtol = RTOL[dtype]
    torch.manual_seed(3)
    layer = EGNNLayer(f, fe, h)
    params = _rounded(layer, dtype)
    gen = torch.Generator().manual_seed(4)
    r, rp = torch.randn(n, h, generator=gen), torch.randn(n, 3, generator=gen)

    xc = x.double().cpu().requires_grad_(True)
    pc = pos.double().cpu().requires_grad_(True)
    want_h, want_pos = O.egnn_full_layer(xc, pc, ei.cpu(), ea.double().cpu(), params["fused_linear.weight"],
                                         params["linear_3.weight"], params["linear_out.weight"],
                                         params["coord_linear.weight"])
    ((want_h * r.double()).sum() + (want_pos * rp.double()).sum()).backward()

    layer = layer.cuda().to(dtype)
    xg = x.clone().requires_grad_(True)
    pg = pos.clone().requires_grad_(True)
    got_h, got_pos = layer(xg, pg, ei, ea)
    assert got_h.dtype == dtype and got_pos.dtype == torch.float32
    ((got_h.float() * r.cuda()).sum() + (got_pos * rp.cuda()).sum()).backward()
    assert_close(got_h.float(), want_h, rtol, what="h")
    assert_close(got_pos, want_pos, rtol, what="pos'")
    kink = KINK[dtype]
    assert_close(xg.grad.float(), xc.grad, rtol, atol_scale=4.0, what="grad x", kink_frac=kink)
    assert_close(pg.grad, pc.grad, rtol, atol_scale=4.0, what="grad pos", kink_frac=kink)
    for k, p in layer.named_parameters():
        assert p.grad.shape == p.shape
        assert_close(p.grad.float(), params[k].grad, rtol, atol_scale=4.0, what=f"grad {k}", kink_frac=kink)

    # a second run gives the same bits (fixed summation order, no atomics)
    for p in layer.parameters():
        p.grad = None
    xg2, pg2 = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    h2, p2 = layer(xg2, pg2, ei, ea)
    ((h2.float() * r.cuda()).sum() + (p2 * rp.cuda()).sum()).backward()
    assert torch.equal(h2, got_h) and torch.equal(p2, got_pos)
    assert torch.equal(xg2.grad, xg.grad) and torch.equal(pg2.grad, pg.grad)


@pytest.mark.parametrize("workload,dtype", CASES[:3])
def test_egnn_model_logits_and_every_gradient(workload, dtype):
    """Two EGNN layers -> mean pool -> head (the MPNN_QM9 shape, models/mpnn.py:38-55) with a full-batch loss."""
    from mpnn_amd import EGNN
    d, x, ea, pos = _data(workload, dtype)
    ei, batch = d["edge_index"], d["batch"]
    f, fe, h, out = x.size(1), ea.size(1), 64, 19
    rtol = RTOL[dtype]
    torch.manual_seed(5)
    model = EGNN(f, fe, h, out, num_layers=2)
    params = _rounded(model, dtype)
    b = int(batch.max()) + 1
    gen = torch.Generator().manual_seed(6)
    r, rp = torch.randn(b, out, generator=gen), torch.randn(x.size(0), 3, generator=gen) * 0.1

    xc = x.double().cpu().requires_grad_(True)
    pc = pos.double().cpu().requires_grad_(True)
    want, want_pos = O.egnn_model_forward(params, xc, pc, ei.cpu(), ea.double().cpu(), batch.cpu())
    ((want * r.double()).sum() + (want_pos * rp.double()).sum()).backward()

    model = model.cuda().to(dtype)
    xg, pg = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    got, got_pos = model(xg, pg, ei, batch, ea, return_pos=True)
    assert got.shape == (b, out)
    ((got.float() * r.cuda()).sum() + (got_pos * rp.cuda()).sum()).backward()
    # two stacked bfloat16 layers: the second sees inputs that already carry the first one's rounding
    loose = 1.0 if dtype == torch.float32 else 2.0
    assert_close(got.float(), want, rtol * loose, what="logits")
    assert_close(got_pos, want_pos, rtol * loose, what="pos after the last layer")
    # Gradients: the second layer's inputs differ from the oracle's by the first one's rounding, so a few of its
    # ~10^6 pre-activations land on the other side of zero (the QM9 batch has one of 1.7e-8 in float64 against a
    # typical 0.26 - tools/egnn_flip_debug.py prints it), and each such relu moves every gradient it feeds by a
    # whole term.  An entry outside the bar passes only within what the undecided relus that reach it explain
    # (oracle/flip_bounds.py: worst case, or 8 standard deviations of independent flips for million-term sums) -
    # no share of a tensor is waved through, and an entry no undecided relu reaches must meet the bar itself.
    from oracle import flip_bounds as FB
    _, _, want_g = FB.egnn_model({k: v.detach() for k, v in params.items()}, x.double().cpu(), pos.double().cpu(),
                                 ei.cpu(), ea.double().cpu(), batch.cpu(), r, rp, FB.DELTA[dtype])
    for k in params:     # the explicit statement is the autograd oracle's
        assert_close(want_g[k][0], params[k].grad, 1e-9, what=f"oracle grad {k}")
    entries = [("grad x", xg.grad.float(), want_g["x"], 4.0), ("grad pos", pg.grad, want_g["pos"], 4.0)]
    entries += [(f"grad {k}", p.grad.float(), want_g[k], 4.0) for k, p in model.named_parameters()]
    rep = FB.report(entries, rtol * loose)
    assert rep["ok"], rep
    print(f"\n{rep}")


def test_bf16_layers_with_input_widths_that_are_not_multiples_of_8_stay_on_the_library_gemms():
    """QM9's 11 atom features and a cloud's 3 coordinates are padded with zero columns for the bfloat16 GEMMs;
    widths the kernels cannot take raise instead of falling back to torch's GEMMs."""
    from mpnn_amd import MPNNLayerSum, dense
    dev = torch.device("cuda:0")
    d, x, ea, _ = _data("qm9", torch.bfloat16)
    torch.manual_seed(0)
    layer = MPNNLayerSum(11, 4, 64)
    params = _rounded(layer, torch.bfloat16)
    xc = x.double().cpu().requires_grad_(True)
    want = O.layer_sum(xc, d["edge_index"].cpu(), ea.double().cpu(), params["fused_linear.weight"],
                       params["linear_3.weight"], params["linear_out.weight"])
    want.sum().backward()
    layer = layer.to(dev).bfloat16()
    xg = x.clone().requires_grad_(True)
    got = layer(xg, d["edge_index"], ea)
    got.float().sum().backward()
    assert_close(got.float(), want, 2e-2, what="bf16 layer, 11 input features")
    assert xg.grad.shape == x.shape
    assert_close(xg.grad.float(), xc.grad, 2e-2, atol_scale=4.0, what="grad x", kink_frac=KINK[torch.bfloat16])
    for k, p in layer.named_parameters():
        assert p.grad.shape == p.shape
        assert_close(p.grad.float(), params[k].grad, 2e-2, atol_scale=4.0, what=f"grad {k}",
                     kink_frac=KINK[torch.bfloat16])
    with pytest.raises(ValueError):
        dense.project(x, layer.fused_linear.weight)          # 11 columns: the caller must pad
    with pytest.raises(TypeError):
        MPNNLayerSum(16, 4, 12).to(dev).bfloat16()(torch.zeros(x.size(0), 16, device=dev, dtype=torch.bfloat16),
                                                   d["edge_index"], ea)   # hidden width 12: no 16-byte rows
