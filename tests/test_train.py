"""Loss and optimizer step: the oracle against torch's own Adam / cross_entropy (the library calls the reference
makes, training/train.py:3,47) on the CPU; the fused kernels, FlatAdam, the trainers and the captured step on the GPU."""
import types

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import reference_oracle as ro


def test_oracle_adam_matches_torch_adam():
    torch.manual_seed(0)
    p = torch.randn(257, dtype=torch.float64)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=3e-3)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, 8):
        g = torch.randn(257, dtype=torch.float64, generator=torch.Generator().manual_seed(t))
        ref.grad = g.clone()
        opt.step()
        p, m, v = ro.adam_step(p, g, m, v, t, lr=3e-3)
        assert_close(p, ref.detach(), 1e-12, what=f"step {t}")


def test_oracle_masked_ce_matches_torch_indexing():
    g = torch.Generator().manual_seed(3)
    out = torch.randn(500, 7, generator=g)
    y = torch.randint(0, 7, (500,), generator=g)
    mask = torch.rand(500, generator=g) < 0.4
    assert_close(ro.masked_cross_entropy(out, y, mask), F.cross_entropy(out[mask].double(), y[mask]), 1e-12)
    assert_close(ro.masked_cross_entropy(out, y), F.cross_entropy(out.double(), y), 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,frac", [(1, 7, 1.0), (1000, 7, 0.5), (89250, 7, 0.5), (33, 40, 0.1), (300000, 10, None)])
def test_masked_ce_kernel(n, c, frac):
    from mpnn_amd.loss import masked_cross_entropy
    g = torch.Generator().manual_seed(n)
    out = (torch.randn(n, c, generator=g) * 3).requires_grad_(True)
    y = torch.randint(0, c, (n,), generator=g)
    mask = None if frac is None else (torch.rand(n, generator=g) < frac)
    if mask is not None and not mask.any():
        mask[0] = True
    want = ro.masked_cross_entropy(out, y, mask)
    (gw,) = torch.autograd.grad(want * 1.7, out)
    xo = out.detach().cuda().requires_grad_(True)
    got = masked_cross_entropy(xo, y.cuda(), None if mask is None else mask.cuda())
    (gg,) = torch.autograd.grad(got * 1.7, xo)
    assert_close(got, want, 1e-5, what="loss")
    assert_close(gg, gw, 1e-5, what="grad")
    got2 = masked_cross_entropy(xo, y.cuda(), None if mask is None else mask.cuda())
    assert torch.equal(got, got2)      # fixed reduction order


@pytest.mark.gpu
def test_masked_ce_kernel_label_handling():
    """A label of -100 is F.cross_entropy's default ignore_index: the row leaves the sum, the count and the
    gradient.  Any other label outside [0, c) is an error in torch; the fused loss turns NaN instead of hiding it."""
    from mpnn_amd.loss import masked_cross_entropy
    g = torch.Generator().manual_seed(3)
    out = (torch.randn(200, 7, generator=g) * 2).requires_grad_(True)
    y = torch.randint(0, 7, (200,), generator=g)
    y[::9] = -100
    mask = torch.rand(200, generator=g) < 0.6
    want = F.cross_entropy(out[mask].double(), y[mask])
    (gw,) = torch.autograd.grad(want, out)
    xo = out.detach().cuda().requires_grad_(True)
    got = masked_cross_entropy(xo, y.cuda(), mask.cuda())
    (gg,) = torch.autograd.grad(got, xo)
    assert_close(got, want, 1e-5, what="loss with ignored rows")
    assert_close(gg, gw, 1e-5, what="grad with ignored rows")
    assert float(gg[::9].abs().max()) == 0.0
    y[1] = 7
    mask[1] = True
    assert torch.isnan(masked_cross_entropy(xo, y.cuda(), mask.cuda()))


@pytest.mark.gpu
def test_flat_adam_follows_torch_adam():
    from mpnn_amd import MPNN
    from mpnn_amd.optim import FlatAdam
    torch.manual_seed(1)
    a = MPNN(16, 32, 5, num_layers=2, mode="sum").cuda()
    torch.manual_seed(1)
    b = MPNN(16, 32, 5, num_layers=2, mode="sum").cuda()
    oa, ob = FlatAdam(a.parameters(), lr=2e-3), torch.optim.Adam(b.parameters(), lr=2e-3)
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(200, 16, device="cuda", generator=gen)
    ei = torch.randint(0, 200, (2, 1500), device="cuda", generator=gen)
    ea = torch.rand(1500, 4, device="cuda", generator=gen)
    y = torch.randint(0, 5, (200,), device="cuda", generator=gen)
    for _ in range(6):
        for model, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            F.cross_entropy(model(x, ei, ea), y).backward()
            opt.step()
    assert oa.steps_taken == 6
    for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert_close(pa, pb, 2e-5, what=k)
    # the parameters are views of the flat buffer, and so are their gradients
    assert all(p.data_ptr() >= oa.flat.data_ptr() for p in a.parameters())
    assert all(p.grad.data_ptr() >= oa.flat_grad.data_ptr() for p in a.parameters())
    # between backward() and step() - after the default zero_grad, which drops the views - reading flat_grad gathers
    # the gradients first: clipping or reducing the vector sees the current values
    oa.zero_grad()
    F.cross_entropy(a(x, ei, ea), y).backward()
    assert any(p.grad.data_ptr() < oa.flat.data_ptr() or p.grad.data_ptr() >= oa.flat.data_ptr() for p in a.parameters())
    want = torch.cat([p.grad.reshape(-1) for p in a.parameters()]).clone()
    assert torch.equal(oa.flat_grad, want)
    assert all(p.grad.data_ptr() >= oa.flat_grad.data_ptr() for p in a.parameters())


def _flickr_like(n=3000, e=20000, f=50):
    from mpnn_amd import synth
    d = synth.flickr("cuda", 0, n, e, f)
    d["test_mask"] = ~d["train_mask"]
    return types.SimpleNamespace(**d)


@pytest.mark.gpu
def test_trainers_match_the_reference_step_written_with_torch_ops():
    from mpnn_amd import MPNN
    from mpnn_amd.optim import FlatAdam
    from mpnn_amd.train import Trainer_Flickr
    data = _flickr_like()
    torch.manual_seed(2)
    a = MPNN(50, 32, 7, num_layers=2, mode="cat").cuda()
    torch.manual_seed(2)
    b = MPNN(50, 32, 7, num_layers=2, mode="cat").cuda()
    ta = Trainer_Flickr(a, FlatAdam(a.parameters(), lr=1e-2), F.cross_entropy, "cuda")
    ob = torch.optim.Adam(b.parameters(), lr=1e-2)
    for _ in range(4):
        la = ta.train(data)
        # the reference's step, literally (training/train.py:41-49)
        b.train()
        ob.zero_grad()
        out = b(data.x, data.edge_index, data.edge_attr)
        loss = F.cross_entropy(out[data.train_mask], data.y[data.train_mask])
        loss.backward()
        ob.step()
        assert abs(la - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    acc = ta.test(data)
    assert 0.0 <= acc <= 1.0


@pytest.mark.gpu
def test_captured_step_equals_eager_steps():
    from mpnn_amd import MPNN
    from mpnn_amd.loss import masked_cross_entropy
    from mpnn_amd.optim import FlatAdam
    from mpnn_amd.train import CapturedStep
    data = _flickr_like()
    models, opts = [], []
    for _ in range(2):
        torch.manual_seed(4)
        m = MPNN(50, 32, 7, num_layers=2, mode="sum").cuda()
        models.append(m)
        opts.append(FlatAdam(m.parameters(), lr=5e-3))

    def make_step(m, o):
        def step():
            o.zero_grad()
            loss = masked_cross_entropy(m(data.x, data.edge_index, data.edge_attr), data.y, data.train_mask)
            loss.backward()
            o.step()
            return loss
        return step

    eager = make_step(models[0], opts[0])
    # capture runs warm-up steps, which are real optimizer steps: give the eager twin the same number first
    captured = CapturedStep(make_step(models[1], opts[1]), warmup=3)
    for _ in range(3):
        eager()
    for _ in range(5):
        le = eager()
        lc = captured.replay()
        assert torch.equal(le, lc)
    for pa, pb in zip(models[0].parameters(), models[1].parameters()):
        assert torch.equal(pa, pb)
