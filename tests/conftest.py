import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (run on the MI355X box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """Golden vectors as a dict of torch tensors (numpy.load, no pickle)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def assert_close(got, want, rtol, atol_scale=1.0, what="", kink_frac=0.0, kink_max=None):
    """|got - want| <= atol + rtol*|want| with atol = rtol * atol_scale * max|want| (outputs pass through relu
    with no bias, so many entries are exactly or nearly zero and a purely relative test is meaningless there).

    kink_frac: fraction of entries allowed outside the bound.  Two correct float32 evaluations of a relu network
    can disagree on the sign of a pre-activation that is zero to rounding; the gradient then differs by one whole
    term there.  Only the multi-million-entry full-size comparisons use a non-zero value (1e-5).
    kink_max: bound (relative to max|want|) on the error of the entries that kink_frac lets through - one
    message's worth of gradient, not an arbitrary amount."""
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = float(want.abs().max()) if want.numel() else 0.0
    atol = rtol * atol_scale * max(scale, 1e-30)
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = err > bound
    too_far = kink_max is not None and bad.any() and float(err[bad].max()) > kink_max * max(scale, 1e-30)
    if bad.any() and (int(bad.sum()) > kink_frac * bad.numel() or too_far):
        i = int(torch.argmax(err - bound))
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} outside rtol={rtol} atol={atol:.3e}; worst "
            f"got={got.flatten()[i].item():.8e} want={want.flatten()[i].item():.8e}")
