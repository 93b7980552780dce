"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol the header
declares, validates its arguments before touching the GPU, and the Python layer keeps the
reference's class/parameter names and refuses to run without the HIP path."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mpnn_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mpnn_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/mpnn_amd.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.mpnn_abi_version() == _lib.ABI_VERSION


def test_header_compiles_as_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "mpnn_amd.h"\nint main(void){return MPNN_ABI_VERSION == 7 ? 0 : 1;}\n')
    rc = os.system(f"gcc -std=c99 -Wall -Werror -I{ROOT}/include -c {src} -o {tmp_path}/t.o")
    assert rc == 0


def test_argument_validation_reports_errors_without_a_gpu():
    from mpnn_amd import _lib
    lib = _lib.load()
    # bad dtype / bad sizes are rejected before any HIP call
    rc = lib.mpnn_aggregate_fwd(None, 0, None, 0, None, 4, None, None, None, 1, None, 0, 0, None, 0, None, None,
                                0, None, None, None, 1, 64, 7, None)
    assert rc == -1 and b"dtype" in lib.mpnn_last_error()
    rc = lib.mpnn_aggregate_fwd(None, 0, None, 0, None, 99, None, None, None, 1, None, 0, 0, None, 0, None, None,
                                0, None, None, None, 1, 64, 0, None)
    assert rc == -1 and b"Fe" in lib.mpnn_last_error()
    rc = lib.mpnn_aggregate_fwd(None, 64, None, 64, None, 4, None, None, None, 1, None, 0, 0, None, 64, None, None,
                                0, None, None, None, 1, 64, 0, None)
    assert rc == -1 and b"null" in lib.mpnn_last_error()
    # row strides must fit 32 bits (the gather address is row * stride in one multiply-add); the pointers are never
    # followed: the call is rejected first
    fake = 1 << 20
    rc = lib.mpnn_aggregate_fwd(fake, 1 << 31, fake, 64, fake, 4, fake, fake, fake, 1, None, 0, 0, fake, 64, None, None,
                                0, None, None, None, 1, 64, 0, None)
    assert rc == -1 and b"above 2^31" in lib.mpnn_last_error()
    rc = lib.mpnn_gather_rows(None, None, None, 4, 3, None)
    assert rc == -1
    rc = lib.mpnn_csr_build(None, None, -1, 4, 4, None, None, None, None, None, 0, None)
    assert rc == -1
    assert lib.mpnn_bwd_scratch_floats(64, 4) == 2048 * 64 * 4
    assert lib.mpnn_mask_words(128, 0) == 4 and lib.mpnn_mask_words(64, 1) == 2      # bf16 rows of 8 / 16 lanes: a byte
    assert lib.mpnn_mask_words(128, 1) == 4 and lib.mpnn_mask_words(256, 1) == 8     # per lane; of <= 4 lanes: two
    assert lib.mpnn_mask_words(32, 1) == 4 and lib.mpnn_mask_words(56, 1) == 2       # 16-bit words per word
    assert lib.mpnn_mask_words(7, 0) == 0 and lib.mpnn_mask_words(256, 0) == 0
    with pytest.raises(_lib.MpnnError):
        _lib.check(rc, "mpnn_csr_build")


def test_layers_keep_reference_names_and_init():
    """Parameter names/shapes of reference models/layers.py:22-23,84-87 and models/mpnn.py:35,83,131;
    default nn.Linear initialisation in the reference's creation order."""
    from mpnn_amd import MPNN, MPNN_QM9, MPNN_ModelNet, MPNNLayerCat, MPNNLayerSum
    cat = MPNNLayerCat(11, 4, 64)
    assert {k: tuple(v.shape) for k, v in cat.state_dict().items()} == {
        "linear.weight": (64, 26), "linear_out.weight": (64, 75)}
    sm = MPNNLayerSum(11, 4, 64)
    assert {k: tuple(v.shape) for k, v in sm.state_dict().items()} == {
        "fused_linear.weight": (128, 11), "linear_3.weight": (64, 4), "linear_out.weight": (64, 75)}
    torch.manual_seed(5)
    a = MPNNLayerSum(11, 4, 64)
    torch.manual_seed(5)
    ref = [torch.nn.Linear(11, 128, bias=False), torch.nn.Linear(4, 64, bias=False),
           torch.nn.Linear(75, 64, bias=False)]
    for got, want in zip([a.fused_linear, a.linear_3, a.linear_out], ref):
        assert torch.equal(got.weight, want.weight)
    m = MPNN(500, 64, 7, num_layers=2, mode="sum")
    keys = set(m.state_dict())
    assert keys == {"layers.0.fused_linear.weight", "layers.0.linear_3.weight", "layers.0.linear_out.weight",
                    "layers.1.fused_linear.weight", "layers.1.linear_3.weight", "layers.1.linear_out.weight",
                    "output_layer.weight", "output_layer.bias"}
    assert m.layers[0].linear_3.weight.shape == (64, 4)      # MPNN hard-codes 4 edge channels
    q = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="cat")
    assert set(q.state_dict()) == {"layers.0.linear.weight", "layers.0.linear_out.weight", "layers.1.linear.weight",
                                   "layers.1.linear_out.weight", "output_layer.weight", "output_layer.bias"}
    assert isinstance(MPNN_ModelNet(3, 1, 64, 10), torch.nn.Module)
    import inspect
    assert list(inspect.signature(q.forward).parameters) == ["x", "edge_index", "batch", "edge_attr"]
    assert list(inspect.signature(m.forward).parameters) == ["x", "edge_index", "edge_attr"]


def test_product_path_has_no_cpu_fallback():
    from mpnn_amd import MPNNLayerSum
    from mpnn_amd._lib import MpnnError
    layer = MPNNLayerSum(4, 4, 8)
    with pytest.raises(MpnnError, match="GPU"):
        layer(torch.zeros(3, 4), torch.zeros((2, 2), dtype=torch.int64), torch.zeros(2, 4))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "mpnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "reference_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_drop_in_package_mirrors_the_reference_imports():
    """`integration/models` stands in for the reference's `models` package: the same three import forms the
    experiments and main.py use (models/__init__.py:1-2, experiments/*_exp.py) resolve to the HIP-backed classes."""
    import importlib
    import sys
    import mpnn_amd
    import mpnn_amd.datasets
    import mpnn_amd.train
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("models", "training", "data")}
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    try:
        models = importlib.import_module("models")
        assert models.MPNN is mpnn_amd.MPNN and models.MPNNLayerCat is mpnn_amd.MPNNLayerCat
        assert models.MPNNLayerSum is mpnn_amd.MPNNLayerSum
        mp = importlib.import_module("models.mpnn")
        assert mp.MPNN_QM9 is mpnn_amd.MPNN_QM9 and mp.MPNN_ModelNet is mpnn_amd.MPNN_ModelNet
        layers = importlib.import_module("models.layers")
        assert layers.MPNNLayerSum is mpnn_amd.MPNNLayerSum
        tr = importlib.import_module("training.train")            # experiments: `import training.train as train`
        for name in ("Trainer_QM9", "Trainer_Flickr", "Trainer_ModelNet", "Train_QM9", "Train_QM9_FB", "Train_Flickr",
                     "Train_ModelNet", "Train_ModelNet_FB"):
            assert getattr(tr, name) is getattr(mpnn_amd.train, name)
        ds = importlib.import_module("data.dataset")              # experiments: `from data.dataset import ...`
        for name in ("FlickrDataset", "QM9Dataset", "ModelNetDataset", "S3DISDataset"):
            assert getattr(ds, name) is getattr(mpnn_amd.datasets, name)
    finally:
        sys.path.remove(os.path.join(ROOT, "integration"))
        for k in [k for k in sys.modules if k.split(".")[0] in ("models", "training", "data")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_leading_dimension_of_single_row_matrices():
    """A one-row matrix keeps whatever stride(0) torch left it (1 after transposing a one-column matrix); the GEMM
    wrappers must pass a leading dimension that the C ABI's `ld >= K` check accepts."""
    from mpnn_amd.dense import _ld
    w = torch.zeros(6, 1).t().contiguous()          # [1, 6] with stride (1, 1)
    assert w.shape == (1, 6) and _ld(w) >= 6
    assert _ld(torch.zeros(5, 8)) == 8 and _ld(torch.zeros(5, 16)[:, :8]) == 16 and _ld(None) == 0
    assert _ld(torch.zeros(1, 16)[:, :8]) >= 8


def test_egnn_layer_rejects_widths_without_a_vector_layout():
    from mpnn_amd import EGNNLayer
    with pytest.raises(ValueError, match="multiple of 4"):
        EGNNLayer(8, 2, 38)
    assert EGNNLayer(8, 2, 40).coord_linear.weight.shape == (1, 40)
