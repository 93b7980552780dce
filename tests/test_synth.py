"""CPU tests of the synthetic workload generators (shapes, dtypes, conventions, determinism)."""
import torch

from mpnn_amd import synth


def _check_edge_index(ei, n):
    assert ei.dtype == torch.int64 and ei.dim() == 2 and ei.size(0) == 2
    assert int(ei.min()) >= 0 and int(ei.max()) < n


def test_flickr_shape():
    d = synth.flickr("cpu", 0, num_nodes=2000, num_edges=20000, num_features=50)
    _check_edge_index(d["edge_index"], 2000)
    assert d["edge_index"].size(1) == 20000
    assert bool((d["edge_index"][0] != d["edge_index"][1]).all())          # no self loops
    assert torch.equal(d["edge_attr"], torch.ones(20000, 4))               # reference data/dataset.py:37
    assert torch.allclose(d["x"].sum(1), torch.ones(2000), atol=1e-5)      # NormalizeFeatures
    # stored in both directions
    fwd = set(map(tuple, d["edge_index"].t().tolist()))
    assert all((b, a) in fwd for a, b in list(fwd)[:200])
    assert torch.equal(synth.flickr("cpu", 0, 2000, 20000, 50)["edge_index"], d["edge_index"])


def test_qm9_shape():
    d = synth.qm9("cpu", 0, num_graphs=128)
    n = d["num_nodes"]
    _check_edge_index(d["edge_index"], n)
    assert 128 * 12 < n < 128 * 24
    e = d["edge_index"].size(1)
    assert 1.8 * n < e < 2.4 * n                                           # ~38 directed bonds per ~18 atoms
    b = d["batch"]
    assert bool((b[1:] >= b[:-1]).all()) and int(b.max()) == 127           # sorted, as PyG batches are
    assert bool((b[d["edge_index"][0]] == b[d["edge_index"][1]]).all())    # no bond crosses molecules
    assert d["edge_attr"].shape == (e, 4) and torch.equal(d["edge_attr"].sum(1), torch.ones(e))
    assert d["x"].shape == (n, 11) and d["y"].shape == (128, 19) and d["pos"].shape == (n, 3)


def test_modelnet_shape():
    d = synth.modelnet("cpu", 0, num_clouds=2, points=64, k=5)
    _check_edge_index(d["edge_index"], 128)
    assert d["edge_index"].size(1) == 2 * 64 * 5
    deg = torch.bincount(d["edge_index"][1], minlength=128)
    assert bool((deg == 5).all())                                          # every target has exactly k neighbours
    assert d["edge_attr"].shape == (640, 1) and float(d["edge_attr"].max()) == 1.0
    src, dst = d["edge_index"]
    dist = (d["pos"][src] - d["pos"][dst]).norm(dim=1)
    assert torch.allclose(dist / dist.max(), d["edge_attr"].view(-1), atol=1e-6)


def test_powerlaw_shape():
    d = synth.powerlaw("cpu", 0, num_nodes=20000, num_edges=500000, num_features=8, max_degree=5000)
    _check_edge_index(d["edge_index"], 20000)
    deg = torch.bincount(d["edge_index"][1], minlength=20000)
    assert 3500 < int(deg.max()) < 6500                                    # expected maximum 5000
    assert float(deg.float().median()) < 25 < float(deg.float().mean()) + 1  # heavy tail: median below the mean
    assert d["x"].shape == (20000, 8) and d["edge_attr"].shape == (500000, 4)


def test_powerlaw_shares_are_the_whole_graphs_values(monkeypatch):
    """A rank of a partitioned run draws only its rows of x and its edges' attributes: the same values the whole
    graph holds there, whatever the block boundaries."""
    import torch
    for block in (1 << 20, 1000):
        monkeypatch.setattr(synth, "_BLOCK", block)
        d = synth.powerlaw("cpu", 3, 5000, 30000, 16, 4)
        assert torch.equal(d["x"][1200:3400], synth.powerlaw_features("cpu", 3, 5000, 16, 1200, 3400))
        ids = torch.randperm(30000, generator=torch.Generator().manual_seed(1))[:7000]
        assert torch.equal(d["edge_attr"][ids], synth.powerlaw_edge_attr("cpu", 3, 30000, 4, ids))
    assert synth.powerlaw_features("cpu", 3, 5000, 16, 100, 100).shape == (0, 16)
    assert "x" not in synth.powerlaw("cpu", 3, 500, 3000, 16, 4, with_features=False)
