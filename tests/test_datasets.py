"""Dataset stand-ins keep the reference's method names (data/dataset.py) and feed its training loops."""
import pytest
import torch
import torch.nn.functional as F


def test_wrappers_expose_the_reference_method_names():
    import inspect
    from mpnn_amd import datasets
    assert {"get_data", "get_split"} <= set(dir(datasets.FlickrDataset))
    assert {"get_loader", "get_full_graphs", "get_data"} <= set(dir(datasets.QM9Dataset))
    assert {"get_loaders", "get_full_graphs"} <= set(dir(datasets.ModelNetDataset))
    assert {"get_train_data", "get_test_data"} <= set(dir(datasets.S3DISDataset))
    assert list(inspect.signature(datasets.S3DISDataset.__init__).parameters)[:4] == ["self", "root", "transform", "test_area"]
    assert list(inspect.signature(datasets.QM9Dataset.get_loader).parameters)[:3] == ["self", "batch_size", "shuffle"]
    assert list(inspect.signature(datasets.ModelNetDataset.get_loaders).parameters) == ["self", "batch_size", "shuffle"]
    assert datasets._split_sizes(1001) == (800, 100, 101)


@pytest.mark.gpu
def test_qm9_training_loop_as_the_reference_writes_it():
    """training/train.py:85-98 (Train_QM9) with the stand-in dataset, the drop-in trainer and model."""
    from mpnn_amd import MPNN_QM9
    from mpnn_amd.datasets import QM9Dataset
    from mpnn_amd.optim import FlatAdam
    from mpnn_amd.train import Trainer_QM9
    ds = QM9Dataset(num_graphs=300)
    train_loader, val_loader, test_loader = ds.get_loader(batch_size=64)
    assert (len(train_loader), len(val_loader), len(test_loader)) == (4, 1, 1)
    torch.manual_seed(0)
    model = MPNN_QM9(11, 4, 32, 19, num_layers=2, mode="sum").cuda()
    trainer = Trainer_QM9(model, FlatAdam(model.parameters(), lr=1e-2), F.mse_loss, "cuda")
    first = last = None
    for epoch in range(3):
        loss = 0.0
        for data in train_loader:
            data.to("cuda")
            loss += trainer.train(data)
        first = loss if first is None else first
        last = loss
    assert last == last and last < first * 1.5          # finite, not diverging
    test_loss = sum(trainer.test(data) for data in test_loader)
    assert test_loss == test_loss
    full = ds.get_full_graphs()
    assert [b.num_graphs for b in full] == [240, 30, 30]
    assert trainer.test(full[2]) == trainer.test(full[2])


@pytest.mark.gpu
def test_modelnet_and_flickr_stand_ins():
    from mpnn_amd import MPNN, MPNN_ModelNet
    from mpnn_amd.datasets import FlickrDataset, ModelNetDataset
    from mpnn_amd.optim import FlatAdam
    from mpnn_amd.train import Trainer_Flickr, Trainer_ModelNet
    ds = ModelNetDataset(train_clouds=12, test_clouds=4, points=128, k=8)
    train_loader, test_loader = ds.get_loaders(batch_size=4)
    model = MPNN_ModelNet(3, 1, 32, 10, num_layers=2, mode="cat").cuda()
    trainer = Trainer_ModelNet(model, FlatAdam(model.parameters(), lr=1e-2), F.cross_entropy, "cuda")
    losses = [trainer.train(data) for data in train_loader]
    assert len(losses) == 3 and all(l == l for l in losses)
    assert all(t == t for t in (trainer.test(d) for d in test_loader))
    data = ds.get_full_graphs()[0]
    assert data.num_graphs == 12 and data.edge_index.shape == (2, 12 * 128 * 8) and data.edge_attr.shape == (12 * 128 * 8, 1)

    fl = FlickrDataset(num_nodes=2000, num_edges=12000, num_features=40)
    d = fl.get_data()
    tr, va, te = fl.get_split()
    assert int((tr | va | te).sum()) == 2000 and int((tr & va).sum()) == 0 and int((va & te).sum()) == 0
    assert d.edge_attr.shape == (12000, 4) and bool((d.edge_attr == 1).all())
    m = MPNN(40, 32, 7, num_layers=2, mode="sum").cuda()
    t = Trainer_Flickr(m, FlatAdam(m.parameters(), lr=1e-2), F.cross_entropy, "cuda")
    l0 = t.train(d)
    for _ in range(5):
        l1 = t.train(d)
    assert l1 < l0 and 0.0 <= t.test(d) <= 1.0


@pytest.mark.gpu
def test_s3dis_stand_in_feeds_a_layer():
    from mpnn_amd import MPNNLayerSum
    from mpnn_amd.datasets import S3DISDataset
    ds = S3DISDataset(train_blocks=3, test_blocks=2, points=256, k=8)
    train, test = ds.get_train_data(), ds.get_test_data()
    assert train.num_graphs == 3 and test.num_graphs == 2 and ds.get_train_data() is train
    assert train.x.shape == (3 * 256, 6) and train.point_y.shape == (3 * 256,) and int(train.point_y.max()) < 13
    batch = train.collate([2, 0])
    assert batch.x.shape == (512, 6) and batch.edge_index.shape == (2, 512 * 8) and batch.edge_attr.shape == (512 * 8, 1)
    layer = MPNNLayerSum(6, 1, 16).cuda()
    out = layer(batch.x, batch.edge_index, batch.edge_attr)
    assert out.shape == (512, 16) and bool(torch.isfinite(out).all())


@pytest.mark.gpu
def test_epoch_loops_with_the_reference_signatures(capsys):
    """Train_QM9 / _FB, Train_Flickr, Train_ModelNet / _FB (training/train.py:88-195) on the stand-in datasets."""
    import inspect
    from mpnn_amd import MPNN, MPNN_ModelNet, MPNN_QM9, train
    from mpnn_amd.datasets import FlickrDataset, ModelNetDataset, QM9Dataset
    from mpnn_amd.optim import FlatAdam
    for name in ("Train_QM9", "Train_QM9_FB", "Train_Flickr", "Train_ModelNet", "Train_ModelNet_FB"):
        assert list(inspect.signature(getattr(train, name)).parameters)[2:] == ["optimizer", "loss_fn", "device", "epochs", "benchmark"]
    qm9 = QM9Dataset(num_graphs=120)
    model = MPNN_QM9(11, 4, 16, 19, num_layers=2, mode="sum").cuda()
    opt = FlatAdam(model.parameters(), lr=1e-2)
    train.Train_QM9(model, qm9.get_loader(batch_size=32), opt, F.mse_loss, "cuda", epochs=2)
    train.Train_QM9_FB(model, qm9.get_full_graphs(), opt, F.mse_loss, "cuda", epochs=2, benchmark=True)
    out = capsys.readouterr().out
    assert out.count("| Test Loss: ") == 2 and "Epoch 002 | Loss: " in out
    assert "Training Time: " in out and "epoch/sec" in out and "Validation Loss: " in out
    flickr = FlickrDataset(num_nodes=2000, num_edges=12000, num_features=20)
    fm = MPNN(20, 16, 7, num_layers=2, mode="cat").cuda()
    train.Train_Flickr(fm, flickr.get_data(), FlatAdam(fm.parameters(), lr=1e-2), F.cross_entropy, "cuda", epochs=2)
    assert capsys.readouterr().out.count("| Test Acc: ") == 2
    mn = ModelNetDataset(train_clouds=8, test_clouds=4, points=64, k=6)
    mm = MPNN_ModelNet(3, 1, 16, 10, num_layers=2, mode="sum").cuda()
    mo = FlatAdam(mm.parameters(), lr=1e-2)
    train.Train_ModelNet(mm, mn.get_loaders(batch_size=4), mo, F.cross_entropy, "cuda", epochs=1, benchmark=True)
    train.Train_ModelNet_FB(mm, mn.get_full_graphs(), mo, F.cross_entropy, "cuda", epochs=2)
    out = capsys.readouterr().out
    assert "Validation Loss: " in out and out.count("| Test Loss: ") == 2


@pytest.mark.gpu
def test_backward_on_the_calling_thread_gives_the_same_gradients():
    """mpnn_amd.train.backward_on_caller (what the trainers wrap their loss.backward() in) only changes WHERE the
    backward runs: every gradient is bit for bit the one autograd's device thread produces."""
    from mpnn_amd import MPNN_QM9, synth
    from mpnn_amd.train import backward_on_caller
    d = synth.qm9(torch.device("cuda:0"), 3, 64)
    torch.manual_seed(0)
    model = MPNN_QM9(11, 4, 64, 19, num_layers=2, mode="sum").cuda()

    def grads(scoped):
        model.zero_grad()
        loss = F.mse_loss(model(d["x"], d["edge_index"], d["batch"], d["edge_attr"])[0], d["y"][0])
        if scoped:
            with backward_on_caller():
                assert not torch.autograd.is_multithreading_enabled()
                loss.backward()
        else:
            loss.backward()
        return [p.grad.clone() for p in model.parameters()]

    there, here = grads(False), grads(True)
    assert torch.autograd.is_multithreading_enabled()          # scoped: back on after the with
    assert all(torch.equal(a, b) for a, b in zip(there, here))
