"""Training steps with the reference's trainer interface (training/train.py:11-82).

``Trainer_QM9``, ``Trainer_Flickr`` and ``Trainer_ModelNet`` take (model, optimizer, loss_fn, device) and offer
``train(data) -> float`` and ``test(data) -> float`` exactly like the reference's classes, including its choice
of scoring only graph 0 of a batch (``loss_fn(out[0], data.y[0])``, training/train.py:22,74).  ``data`` is
anything with the attributes the reference reads (x / pos, edge_index, batch, edge_attr, y, train_mask,
test_mask): a ``mpnn_amd.data.Batch`` or a plain namespace.

``Train_QM9``, ``Train_QM9_FB``, ``Train_Flickr``, ``Train_ModelNet`` and ``Train_ModelNet_FB`` are the reference's
epoch loops (training/train.py:88-195) with their signatures: mini-batch loaders or one full batch per split,
per-epoch report lines, and with ``benchmark=True`` only the timing summary plus one evaluation at the end.

``CapturedStep`` records one whole training step (forward, loss, backward, optimizer) into a HIP graph when the
step's tensors keep their addresses and shapes - a full-graph trainer, or fixed-size point-cloud batches written
into the same buffers - and replays it with no per-launch host work.
"""
from __future__ import annotations

import time

import torch
import torch.nn.functional as F

from .loss import masked_cross_entropy


def backward_on_caller():
    """Context in which ``loss.backward()`` runs on the calling thread instead of autograd's device thread.

    A training step on a mini-batch of molecules or point clouds is a handful of short launches; handing the
    backward to another thread and waiting for it (two wake-ups, the interpreter lock changing hands) costs more
    than the launches themselves: 0.13-0.18 -> 0.09 ms for one layer step on 1024 molecules, which is then bound
    by the GPU again.  The trainers below use it around their ``loss.backward()``; a hand-written loop wraps its
    own.  Scoped and thread-local - nothing is changed for code outside the ``with``."""
    return torch.autograd.set_multithreading_enabled(False)


def _dev(t, device):
    return t if t.device == torch.device(device) else t.to(device)


class Trainer_QM9:
    def __init__(self, model, optimizer, loss_fn, device):
        self.model, self.optimizer, self.loss_fn, self.device = model, optimizer, loss_fn, device

    def _forward(self, data):
        d = self.device
        return self.model(_dev(data.x, d), _dev(data.edge_index, d), _dev(data.batch, d), _dev(data.edge_attr, d))

    def train(self, data):
        self.model.train()
        self.optimizer.zero_grad()
        out = self._forward(data)
        loss = self.loss_fn(out[0], _dev(data.y, self.device)[0])
        with backward_on_caller():
            loss.backward()
        self.optimizer.step()
        return loss.item()

    def test(self, data):
        self.model.eval()
        with torch.no_grad():
            out = self._forward(data)
            return self.loss_fn(out[0], _dev(data.y, self.device)[0]).item()


class Trainer_ModelNet(Trainer_QM9):
    """Same step with the point coordinates as node features (training/train.py:73)."""

    def _forward(self, data):
        d = self.device
        return self.model(_dev(data.pos, d), _dev(data.edge_index, d), _dev(data.batch, d), _dev(data.edge_attr, d))


class Trainer_Flickr:
    """Transductive full-graph trainer (training/train.py:33-58).  With ``F.cross_entropy`` as the loss the masked
    mean is computed by the fused kernel (same value, no row compaction); any other loss takes the reference's
    indexing path."""

    def __init__(self, model, optimizer, loss_fn, device):
        self.model, self.optimizer, self.loss_fn, self.device = model, optimizer, loss_fn, device

    def loss(self, out, data):
        y, mask = _dev(data.y, self.device), _dev(data.train_mask, self.device)
        if self.loss_fn is F.cross_entropy and out.is_cuda and out.dtype == torch.float32:
            return masked_cross_entropy(out, y, mask)
        return self.loss_fn(out[mask], y[mask])

    def train(self, data):
        self.model.train()
        self.optimizer.zero_grad()
        d = self.device
        out = self.model(_dev(data.x, d), _dev(data.edge_index, d), _dev(data.edge_attr, d))
        loss = self.loss(out, data)
        with backward_on_caller():
            loss.backward()
        self.optimizer.step()
        return loss.item()

    def test(self, data):
        """Accuracy on test_mask.  (The reference's test() omits edge_attr from the model call,
        training/train.py:54, which its own MPNN.forward does not accept; it is passed here.)"""
        self.model.eval()
        with torch.no_grad():
            d = self.device
            out = self.model(_dev(data.x, d), _dev(data.edge_index, d), _dev(data.edge_attr, d))
            pred = out.argmax(dim=1)
            mask, y = _dev(data.test_mask, d), _dev(data.y, d)
            return ((pred == y) & mask).sum().item() / mask.sum().item()


class CapturedStep:
    """One training step as a HIP graph.

    ``step_fn()`` must run the whole step on inputs and parameters that stay where they are (zero_grad, forward,
    loss, backward, optimizer step without host reads - e.g. FlatAdam; tensors created inside the step, gradients
    included, come from the graph's own memory pool and keep their addresses from replay to replay) and return the
    loss tensor.  ``replay()`` relaunches it and returns that same (device) loss tensor."""

    def __init__(self, step_fn, warmup: int = 3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step_fn()

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss


class CapturedBatchStep:
    """Mini-batch training where every full batch is one HIP-graph replay.

    The graphs of a batch are written into ONE padded batch of fixed capacity (``GraphStore.collate(capacity=...)``:
    molecules of any size fit, the real sizes travel on the device), and the step on that batch - zero_grad,
    forward, ``loss_of(out, batch)``, backward, optimizer step - is recorded once.  ``train(ids)`` then costs the
    slot-table copy, the two assembly kernels and one graph launch, whatever the number of kernels in the step.
    ``loss_of`` defaults to the reference's choice, ``loss_fn(out[0], y[0])`` (training/train.py:22,74).  The model
    output carries one extra last row (the padding graph); ``loss_of`` must not use it.  A batch with another
    number of graphs (the short last batch of an epoch) runs eagerly through the same functions.
    The optimizer must not read the device (``mpnn_amd.optim.FlatAdam``)."""

    def __init__(self, model, optimizer, loss_fn, store, batch_size: int, features: str = "x", loss_of=None,
                 call=None):
        self.model, self.optimizer, self.loss_fn, self.store = model, optimizer, loss_fn, store
        self.batch_size, self.features = int(batch_size), features
        self.loss_of = loss_of or (lambda out, b: loss_fn(out[0], b.y[0]))
        # how the model takes a batch: the pooled models' forward(x, edge_index, batch, edge_attr) (models/mpnn.py:38)
        # unless told otherwise, e.g. call=lambda m, b: m(b.x, b.pos, b.edge_index, b.batch, b.edge_attr) for EGNN
        self.call = call or (lambda m, b: m(getattr(b, self.features), b.edge_index, b.batch, b.edge_attr))
        if not (hasattr(optimizer, "state_snapshot") and hasattr(optimizer, "state_restore")):
            raise TypeError("CapturedBatchStep runs warm-up steps before recording and must put the optimizer back: "
                            "it needs state_snapshot() / state_restore() (mpnn_amd.optim.FlatAdam has them)")
        self.capacity = store.capacity_for(self.batch_size)
        self.batch = None
        self.captured = None
        self.graph = None      # the MPGraph whose scratch buffers the recorded launches write to

    def _step(self, b):
        self.optimizer.zero_grad()
        out = self.call(self.model, b)
        loss = self.loss_of(out, b)
        loss.backward()
        self.optimizer.step()
        return loss

    def train(self, ids) -> torch.Tensor:
        """One step on the graphs ``ids``; returns the loss as a device tensor (no host read)."""
        self.model.train()
        if len(ids) != self.batch_size:
            return self._step(self.store.collate(ids))
        if self.batch is None:
            self.batch = self.store.collate(ids, capacity=self.capacity)
            # the warm-up steps of the capture would train on this batch: record on a copy of the state
            state = [p.detach().clone() for p in self.model.parameters()]
            opt_state = self.optimizer.state_snapshot()
            self.captured = CapturedStep(lambda: self._step(self.batch), warmup=2)
            self.graph = self.batch._graph       # pinned here, not only through the graph cache's 8 entries
            with torch.no_grad():
                for p, s0 in zip(self.model.parameters(), state):
                    p.copy_(s0)
            self.optimizer.state_restore(opt_state)
        else:
            self.store.collate(ids, out=self.batch)
        return self.captured.replay()


def _batches(data, device):
    """A loader yields batches; a single full batch (the reference's _FB variants) is its own one-element epoch."""
    for b in (data if hasattr(data, "__iter__") and not hasattr(data, "edge_index") else (data,)):
        b.to(device)
        yield b


def _epochs(trainer, train_data, report_data, final_data, device, epochs, benchmark, report_name, final_name):
    """The loop all five reference functions share: train on every batch of ``train_data`` per epoch; without
    ``benchmark`` evaluate ``report_data`` after each epoch and print one line; with it, time the epochs, print
    the time and epochs per second, then evaluate ``final_data`` once."""
    start = time.time()
    for epoch in range(epochs):
        loss = sum(trainer.train(b) for b in _batches(train_data, device))
        if not benchmark and report_data is not None:
            value = sum(trainer.test(b) for b in _batches(report_data, device))
            print(f"Epoch {epoch + 1:03d} | Loss: {loss:.4f} | {report_name}: {value:.4f}")
    if benchmark:
        elapsed = time.time() - start
        print(f"Training Time: {elapsed:.4f} seconds")
        print(f"Troughput: {epochs / elapsed:.3f} epoch/sec")     # the reference's spelling (training/train.py:109)
        if final_data is not None:
            value = sum(trainer.test(b) for b in _batches(final_data, device))
            print(f"{final_name}: {value:.4f}")


def Train_QM9(model, data_loader, optimizer, loss_fn, device, epochs=3, benchmark=False):
    """training/train.py:88-114: (train, val, test) loaders; test loss per epoch, validation loss after a benchmark."""
    train_loader, val_loader, test_loader = data_loader
    _epochs(Trainer_QM9(model, optimizer, loss_fn, device), train_loader, test_loader, val_loader, device, epochs,
            benchmark, "Test Loss", "Validation Loss")


def Train_QM9_FB(model, data_set, optimizer, loss_fn, device, epochs=3, benchmark=False):
    """training/train.py:116-136: one full batch per split."""
    train_set, val_set, test_set = data_set
    _epochs(Trainer_QM9(model, optimizer, loss_fn, device), train_set, test_set, val_set, device, epochs, benchmark,
            "Test Loss", "Validation Loss")


def Train_Flickr(model, data, optimizer, loss_fn, device, epochs=3, benchmark=False):
    """training/train.py:138-152: the whole graph every epoch; accuracy on the test mask per epoch."""
    trainer = Trainer_Flickr(model, optimizer, loss_fn, device)
    start = time.time()
    for epoch in range(epochs):
        loss = trainer.train(data)
        if not benchmark:
            print(f"Epoch {epoch + 1:03d} | Loss: {loss:.4f} | Test Acc: {trainer.test(data):.4f}")
    if benchmark:
        elapsed = time.time() - start
        print(f"Training Time: {elapsed:.4f} seconds")
        print(f"Troughput: {epochs / elapsed:.3f} epoch/sec")


def Train_ModelNet(model, data_loader, optimizer, loss_fn, device, epochs=3, benchmark=False):
    """training/train.py:155-176: (train, test) loaders; no per-epoch report, test loss after a benchmark."""
    train_loader, test_loader = data_loader
    _epochs(Trainer_ModelNet(model, optimizer, loss_fn, device), train_loader, None, test_loader, device, epochs,
            benchmark, "", "Validation Loss")


def Train_ModelNet_FB(model, data_loader, optimizer, loss_fn, device, epochs=3, benchmark=False):
    """training/train.py:178-195: one full batch per split; test loss per epoch."""
    train_set, test_set = data_loader
    _epochs(Trainer_ModelNet(model, optimizer, loss_fn, device), train_set, test_set, None, device, epochs,
            benchmark, "Test Loss", "")
