"""Model stacks with the reference's class names, constructor and forward signatures
(reference models/mpnn.py:8-55 MPNN_QM9, :57-101 MPNN, :104-151 MPNN_ModelNet).

Note the reference's argument order for the pooled models: forward(x, edge_index, batch, edge_attr).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.nn import Linear

from .layers import EGNNLayer, MPNNLayerCat, MPNNLayerSum
from . import compiled_ops
from .dense import head
from .ops import global_mean_pool


def _relu_after(layer, x):
    """x = F.relu(x) after a layer (models/mpnn.py:48,98,144).  Both layer classes end in relu(linear_out(..)), and
    relu of a relu output is the identity in the forward (x >= 0) and in the backward (the layer applies the same
    mask [x > 0] to the incoming gradient), so the two elementwise passes over [N, H] are skipped for them - the
    results are bit for bit those of the literal call."""
    return x if isinstance(layer, (MPNNLayerCat, MPNNLayerSum)) else F.relu(x)


def _make_layers(in_channels, edge_channels, hidden_channels, num_layers, mode):
    cls = MPNNLayerCat if mode == 'cat' else MPNNLayerSum
    layers = torch.nn.ModuleList()
    layers.append(cls(in_channels, edge_channels, hidden_channels))
    for _ in range(num_layers - 1):
        layers.append(cls(hidden_channels, edge_channels, hidden_channels))
    return layers


class MPNN(torch.nn.Module):
    """Node-level model (Flickr); edge_channels is fixed to 4 as in the reference (models/mpnn.py:72-81)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, mode="cat"):
        super().__init__()
        self.layers = _make_layers(in_channels, 4, hidden_channels, num_layers, mode)
        self.output_layer = Linear(hidden_channels, out_channels)

    def forward(self, x, edge_index, edge_attr):
        for layer in self.layers:
            x = _relu_after(layer, layer(x, edge_index, edge_attr))
        return head(x, self.output_layer)


class _PooledMPNN(torch.nn.Module):
    def __init__(self, in_channels, edge_channels, hidden_channels, out_channels, num_layers=2, mode='cat'):
        super().__init__()
        print('Using cat' if mode == 'cat' else 'Using sum')  # models/mpnn.py:23,26,119,122
        self.layers = _make_layers(in_channels, edge_channels, hidden_channels, num_layers, mode)
        self.output_layer = Linear(hidden_channels, out_channels)

    def forward(self, x, edge_index, batch, edge_attr):
        for layer in self.layers:
            x = _relu_after(layer, layer(x, edge_index, edge_attr))
        # while the model is being compiled (torch.compile(model), experiments/qm9_exp.py:34) the pooling is one
        # registered operator; eager calls take the same kernels directly
        x = torch.ops.mpnn_amd.mean_pool(x, batch) if torch.compiler.is_compiling() else global_mean_pool(x, batch)
        return self.output_layer(x)


class MPNN_QM9(_PooledMPNN):
    pass


class MPNN_ModelNet(_PooledMPNN):
    pass


class EGNN(torch.nn.Module):
    """Graph-level model on EGNN layers, the shape of MPNN_QM9 / MPNN_ModelNet (models/mpnn.py:8-55, 104-151: layers ->
    relu -> global_mean_pool -> Linear head) with the coordinates threaded through the layers.  The reference has no
    EGNN; constructor arguments follow MPNN_QM9's, forward takes ``pos`` after ``x``:
    forward(x, pos, edge_index, batch, edge_attr) -> logits [num_graphs, out_channels]
    (``return_pos=True``: also the coordinates after the last layer, float32 [N, 3])."""

    def __init__(self, in_channels, edge_channels, hidden_channels, out_channels, num_layers=2):
        super().__init__()
        self.layers = torch.nn.ModuleList(
            [EGNNLayer(in_channels if k == 0 else hidden_channels, edge_channels, hidden_channels)
             for k in range(num_layers)])
        self.output_layer = Linear(hidden_channels, out_channels)

    def forward(self, x, pos, edge_index, batch, edge_attr, return_pos=False):
        for layer in self.layers:
            x, pos = layer(x, pos, edge_index, edge_attr)   # ends in a relu: the model loop's F.relu is the identity
        out = self.output_layer(global_mean_pool(x, batch))
        return (out, pos) if return_pos else out
