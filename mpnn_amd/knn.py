"""k-NN edges and the normalised-distance edge feature of point-cloud batches, built on the device.

Stands where the reference prepares point-cloud graphs on the CPU in its dataset transforms
(experiments/modelnet_exp.py:58 ``T.Distance(norm=True)``, data/dataset.py:193).
"""
from __future__ import annotations

import torch

from . import _lib


def knn_graph(pos: torch.Tensor, num_clouds: int, k: int):
    """pos [num_clouds * P, 3] float32 -> (edge_index int64 [2, num_clouds * P * k], edge_attr float32 [E, 1]).

    Edges j -> i for the k nearest j of every point i within its cloud, grouped by i in ascending distance
    (ties: lower j); edge_attr = distance / largest edge distance of the cloud."""
    _lib.require_device(pos, "pos")
    if pos.dtype != torch.float32 or pos.dim() != 2 or pos.size(1) != 3:
        raise TypeError("pos must be a float32 [N, 3] matrix")
    if num_clouds < 1 or pos.size(0) % num_clouds:
        raise ValueError("pos must hold num_clouds clouds of equal size")
    points = pos.size(0) // num_clouds
    lib = _lib.load()
    pos = pos.contiguous()
    e = pos.size(0) * k
    edge_index = torch.empty((2, e), dtype=torch.int64, device=pos.device)
    edge_attr = torch.empty((e, 1), dtype=torch.float32, device=pos.device)
    ws = torch.empty(max(lib.mpnn_knn_workspace_floats(num_clouds, points, k), 1), dtype=torch.float32, device=pos.device)
    _lib.check(lib.mpnn_knn_graph(_lib.ptr(pos), num_clouds, points, k, _lib.ptr(edge_index), _lib.ptr(edge_attr),
                                  _lib.ptr(ws), _lib.stream_handle(pos.device)), "mpnn_knn_graph")
    return edge_index, edge_attr
