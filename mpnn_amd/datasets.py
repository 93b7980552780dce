"""Offline stand-ins for the reference's dataset wrappers (data/dataset.py), same class and method names.

The reference downloads Flickr / QM9 / ModelNet through torch_geometric on first use; there is no network here,
so these classes serve SYNTHETIC data of the same shapes (mpnn_amd.synth) from GPU memory through the same
methods: ``FlickrDataset.get_data/get_split``, ``QM9Dataset.get_loader/get_full_graphs``,
``ModelNetDataset.get_loaders/get_full_graphs``, ``S3DISDataset.get_train_data/get_test_data``.  Loaders are GraphLoaders (device-side batch assembly); the
full-graph variants return one Batch holding every graph of a split, as ``Batch.from_data_list`` does in the
reference (data/dataset.py:120-124, 225-227).
"""
from __future__ import annotations

import types

import numpy as np
import torch

from . import synth
from .data import GraphLoader, GraphStore
from .knn import knn_graph


class FlickrDataset:
    """data/dataset.py:16-48: one large graph, edge_attr = ones(E, 4), train / val / test masks."""

    def __init__(self, root: str = "data/Flickr", transform=None, device="cuda", seed: int = 0, **shape):
        d = synth.flickr(device, seed, **shape)
        g = torch.Generator(device=device).manual_seed(seed + 1)
        rest = ~d["train_mask"]
        coin = torch.rand(d["num_nodes"], generator=g, device=device) < 0.5
        d["val_mask"], d["test_mask"] = rest & coin, rest & ~coin
        self._data = types.SimpleNamespace(**d)

    def get_data(self):
        return self._data

    def get_split(self):
        d = self._data
        return d.train_mask, d.val_mask, d.test_mask


def _split_sizes(n_total: int):
    n_train, n_val = int(0.8 * n_total), int(0.1 * n_total)     # data/dataset.py:74-81
    return n_train, n_val, n_total - n_train - n_val


class QM9Dataset:
    """data/dataset.py:50-145: molecules with 11 node features, 4 bond types, 19 targets."""

    def __init__(self, root: str = "data/QM9", transform=None, device="cuda", seed: int = 0, num_graphs: int = 4096):
        self.store = GraphStore.from_batched(synth.qm9(device, seed, num_graphs))
        self._seed = seed

    def __len__(self):
        return self.store.num_graphs

    def get_loader(self, batch_size: int = 32, shuffle: bool = True, **kwargs):
        """(train_loader, val_loader, test_loader) over a random 80/10/10 split (random_split in the reference,
        data/dataset.py:84-86); only the training loader shuffles (data/dataset.py:94-96)."""
        n_train, n_val, _ = _split_sizes(len(self))
        perm = np.random.default_rng(self._seed).permutation(len(self))
        parts = perm[:n_train], perm[n_train:n_train + n_val], perm[n_train + n_val:]
        return (GraphLoader(self.store, batch_size, shuffle=True, seed=self._seed, indices=parts[0]),
                GraphLoader(self.store, batch_size, shuffle=False, indices=parts[1]),
                GraphLoader(self.store, batch_size, shuffle=False, indices=parts[2]))

    def get_full_graphs(self):
        """One batch per split, graphs in dataset order (slices, data/dataset.py:115-117)."""
        n_train, n_val, _ = _split_sizes(len(self))
        ids = np.arange(len(self))
        return (self.store.collate(ids[:n_train]), self.store.collate(ids[n_train:n_train + n_val]),
                self.store.collate(ids[n_train + n_val:]))

    def get_data(self):
        return self.store


class ModelNetDataset:
    """data/dataset.py:180-229: point clouds; here ``points`` sampled points per cloud joined to their k nearest
    neighbours on the device, with the normalised distance as the edge feature (experiments/modelnet_exp.py:58)."""

    def __init__(self, root: str = "data/ModelNet", name: str = "10", transform=None, device="cuda", seed: int = 0,
                 train_clouds: int = 256, test_clouds: int = 64, points: int = 1024, k: int = 20):
        self.train_store = self._make(device, seed, train_clouds, points, k, int(name))
        self.test_store = self._make(device, seed + 1, test_clouds, points, k, int(name))

    @staticmethod
    def _make(device, seed, clouds, points, k, classes):
        g = torch.Generator(device=device).manual_seed(seed)
        pos = torch.rand(clouds * points, 3, generator=g, device=device) * 2 - 1
        edge_index, edge_attr = knn_graph(pos, clouds, k)
        y = torch.randint(0, classes, (clouds,), generator=g, device=device)
        node_ptr = np.arange(clouds + 1, dtype=np.int64) * points
        edge_ptr = np.arange(clouds + 1, dtype=np.int64) * points * k
        return GraphStore(pos, edge_index, edge_attr, node_ptr, edge_ptr, y=y, pos=pos)

    def get_loaders(self, batch_size: int = 32, shuffle: bool = True):
        return (GraphLoader(self.train_store, batch_size, shuffle=shuffle),
                GraphLoader(self.test_store, batch_size, shuffle=False))

    def get_full_graphs(self):
        return (self.train_store.collate(np.arange(self.train_store.num_graphs)),
                self.test_store.collate(np.arange(self.test_store.num_graphs)))


class S3DISDataset:
    """data/dataset.py:147-178: indoor scans cut into blocks of ``points`` points with 6 features (colour and
    normalised position) and one of 13 classes per point; ``get_train_data`` / ``get_test_data`` return the blocks
    of the training areas / of ``test_area``.  The reference's experiment for it is a stub
    (experiments/s3dis_exp.py:13-22) and the blocks carry no edges there; here every block comes with its k-NN
    graph and normalised-distance edge feature, ready for the layers.  Per-point labels are ``store.point_y``; the
    per-block ``y`` is the block's most frequent class."""

    NUM_CLASSES = 13

    def __init__(self, root: str = "data/S3DIS", transform=None, test_area=1, device="cuda", seed: int = 0,
                 train_blocks: int = 64, test_blocks: int = 16, points: int = 4096, k: int = 16):
        self.test_area = test_area
        self._args = (device, seed, points, k)
        self._sizes = {"train": train_blocks, "test": test_blocks}
        self._stores = {}

    def _make(self, split: str) -> GraphStore:
        if split not in self._stores:
            device, seed, points, k = self._args
            blocks = self._sizes[split]
            g = torch.Generator(device=device).manual_seed(seed + (0 if split == "train" else 1000 + int(self.test_area)))
            pos = torch.rand(blocks * points, 3, generator=g, device=device) * 2 - 1
            x = torch.cat([torch.rand(blocks * points, 3, generator=g, device=device), (pos + 1) / 2], dim=1)
            edge_index, edge_attr = knn_graph(pos, blocks, k)
            point_y = torch.randint(0, self.NUM_CLASSES, (blocks * points,), generator=g, device=device)
            y = torch.mode(point_y.view(blocks, points), dim=1).values
            node_ptr = np.arange(blocks + 1, dtype=np.int64) * points
            edge_ptr = np.arange(blocks + 1, dtype=np.int64) * points * k
            store = GraphStore(x, edge_index, edge_attr, node_ptr, edge_ptr, y=y, pos=pos)
            store.point_y = point_y
            self._stores[split] = store
        return self._stores[split]

    def get_train_data(self) -> GraphStore:
        return self._make("train")

    def get_test_data(self) -> GraphStore:
        return self._make("test")
