"""Replacement for the reference's data/dataset.py: same class and method names, synthetic data of the same shapes
served from GPU memory (the reference's classes download through torch_geometric, which needs a network)."""
from mpnn_amd.datasets import FlickrDataset, QM9Dataset, ModelNetDataset, S3DISDataset  # noqa: F401
