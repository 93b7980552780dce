"""Replacement for the reference's models/layers.py: same class names, HIP engine underneath."""
from mpnn_amd.layers import MPNNLayerCat, MPNNLayerSum  # noqa: F401
