"""Replacement for the reference's models/mpnn.py: same class names, HIP engine underneath."""
from mpnn_amd.mpnn import MPNN, MPNN_QM9, MPNN_ModelNet  # noqa: F401
