"""MI355X-native message-passing engine behind the KutayTasci/MPNN model API."""
from .layers import EGNNLayer, MPNNLayerCat, MPNNLayerSum
from .mpnn import EGNN, MPNN, MPNN_QM9, MPNN_ModelNet
from .graph import MPGraph, graph_for
from .ops import aggregate, aggregate_fused, egnn_aggregate, global_mean_pool
from .data import Batch, GraphLoader, GraphStore

__all__ = ["EGNN", "EGNNLayer", "egnn_aggregate", "MPNNLayerCat", "MPNNLayerSum", "MPNN", "MPNN_QM9", "MPNN_ModelNet", "MPGraph", "graph_for",
           "aggregate", "aggregate_fused", "global_mean_pool", "Batch", "GraphLoader", "GraphStore"]
