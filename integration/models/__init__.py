from .mpnn import MPNN
from .layers import MPNNLayerCat, MPNNLayerSum
