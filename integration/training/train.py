"""Replacement for the reference's training/train.py: same trainer classes and epoch loops, HIP engine underneath."""
from mpnn_amd.train import (Trainer_QM9, Trainer_Flickr, Trainer_ModelNet, Train_QM9, Train_QM9_FB,  # noqa: F401
                            Train_Flickr, Train_ModelNet, Train_ModelNet_FB)
