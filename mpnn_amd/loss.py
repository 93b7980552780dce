"""Masked mean cross-entropy as one fused forward+gradient pass.

``masked_cross_entropy(out, y, mask)`` equals ``F.cross_entropy(out[mask], y[mask])``, the loss of the
reference's transductive trainer (training/train.py:47 with experiments/flickr_exp.py:31), without the
boolean-mask indexing - that indexing makes the host wait for the device to count the selected rows on
every step and cannot be captured into a HIP graph.
"""
from __future__ import annotations

import torch

from . import _lib


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, mask):
        lib = _lib.load()
        x = logits if logits.stride(-1) == 1 else logits.contiguous()
        n, c = x.shape
        dev = x.device
        grad = torch.empty((n, c), dtype=torch.float32, device=dev)
        result = torch.empty(2, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.mpnn_masked_ce_scratch_floats(), dtype=torch.float32, device=dev)
        m8 = mask.contiguous().view(torch.uint8) if mask is not None else None
        _lib.check(lib.mpnn_masked_ce(_lib.ptr(x), x.stride(0), _lib.ptr(labels), _lib.ptr(m8), n, c, _lib.ptr(grad),
                                      grad.stride(0), _lib.ptr(result), _lib.ptr(scratch), _lib.stream_handle(dev)),
                   "mpnn_masked_ce")
        ctx.save_for_backward(grad, result)
        return result[0]

    @staticmethod
    def backward(ctx, g):
        grad, result = ctx.saved_tensors
        return grad * (g * result[1]), None, None


@torch.compiler.disable
def masked_cross_entropy(logits: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor | None = None) -> torch.Tensor:
    _lib.require_device(logits, "logits")
    if logits.dim() != 2 or logits.dtype != torch.float32:
        raise TypeError("logits must be a float32 [N, C] matrix")
    if labels.dtype != torch.int64 or labels.shape != (logits.size(0),):
        raise TypeError("labels must be an int64 vector with one entry per row")
    if mask is not None and (mask.dtype != torch.bool or mask.shape != labels.shape):
        raise TypeError("mask must be a bool vector with one entry per row")
    return _MaskedCE.apply(logits, labels.contiguous(), mask)
