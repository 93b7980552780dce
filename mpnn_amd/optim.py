"""Adam over one flat parameter buffer (a single library launch per step).

Same update rule and defaults as ``torch.optim.Adam(params, lr)`` the reference constructs
(experiments/flickr_exp.py:26, qm9_exp.py:33, modelnet_exp.py:49): betas (0.9, 0.999), eps 1e-8, no weight
decay, no amsgrad.  The parameters are re-pointed at slices of one float32 buffer; ``zero_grad()`` drops the
gradients (as torch's optimizers do by default), ``step()`` moves whatever autograd left in ``.grad`` into the
matching slices of one gradient buffer with a single multi-tensor copy and updates everything in one library call.
Nothing is read back to the host: the step counter lives on the device, which lets a whole training step be
captured into a HIP graph.
"""
from __future__ import annotations

import torch

from . import _lib


class FlatAdam:
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        for p in self.params:
            _lib.require_device(p, "parameter")
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("FlatAdam needs float32 parameters on one GPU")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self._flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.state = torch.zeros(3, dtype=torch.float32, device=dev)   # [steps taken, step size, sqrt(1 - beta2^t)]
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self._flat_grad[off:off + k].view(p.shape)
                off += k

    @property
    def flat_grad(self) -> torch.Tensor:
        """All gradients as one float32 vector (parameter order).  Reading it first gathers whatever backward left
        outside the buffer (zero_grad's default drops the views), so clipping or reducing ``opt.flat_grad`` between
        backward() and step() always sees the current gradients."""
        self.gather_stray_gradients()
        return self._flat_grad

    def zero_grad(self, set_to_none: bool = True) -> None:
        """Default: drops the gradients, so that backward hands its freshly written tensors over instead of adding
        them to zeros with one launch per parameter.  ``set_to_none=False`` keeps ``.grad`` as views of the flat
        buffer and clears it in one pass."""
        if set_to_none:
            for p in self.params:
                p.grad = None
            return
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self._flat_grad[off:off + k].view(p.shape)
            off += k
        self._flat_grad.zero_()

    def gather_stray_gradients(self) -> None:
        """Makes ``flat_grad`` hold every gradient and every ``.grad`` a view of it: gradients that live elsewhere
        (whatever backward produced after zero_grad) are copied into their slices by one multi-tensor copy; a
        parameter without a gradient gets zeros."""
        off = 0
        slots, strays = [], []
        for p in self.params:
            k = p.numel()
            slot = self._flat_grad[off:off + k].view(p.shape)
            if p.grad is None:
                slot.zero_()
                p.grad = slot
            elif p.grad.data_ptr() != self._flat_grad.data_ptr() + 4 * off:
                slots.append(slot)
                strays.append(p.grad.detach())
                p.grad = slot
            off += k
        if strays:
            torch._foreach_copy_(slots, strays)

    def step(self) -> None:
        self.gather_stray_gradients()
        lib = _lib.load()
        _lib.check(lib.mpnn_adam_step(_lib.ptr(self.flat), _lib.ptr(self._flat_grad), _lib.ptr(self.exp_avg),
                                      _lib.ptr(self.exp_avg_sq), self.flat.numel(), _lib.ptr(self.state), self.lr,
                                      self.betas[0], self.betas[1], self.eps, _lib.stream_handle(self.flat.device)),
                   "mpnn_adam_step")

    def state_snapshot(self):
        """Copies of everything a step changes (parameters included); see state_restore."""
        return [t.clone() for t in (self.flat, self.exp_avg, self.exp_avg_sq, self.state)]

    def state_restore(self, snapshot) -> None:
        """Puts a snapshot back in place (the tensors keep their addresses, so a recorded step stays valid)."""
        with torch.no_grad():
            for t, s0 in zip((self.flat, self.exp_avg, self.exp_avg_sq, self.state), snapshot):
                t.copy_(s0)

    @property
    def steps_taken(self) -> int:
        return int(self.state[0].item())
