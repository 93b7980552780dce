"""Dense node-level operators of the layer on the fp32 MFMA kernels of libmpnn_amd.so.

``project``  : AB = x @ W^T                                     (models/layers.py:104, and :56 after the column split)
``update``   : relu(x @ W_out[:, :F]^T + aggr @ W_out[:, F:]^T)  (models/layers.py:69,140, no cat)

Both carry their own backward (input gradients through the same NT kernel with a transposed weight, weight
gradients through the TN kernel).  float32, or bfloat16 with widths that are multiples of 8, on the GPU only:
anything else raises - there is no fallback to torch's library GEMMs.
"""
from __future__ import annotations

import torch

from . import _lib

_SCRATCH = {}


def _scratch(n: int, device) -> torch.Tensor:
    """Partial-tile scratch of the weight-gradient products, one buffer per (device, stream): calls on one stream
    reuse it in stream order, calls on different streams never share it."""
    key = (device, _lib.stream_handle(device))
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, dtype=torch.float32, device=device)
        _SCRATCH[key] = buf
    return buf


_WSPLIT = {}


def _wsplit(nbytes: int, device) -> torch.Tensor:
    """Room for the split weight images of one NT product.  One buffer per device and stream-ordered reuse: every
    call rewrites it before its own GEMM reads it, on the stream both run on."""
    key = (device, _lib.stream_handle(device))
    buf = _WSPLIT.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WSPLIT[key] = buf
    return buf


def _rows(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 and t.dim() == 2 else t.contiguous()


def _ld(t) -> int:
    """Leading dimension of a matrix with unit column stride, 0 for None.  A single-row matrix may carry any
    stride(0) (torch keeps e.g. 1 after ``.t().contiguous()`` of a one-column matrix): its row length serves."""
    if t is None:
        return 0
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


def _emul_kind(emul) -> int:
    """Kind of an epilogue multiplier: float32, or one byte per entry (the counts of the forward edge pass)."""
    if emul is None or emul.dtype == torch.float32:
        return _lib.MPNN_F32
    if emul.dtype == torch.uint8:
        return _lib.MPNN_U8
    raise TypeError(f"emul must be float32 or uint8, got {emul.dtype}")


def gemm_nt(x1, w1, x2=None, w2=None, mask1=None, mask2=None, relu=False, emul=None, out2=None, add=None,
            out=None) -> torch.Tensor:
    """act(x1 @ w1^T [+ x2 @ w2^T] [+ add]); maskI > 0 selects the entries of xI that are used.

    With ``emul`` [N, M] (float32 or uint8) and ``out2`` (a [N, M] view with unit column stride) the epilogue also
    writes out2 = result * emul.  ``add`` is a [N, M] matrix (unit column stride) added before the activation;
    ``out`` a [N, M] view to write into instead of a new matrix."""
    lib = _lib.load()
    x1, w1 = _rows(x1), _rows(w1)
    n, k1 = x1.shape
    m = w1.size(0)
    y = torch.empty((n, m), dtype=torch.float32, device=x1.device) if out is None else out
    if x2 is not None:
        x2, w2 = _rows(x2), _rows(w2)
    mask1 = _rows(mask1) if mask1 is not None else None
    mask2 = _rows(mask2) if mask2 is not None else None
    k2 = x2.size(1) if x2 is not None else 0
    wimg = _wsplit(lib.mpnn_gemm_nt_scratch_bytes(m, k1, k2), x1.device)
    _lib.check(lib.mpnn_gemm_nt_f32(
        _lib.ptr(x1), _ld(x1), _lib.ptr(mask1), _ld(mask1), _lib.ptr(w1),
        _ld(w1), k1,
        _lib.ptr(x2), _ld(x2), _lib.ptr(mask2),
        _ld(mask2), _lib.ptr(w2), _ld(w2),
        x2.size(1) if x2 is not None else 0,
        _lib.ptr(y), _ld(y), _lib.ptr(emul), _emul_kind(emul), _ld(emul), _lib.ptr(out2),
        _ld(out2), _lib.ptr(add), _ld(add), n, m, 1 if relu else 0, _lib.ptr(wimg),
        _lib.stream_handle(x1.device)),
        "mpnn_gemm_nt_f32")
    return y


def bf16_ok(*ts) -> bool:
    """True if every matrix can go through mpnn_gemm_nt_bf16: bfloat16, unit column stride, a width that is a
    multiple of 8 and 16-byte aligned rows."""
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
            return False
        if t.size(1) % 8 or t.stride(0) % 8 or t.data_ptr() % 16:
            return False
    return True


def gemm_nt_bf16(x1, w1, x2=None, w2=None, relu=False, emul=None, out2=None) -> torch.Tensor:
    """bfloat16 act(x1 @ w1^T [+ x2 @ w2^T]) with float32 accumulation; ``emul`` (float32 [N, M]) and ``out2``
    (bfloat16 [N, M] view) give the second output out2 = result * emul."""
    lib = _lib.load()
    n, k1 = x1.shape
    m = w1.size(0)
    y = torch.empty((n, m), dtype=torch.bfloat16, device=x1.device)
    _lib.check(lib.mpnn_gemm_nt_bf16(
        _lib.ptr(x1), _ld(x1), _lib.ptr(w1), _ld(w1), k1,
        _lib.ptr(x2), _ld(x2), _lib.ptr(w2), _ld(w2),
        x2.size(1) if x2 is not None else 0,
        _lib.ptr(y), _ld(y), _lib.ptr(emul), _emul_kind(emul), _ld(emul), _lib.ptr(out2),
        _ld(out2), n, m, 1 if relu else 0, _lib.stream_handle(x1.device)),
        "mpnn_gemm_nt_bf16")
    return y


def gemm_tn_bf16(g, x, mask=None, out=None, g_out=None, x2=None) -> torch.Tensor:
    """bfloat16 (g * (mask > 0))^T @ [x | x2] -> [M, K (+ K2)] with float32 accumulation; ``out`` may be a column
    slice of a wider weight gradient, ``g_out`` ([N, M]) receives g * (mask > 0)."""
    lib = _lib.load()
    n, m = g.shape
    k = x.size(1)
    k2 = x2.size(1) if x2 is not None else 0
    if out is None:
        out = torch.empty((m, k + k2), dtype=torch.bfloat16, device=g.device)
    scratch = _scratch(lib.mpnn_gemm_tn_bf16_scratch_floats(n, m, k + k2), g.device)
    _lib.check(lib.mpnn_gemm_tn_bf16(
        _lib.ptr(g), _ld(g), _lib.ptr(mask), _ld(mask), _lib.ptr(x),
        _ld(x), _lib.ptr(x2), _ld(x2), _lib.ptr(out), _ld(out),
        _lib.ptr(scratch), _lib.ptr(g_out), _ld(g_out), n, m, k, k2,
        _lib.stream_handle(g.device)), "mpnn_gemm_tn_bf16")
    return out


def gemm_tn(g, x, mask=None, out=None, g_out=None, x2=None) -> torch.Tensor:
    """(g * (mask > 0))^T @ [x | x2]  ->  [M, K (+ K2)]; ``out`` may be a column slice of a wider weight gradient.
    ``g_out`` ([N, M]) receives g * (mask > 0)."""
    lib = _lib.load()
    g, x = _rows(g), _rows(x)
    x2 = _rows(x2) if x2 is not None else None
    n, m = g.shape
    k = x.size(1)
    k2 = x2.size(1) if x2 is not None else 0
    if out is None:
        out = torch.empty((m, k + k2), dtype=torch.float32, device=g.device)
    mask = _rows(mask) if mask is not None else None
    scratch = _scratch(lib.mpnn_gemm_tn_scratch_floats(n, m, k + k2), g.device)
    _lib.check(lib.mpnn_gemm_tn_f32(
        _lib.ptr(g), _ld(g), _lib.ptr(mask), _ld(mask), _lib.ptr(x),
        _ld(x), _lib.ptr(x2), _ld(x2), _lib.ptr(out), _ld(out),
        _lib.ptr(scratch), _lib.ptr(g_out), _ld(g_out), n, m, k, k2,
        _lib.stream_handle(g.device)), "mpnn_gemm_tn_f32")
    return out


def _kit(dtype):
    """(NT product, TN product) of the library for a storage dtype."""
    return (gemm_nt_bf16, gemm_tn_bf16) if dtype == torch.bfloat16 else (gemm_nt, gemm_tn)


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _kit(x.dtype)[0](x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        nt, tn = _kit(x.dtype)
        g = _rows(g)
        gx = nt(g, w.t().contiguous()) if ctx.needs_input_grad[0] else None
        gw = tn(g, x) if ctx.needs_input_grad[1] else None
        return gx, gw


class _Update(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, aggr, w_out):
        f = x.size(1)
        wx, wa = w_out[:, :f], w_out[:, f:]
        if x.dtype == torch.bfloat16:
            wx, wa = wx.contiguous(), wa.contiguous()
        y = _kit(x.dtype)[0](x, wx, aggr, wa, relu=True)
        ctx.save_for_backward(x, aggr, w_out, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, aggr, w_out, y = ctx.saved_tensors
        nt, tn = _kit(x.dtype)
        f = x.size(1)
        gz = torch.ops.aten.threshold_backward(_rows(gy), y, 0)    # gy * [y > 0], elementwise
        gx = ga = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wt = w_out.t().contiguous()          # [F+H, H]: rows of W^T are the outputs of the input-gradient GEMMs
            if ctx.needs_input_grad[0]:
                gx = nt(gz, wt[:f])
            if ctx.needs_input_grad[1]:
                ga = nt(gz, wt[f:])
        if ctx.needs_input_grad[2]:
            gw = torch.empty_like(w_out)
            tn(gz, x, out=gw, x2=aggr)
        return gx, ga, gw


def _require(name, *ts) -> None:
    """The dense operators exist on the GPU only: float32, or bfloat16 with 16-byte rows.  Nothing falls back to
    torch's library GEMMs."""
    for t in ts:
        if not t.is_cuda:
            raise _lib.MpnnError(f"dense.{name}: operand on {t.device}; the MFMA kernels run on an AMD GPU only")
    if all(t.dtype == torch.float32 for t in ts):
        return
    if all(t.dtype == torch.bfloat16 for t in ts):
        x = ts[0]
        if x.size(1) % 8 or any(t.size(0) % 8 for t in ts[-1:]):
            raise ValueError(f"dense.{name}: bfloat16 operands need widths that are multiples of 8 "
                             f"(got {[tuple(t.shape) for t in ts]}); pad the feature columns with zeros")
        return
    raise TypeError(f"dense.{name}: operands must share float32 or bfloat16, got {[t.dtype for t in ts]}")


@torch.compiler.disable
def project(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    _require("project", x, w)
    if x.dtype == torch.bfloat16:
        x, w = x.contiguous(), w.contiguous()
    return _Project.apply(x, w)


@torch.compiler.disable
def update(x: torch.Tensor, aggr: torch.Tensor, w_out: torch.Tensor) -> torch.Tensor:
    _require("update", x, aggr, w_out)
    if x.dtype == torch.bfloat16:
        x, aggr = x.contiguous(), aggr.contiguous()
    return _Update.apply(x, aggr, w_out)


_ONES = {}


def _ones(n: int, device) -> torch.Tensor:
    """[n, 1] ones (the right-hand side that turns a weight-gradient product into column sums), kept per device."""
    t = _ONES.get(device)
    if t is None or t.size(0) < n:
        t = _ONES[device] = torch.ones((max(n, 1024), 1), dtype=torch.float32, device=device)
    return t[:n]


class _Head(torch.autograd.Function):
    """y = x W^T + b on the library's GEMMs.  The backward of a node-level head (models/mpnn.py:99-101 on 89,250 rows)
    is two reductions over the rows - the weight gradient g^T x and the bias gradient, the column sums of g; both
    come out of one weight-gradient product over [x | 1]."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        y = gemm_nt(x, w)
        return y.add_(b) if b is not None else y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _rows(g)
        k = x.size(1)
        gx = gemm_nt(g, w.t().contiguous()) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[2]:
            both = gemm_tn(g, x, x2=_ones(x.size(0), x.device))
            gw, gb = both[:, :k].contiguous(), both[:, k].contiguous()
        elif ctx.needs_input_grad[1]:
            gw = gemm_tn(g, x)
        return gx, gw, gb


HEAD_MIN_ROWS = 4096


@torch.compiler.disable
def _head(x, weight, bias):
    return _Head.apply(x if x.stride(-1) == 1 else x.contiguous(), weight, bias)


def head(x: torch.Tensor, linear: torch.nn.Linear) -> torch.Tensor:
    """``linear(x)`` for the output layer of a model.  Tall float32 inputs (a node-level head: one row per node) go
    through the library's GEMMs, whose weight-gradient product reduces over the rows in one pass; the few pooled rows
    of a graph-level head, bfloat16 models and traced (torch.compile) calls stay with the module itself."""
    if (x.is_cuda and x.dtype == torch.float32 and linear.weight.dtype == torch.float32 and x.dim() == 2
            and x.size(0) >= HEAD_MIN_ROWS and not torch.compiler.is_compiling()):
        return _head(x, linear.weight, linear.bias)
    return linear(x)
