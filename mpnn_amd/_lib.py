"""ctypes binding of libmpnn_amd.so (the C ABI declared in include/mpnn_amd.h).

The library is the product: there is no fallback.  If it is missing or an entry point
fails, the caller gets an exception carrying the library's own error text.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int32, c_int64, c_void_p

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

MPNN_F32 = 0
MPNN_BF16 = 1
MPNN_U8 = 2
ABI_VERSION = 17

LIB_PATH = os.environ.get("MPNN_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmpnn_amd.so")



class CollateDesc(ctypes.Structure):
    """struct mpnn_collate_desc of include/mpnn_amd.h (field for field)."""
    _fields_ = [
        ("slots", c_void_p), ("num_slots", c_int32), ("num_nodes", c_int64), ("num_edges", c_int64),
        ("x_all", c_void_p), ("x_row_bytes", c_int32), ("pos_all", c_void_p), ("pos_row_bytes", c_int32),
        ("y_all", c_void_p), ("y_row_bytes", c_int32),
        ("src_all", c_void_p), ("dst_all", c_void_p),
        ("ea_all", c_void_p), ("ea_dst_all", c_void_p), ("ea_src_all", c_void_p), ("ea_row_bytes", c_int32),
        ("rowptr_dst_all", c_void_p), ("col_dst_all", c_void_p), ("eid_dst_all", c_void_p),
        ("rowptr_src_all", c_void_p), ("col_src_all", c_void_p), ("eid_src_all", c_void_p), ("tpos_all", c_void_p),
        ("x", c_void_p), ("pos", c_void_p), ("y", c_void_p), ("batch", c_void_p), ("edge_index", c_void_p),
        ("ea", c_void_p), ("ea_dst", c_void_p), ("ea_src", c_void_p),
        ("rowptr_dst", c_void_p), ("col_dst", c_void_p), ("eid_dst", c_void_p), ("items_dst", c_void_p),
        ("rowptr_src", c_void_p), ("col_src", c_void_p), ("eid_src", c_void_p), ("items_src", c_void_p),
        ("tpos", c_void_p),
        ("node_capacity", c_int64), ("edge_capacity", c_int64), ("pool_rowptr", c_void_p),
    ]


class IslandLayerDesc(ctypes.Structure):
    """struct mpnn_island_layer of include/mpnn_amd.h (field for field)."""
    _fields_ = [
        ("rowptr", c_void_p), ("col", c_void_p), ("island_ptr", c_void_p),
        ("n_islands", c_int32), ("tail_units", c_int32), ("rows", c_int32), ("num_nodes", c_int64),
        ("x", c_void_p), ("ldx", c_int64),
        ("ea", c_void_p), ("w_nodes", c_void_p), ("w_edge", c_void_p), ("w_out", c_void_p),
        ("pos", c_void_p), ("w_pos", c_void_p),
        ("F", c_int32), ("H", c_int32), ("Fe", c_int32), ("dtype", c_int32),
        ("y", c_void_p), ("ldy", c_int64), ("pos_out", c_void_p),
        ("gy", c_void_p), ("ldgy", c_int64), ("gpos_out", c_void_p),
        ("gx", c_void_p), ("ldgx", c_int64), ("gpos", c_void_p),
        ("g_wn", c_void_p), ("g_we", c_void_p), ("g_wout", c_void_p), ("g_wpos", c_void_p),
        ("partial", c_void_p), ("gx_part", c_void_p),
    ]


# name -> (restype, argtypes); mirrors include/mpnn_amd.h one to one.
SIGNATURES = {
    "mpnn_islands_workspace_bytes": (c_int64, [c_int64]),
    "mpnn_islands_build": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mpnn_island_layer_waves": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mpnn_island_layer_partial_floats": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mpnn_island_layer_fwd": (c_int32, [ctypes.POINTER(IslandLayerDesc), c_void_p]),
    "mpnn_island_layer_bwd": (c_int32, [ctypes.POINTER(IslandLayerDesc), c_void_p]),
    "mpnn_region_layer_fits": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mpnn_region_layer_partial_floats": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mpnn_region_layer_fwd": (c_int32, [ctypes.POINTER(IslandLayerDesc), c_int32, c_void_p, c_void_p]),
    "mpnn_region_layer_bwd": (c_int32, [ctypes.POINTER(IslandLayerDesc), c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "mpnn_collate": (c_int32, [ctypes.POINTER(CollateDesc), c_void_p]),
    "mpnn_knn_workspace_floats": (c_int64, [c_int32, c_int32, c_int32]),
    "mpnn_knn_graph": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mpnn_masked_ce_scratch_floats": (c_int64, []),
    "mpnn_masked_ce": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64,
                                 c_void_p, c_void_p, c_void_p]),
    "mpnn_adam_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_float, ctypes.c_float, c_void_p]),
    "mpnn_abi_version": (c_int32, []),
    "mpnn_last_error": (c_char_p, []),
    "mpnn_csr_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "mpnn_csr_build": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_int64, c_void_p]),
    "mpnn_items_workspace_bytes": (c_int64, [c_int64]),
    "mpnn_items_build": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_void_p, c_int64, c_void_p]),
    "mpnn_csr_item_order_workspace_bytes": (c_int64, [c_int64]),
    "mpnn_csr_item_order": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mpnn_gather_rows": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "mpnn_aggregate_fwd": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_void_p,
                                     c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                     c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                     c_int64, c_int32, c_int32, c_void_p]),
    "mpnn_count_fixup": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                   c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "mpnn_aggregate_fwd_noalloc": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_void_p,
                                             c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                             c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                             c_int32, c_int32, c_void_p]),
    "mpnn_bwd_scratch_floats": (c_int64, [c_int32, c_int32]),
    "mpnn_mask_words": (c_int32, [c_int32, c_int32]),
    "mpnn_aggregate_bwd_src_masked": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                                c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                                c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "mpnn_aggregate_bwd_src": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                         c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_int32, c_int32, c_void_p]),
    "mpnn_egnn_fwd": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                c_void_p]),
    "mpnn_egnn_bwd_dst": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                    c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "mpnn_egnn_bwd_src": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                    c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "mpnn_gemm_nt_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                   c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                   c_void_p, c_int64, c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "mpnn_gemm_nt_scratch_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "mpnn_gemm_nt_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p,
                                    c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int64, c_void_p, c_int64,
                                    c_int64, c_int32, c_int32, c_void_p]),
    "mpnn_gemm_tn_scratch_floats": (c_int64, [c_int64, c_int32, c_int32]),
    "mpnn_gemm_tn_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
                                   c_void_p]),
    "mpnn_gemm_tn_bf16_scratch_floats": (c_int64, [c_int64, c_int32, c_int32]),
    "mpnn_gemm_tn_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
                                    c_void_p]),
    "mpnn_halo_pack": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p]),
    "mpnn_halo_unpack": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_int32,
                                   c_void_p]),
    "mpnn_segment_mean_fwd": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                        c_int32, c_int32, c_void_p]),
    "mpnn_segment_mean_bwd": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                        c_int32, c_int32, c_void_p]),
}


class MpnnError(RuntimeError):
    """An entry point of libmpnn_amd.so returned a non-zero status."""


_lib = None


def load():
    """Loads the shared library once; raises if it is absent or of another ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpnnError(
            f"{LIB_PATH} not found: build it with `make -C mpnn_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU or PyTorch fallback for the message-passing path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.mpnn_abi_version()
    if got != ABI_VERSION:
        raise MpnnError(f"libmpnn_amd.so has ABI version {got}, this package needs {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().mpnn_last_error()
        raise MpnnError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def ptr(t) -> int | None:
    """Device address of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_handle(device) -> int:
    """The current stream of ``device``, which must be the current device: the library launches on the device that
    is current in the calling thread (wrap calls on another GPU's tensors in ``torch.cuda.device(...)``)."""
    current = torch.cuda.current_device()
    index = device.index if device.index is not None else current
    if index != current:
        raise MpnnError(f"tensors live on cuda:{index} but cuda:{current} is the current device; "
                        "run the call under torch.cuda.device(tensor.device)")
    return _raw_stream(index)


def _raw_stream_slow(index: int) -> int:
    return torch.cuda.current_stream(index).cuda_stream


# the handle without the Stream object around it (a fifth of the cost; every library call asks for it)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", _raw_stream_slow)


def dtype_code(dtype) -> int:
    if dtype == torch.float32:
        return MPNN_F32
    if dtype == torch.bfloat16:
        return MPNN_BF16
    raise TypeError(f"mpnn_amd supports float32 and bfloat16 features, got {dtype}")


def require_device(t, name: str) -> None:
    if not t.is_cuda:
        raise MpnnError(
            f"{name} is on {t.device}: the message-passing engine runs on an AMD GPU only "
            "(there is no CPU path; the CPU oracle under oracle/ is test infrastructure).")
