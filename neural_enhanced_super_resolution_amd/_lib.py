"""ctypes binding of libnesr_hip.so (the C ABI declared in include/nesr_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, the caller gets
an exception (the reference turns exceptions from its ESRGAN backend into silent bicubic
results, nesr/nesr.py:815-843, so a quiet fallback here would be invisible).
"""
from __future__ import annotations

import ctypes
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnesr_hip.so")

DTYPE_F32, DTYPE_BF16, DTYPE_F32_WINOGRAD, DTYPE_F32_SPLIT = 0, 1, 2, 3
ROUND_TRUNC, ROUND_NEAREST = 0, 1

# name -> (restype, argtypes); must list every symbol include/nesr_hip.h declares
_c = ctypes
SIGNATURES = {
    "nesr_create": (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "nesr_load_weight": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_void_p, _c.POINTER(_c.c_int64), _c.c_int]),
    "nesr_finalize_weights": (_c.c_int, [_c.c_void_p]),
    "nesr_num_tensors": (_c.c_int, [_c.c_void_p]),
    "nesr_forward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "nesr_forward_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p]),
    "nesr_forward_ragged": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.c_void_p, _c.c_void_p]),
    "nesr_set_size_independent": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "nesr_workspace_bytes": (_c.c_size_t, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int]),
    "nesr_reserve": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int]),
    "nesr_preferred_batch": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int]),
    "nesr_forward_flops": (_c.c_double, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int]),
    "nesr_band_begin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "nesr_band_rdb": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p]),
    "nesr_band_tail": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "nesr_band_rdb_phase": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "nesr_band_pack_edges": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "nesr_band_unpack_aprons": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "nesr_band_row_bytes": (_c.c_size_t, [_c.c_void_p]),
    "nesr_band_rows": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "nesr_set_concurrent": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "nesr_set_fused": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "nesr_fused_state": (_c.c_int, [_c.c_void_p]),
    "nesr_debug_fault": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "nesr_set_kernel_timing": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "nesr_kernel_time_ms": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_double)]),
    "nesr_check_status": (_c.c_int, [_c.c_void_p]),
    "nesr_check_range": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "nesr_destroy": (None, [_c.c_void_p]),
    "nesr_cut_tiles_u8": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "nesr_paste_tiles_u8": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int64), _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int,
                                       _c.c_int, _c.c_void_p]),
    "nesr_comm_unique_id": (_c.c_int, [_c.c_void_p]),
    "nesr_comm_init": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p]),
    "nesr_comm_destroy": (_c.c_int, [_c.c_void_p]),
    "nesr_forward_sharded_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "nesr_shard_plan": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.c_int, _c.POINTER(_c.c_int),
                                   _c.POINTER(_c.c_int), _c.c_int, _c.POINTER(_c.c_int)]),
    "nesr_nl_means_u8": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "nesr_clahe_u8": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "nesr_conv3x3": (_c.c_int, [_c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "nesr_last_error": (_c.c_char_p, []),
    "nesr_version": (_c.c_char_p, []),
}

_lib = None
_lock = threading.Lock()


ERR_RANGE = -5


class NesrHipError(RuntimeError):
    """A libnesr_hip.so call returned a negative status."""


class NesrRangeError(NesrHipError, FloatingPointError):
    """NESR_ERR_RANGE: the f16-pair fp32 form met a weight, input or activation that is non-finite or beyond
    +-65504 (the reference would carry it in float32; here it is an error, never a saturated image)."""


def load():
    """Loads (once) and returns the ctypes handle; raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NesrHipError(
                f"{LIB_PATH} is missing: build it with `python -m neural_enhanced_super_resolution_amd.build` "
                "(hipcc, gfx950). There is no CPU or PyTorch fallback for this path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().nesr_last_error()
        cls = NesrRangeError if rc == ERR_RANGE else NesrHipError
        raise cls(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
