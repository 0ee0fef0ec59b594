"""ctypes loader for crypto_amd/lib/libqfgpu.so (the C ABI declared in include/qfgpu.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the
product path raises -- a GPU box must never silently run something else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QF_LIB_PATH") or os.path.join(_HERE, "lib", "libqfgpu.so")  # override: diagnostic builds only

QF_OK, QF_ERR_ARG, QF_ERR_DIM, QF_ERR_MODULUS, QF_ERR_DEVICE, QF_ERR_RANGE, QF_ERR_NOTSUP = range(7)
X_BITS, X_U32, X_I32, X_BITS_KT, X_PLANES = 0, 1, 2, 3, 4

_ERR_NAMES = {1: "QF_ERR_ARG", 2: "QF_ERR_DIM", 3: "QF_ERR_MODULUS", 4: "QF_ERR_DEVICE",
              5: "QF_ERR_RANGE", 6: "QF_ERR_NOTSUP"}


class QfError(RuntimeError):
    """Nonzero return code of the C ABI (the reference panics in the same situations)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


# every symbol include/qfgpu.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _i, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t
_pp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "qf_version": (C.c_char_p, []),
    "qf_device_count": (_i, [C.POINTER(_i)]),
    "qf_ctx_create": (_i, [_i, _vp, _pp]),
    "qf_ctx_set_stream": (_i, [_vp, _vp]),
    "qf_ctx_destroy": (None, [_vp]),
    "qf_sync": (_i, [_vp]),
    "qf_last_error": (C.c_char_p, [_vp]),
    "qf_malloc": (_i, [_vp, _sz, _pp]),
    "qf_free": (_i, [_vp, _vp]),
    "qf_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "qf_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "qf_key_upload_host": (_i, [_vp, _vp, _u32, _u32, _u32, _i, _pp]),
    "qf_key_upload_dev": (_i, [_vp, _vp, _u32, _u32, _u32, _i, _pp]),
    "qf_key_destroy": (None, [_vp]),
    "qf_key_dims": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "qf_key_u32_dev": (_i, [_vp, _pp, C.POINTER(_sz)]),
    "qf_key_replicate": (_i, [_pp, _i, _vp, _pp]),
    "qf_matmul_modq_dev": (_i, [_vp, _vp, _vp, _i, _sz, _i, _u64, _vp, _vp, _vp]),
    "qf_bits_to_kt_dev": (_i, [_vp, _vp, _sz, _u64, _u32, _vp, _sz]),
    "qf_operand_planes_bytes": (_sz, [_vp, _u64, _i]),
    "qf_recode_planes_dev": (_i, [_vp, _vp, _vp, _i, _sz, _i, _u64, _vp]),
    "qf_matvec_modq_host": (_i, [_vp, _vp, _vp, _vp]),
    "qf_vecmat_modq_host": (_i, [_vp, _vp, _vp, _vp]),
    "qf_matvec_modq_dev": (_i, [_vp, _vp, _vp, _vp]),
    "qf_vecmat_modq_dev": (_i, [_vp, _vp, _vp, _vp]),
    "qf_vecmat_u32_dev": (_i, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "qf_matvec_modq_static_dev": (_i, [_vp, _vp, _vp, _vp]),
    "qf_dot_decode_dev": (_i, [_vp, _vp, _u32, _u32, _vp, _u64, _vp]),
    "qf_hash_to_mat_zq_sha256_dev": (_i, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "qf_ring_create": (_i, [_vp, _u32, _u32, _pp]),
    "qf_ring_destroy": (None, [_vp]),
    "qf_ring_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_i)]),
    "qf_rkey_upload_host": (_i, [_vp, _vp, _vp, _u32, _pp]),
    "qf_rkey_upload_dev": (_i, [_vp, _vp, _vp, _u32, _pp]),
    "qf_rkey_destroy": (None, [_vp]),
    "qf_polymul_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "qf_ringlpr_enc_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp]),
    "qf_ringlpr_dec_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _u64, _vp]),
    "qf_fdh_ring_verify_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _vp]),
    "qf_sample_uniform_dev": (_i, [_vp, _u64, _u32, _u64, _u64, _u32, _vp]),
    "qf_sample_bits_dev": (_i, [_vp, _u64, _u32, _u64, _u64, _u32, _u32, _vp]),
    "qf_sample_bits_kt_dev": (_i, [_vp, _u64, _u32, _u64, _u64, _u32, _sz, _vp]),
    "qf_sample_noise_dev": (_i, [_vp, _u64, _u32, _u64, _u64, _u32, _u32, _vp]),
    "qf_sample_discrete_gauss_dev": (_i, [_vp, _u64, _u32, _u64, _u64, _u32, _u32, C.c_double, _vp]),
    "qf_sample_uniform_keyed_dev": (_i, [_vp, C.c_char_p, _u64, _u32, _u64, _u64, _u32, _vp]),
    "qf_sample_bits_keyed_dev": (_i, [_vp, C.c_char_p, _u64, _u32, _u64, _u64, _u32, _u32, _vp]),
    "qf_sample_bits_kt_keyed_dev": (_i, [_vp, C.c_char_p, _u64, _u32, _u64, _u64, _u32, _sz, _vp]),
    "qf_sample_noise_keyed_dev": (_i, [_vp, C.c_char_p, _u64, _u32, _u64, _u64, _u32, _u32, _vp]),
    "qf_sample_discrete_gauss_keyed_dev": (_i, [_vp, C.c_char_p, _u64, _u32, _u64, _u64, _u32, _u32, C.c_double, _vp]),
    "qf_set_timing": (_i, [_vp, _i]),
    "qf_last_kernel_ms": (_i, [_vp, C.c_char_p, C.POINTER(C.c_float)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libqfgpu.so and type every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the product path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
