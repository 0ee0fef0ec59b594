"""ctypes binding of oracle/_build/libqf_oracle.so (the C restatement, qf_oracle.c).

TEST INFRASTRUCTURE -- see qf_oracle.c.  numpy in / numpy out.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqf_oracle.so")

X_BITS, X_U32, X_I32 = 0, 1, 2


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("qf_oracle.c", "qf_oracle_mt.c", "Makefile")]
    if force or not os.path.exists(_SO) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.qfo_rand64.restype = C.c_uint64
        _lib.qfo_rand64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        _lib.qfo_rand64_keyed.restype = C.c_uint64
        _lib.qfo_rand64_keyed.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64]
        _lib.qfo_find_psi.restype = C.c_uint32
        _lib.qfo_hash_to_zq.restype = C.c_uint32
        _lib.qfo_decide_bit.restype = C.c_uint32
        _lib.qfo_regev_cycle.restype = C.c_uint32
        _lib.qfo_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def sample_uniform(seed, stream, first, count, q):
    if isinstance(seed, tuple):          # (key32, nonce): the keyed generator
        return sample_uniform_keyed(seed[0], seed[1], stream, first, count, q)
    out = np.empty(count, np.uint32)
    lib().qfo_sample_uniform(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(first),
                             C.c_uint64(count), C.c_uint32(q), _p(out))
    return out


def sample_noise(seed, stream, first, count, q, W):
    if isinstance(seed, tuple):          # (key32, nonce): the keyed generator
        return sample_noise_keyed(seed[0], seed[1], stream, first, count, q, W)
    out = np.empty(count, np.uint32)
    lib().qfo_sample_noise(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(first),
                           C.c_uint64(count), C.c_uint32(q), C.c_uint32(W), _p(out))
    return out


def sample_discrete_gauss(seed, stream, first, count, q, n, s):
    if isinstance(seed, tuple):          # (key32, nonce): the keyed generator
        return sample_discrete_gauss_keyed(seed[0], seed[1], stream, first, count, q, n, s)
    out = np.empty(count, np.uint32)
    rc = lib().qfo_sample_discrete_gauss(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(first),
                                         C.c_uint64(count), C.c_uint32(q), C.c_uint32(n), C.c_double(s), _p(out))
    if rc:
        raise ValueError("discrete gauss: support s*log2(n) outside [1, 2^21]")
    return out


def sample_bits(seed, stream, first_row, rows, cols, stride_bytes):
    if isinstance(seed, tuple):          # (key32, nonce): the keyed generator
        return sample_bits_keyed(seed[0], seed[1], stream, first_row, rows, cols, stride_bytes)
    out = np.empty((rows, stride_bytes), np.uint8)
    lib().qfo_sample_bits(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(first_row),
                          C.c_uint64(rows), C.c_uint32(cols), C.c_uint32(stride_bytes), _p(out))
    return out


# ---- the same samplers under a 256-bit key (ChaCha20 value function; include/qfgpu.h "*_keyed_dev")
def chacha20_block(key32: bytes, counter: int, nonce: int):
    out = np.empty(16, np.uint32)
    lib().qfo_chacha20_block(C.c_char_p(bytes(key32)), C.c_uint64(counter), C.c_uint64(nonce), _p(out))
    return out


def rand64_keyed(key32: bytes, nonce, stream, idx):
    return int(lib().qfo_rand64_keyed(bytes(key32), nonce, stream, idx))


def sample_uniform_keyed(key32, nonce, stream, first, count, q):
    out = np.empty(count, np.uint32)
    lib().qfo_sample_uniform_keyed(C.c_char_p(bytes(key32)), C.c_uint64(nonce), C.c_uint32(stream), C.c_uint64(first),
                                   C.c_uint64(count), C.c_uint32(q), _p(out))
    return out


def sample_noise_keyed(key32, nonce, stream, first, count, q, W):
    out = np.empty(count, np.uint32)
    lib().qfo_sample_noise_keyed(C.c_char_p(bytes(key32)), C.c_uint64(nonce), C.c_uint32(stream), C.c_uint64(first),
                                 C.c_uint64(count), C.c_uint32(q), C.c_uint32(W), _p(out))
    return out


def sample_discrete_gauss_keyed(key32, nonce, stream, first, count, q, n, s):
    out = np.empty(count, np.uint32)
    rc = lib().qfo_sample_discrete_gauss_keyed(C.c_char_p(bytes(key32)), C.c_uint64(nonce), C.c_uint32(stream),
                                               C.c_uint64(first), C.c_uint64(count), C.c_uint32(q), C.c_uint32(n),
                                               C.c_double(s), _p(out))
    if rc:
        raise ValueError("discrete gauss: support s*log2(n) outside [1, 2^21]")
    return out


def sample_bits_keyed(key32, nonce, stream, first_row, rows, cols, stride_bytes):
    out = np.empty((rows, stride_bytes), np.uint8)
    lib().qfo_sample_bits_keyed(C.c_char_p(bytes(key32)), C.c_uint64(nonce), C.c_uint32(stream), C.c_uint64(first_row),
                                C.c_uint64(rows), C.c_uint32(cols), C.c_uint32(stride_bytes), _p(out))
    return out


def matvec(A, x, q):
    A = _u32(A); x = _u32(x)
    y = np.empty(A.shape[0], np.uint32)
    lib().qfo_matvec(_p(A), C.c_uint32(A.shape[0]), C.c_uint32(A.shape[1]), C.c_uint32(q), _p(x), _p(y))
    return y


def vecmat(x, A, q):
    A = _u32(A); x = _u32(x)
    y = np.empty(A.shape[1], np.uint32)
    lib().qfo_vecmat(_p(A), C.c_uint32(A.shape[0]), C.c_uint32(A.shape[1]), C.c_uint32(q), _p(x), _p(y))
    return y


def matmul_batch(A, q, X, xkind, B, addend=None, msg=None, mt=False):
    A = _u32(A)
    X = np.ascontiguousarray(X)
    stride = X.strides[0]
    out = np.empty((B, A.shape[0]), np.uint32)
    addend = _u32(addend) if addend is not None else None
    msg = np.ascontiguousarray(msg, np.uint8) if msg is not None else None
    fn = lib().qfo_matmul_batch_mt if mt else lib().qfo_matmul_batch
    fn(_p(A), C.c_uint32(A.shape[0]), C.c_uint32(A.shape[1]), C.c_uint32(q), _p(X), C.c_int(xkind),
       C.c_size_t(stride), C.c_uint64(B), _p(addend), _p(msg), _p(out))
    return out


def dot_decode_batch(s, q, Cm, mt=False):
    s = _u32(s); Cm = _u32(Cm)
    B = Cm.shape[0]
    bits = np.empty(B, np.uint8)
    fn = lib().qfo_dot_decode_batch_mt if mt else lib().qfo_dot_decode_batch
    fn(_p(s), C.c_uint32(s.size), C.c_uint32(q), _p(Cm), C.c_uint64(B), _p(bits))
    return bits


def dualregev_decode_batch(x, q, Cm):
    x = np.ascontiguousarray(x, np.int32); Cm = _u32(Cm)
    B = Cm.shape[0]
    bits = np.empty(B, np.uint8)
    lib().qfo_dualregev_decode_batch(_p(x), C.c_uint32(x.size), C.c_uint32(q), _p(Cm), C.c_uint64(B), _p(bits))
    return bits


def rot_minus_i64(a):
    a = np.ascontiguousarray(a, np.int64)
    out = np.empty((a.size, a.size), np.int64)
    lib().qfo_rot_minus_i64(_p(a), C.c_uint32(a.size), _p(out))
    return out


def rot_minus_modq(a, q):
    a = _u32(a)
    out = np.empty((a.size, a.size), np.uint32)
    lib().qfo_rot_minus_modq(_p(a), C.c_uint32(a.size), C.c_uint32(q), _p(out))
    return out


def polymul_batch(a, b, q, use_ntt=False, mt=False):
    a = _u32(a); b = _u32(b)
    a2 = a.reshape(-1, a.shape[-1]); b2 = b.reshape(-1, b.shape[-1])
    B, n = a2.shape
    c = np.empty((B, n), np.uint32)
    fn = lib().qfo_polymul_batch_mt if mt else lib().qfo_polymul_batch
    rc = fn(_p(a2), _p(b2), C.c_uint32(n), C.c_uint32(q), C.c_uint64(B), C.c_int(int(use_ntt)), _p(c))
    if rc:
        raise ValueError("q is not NTT-friendly for this n")
    return c.reshape(a.shape)


def find_psi(n, q):
    return int(lib().qfo_find_psi(C.c_uint32(n), C.c_uint32(q)))


def ringlpr_enc_batch(a, b, q, r, e1, e2, mu_bits):
    a = _u32(a); b = _u32(b); r = _u32(r); e1 = _u32(e1); e2 = _u32(e2)
    mu_bits = np.ascontiguousarray(mu_bits, np.uint8)
    B, n = r.shape
    u = np.empty((B, n), np.uint32); v = np.empty((B, n), np.uint32)
    lib().qfo_ringlpr_enc_batch(_p(a), _p(b), C.c_uint32(n), C.c_uint32(q), _p(r), _p(e1), _p(e2),
                                _p(mu_bits), C.c_uint64(B), _p(u), _p(v))
    return u, v


def ringlpr_dec_batch(s, q, u, v):
    s = _u32(s); u = _u32(u); v = _u32(v)
    B, n = u.shape
    mu = np.empty((B, (n + 7) // 8), np.uint8)
    lib().qfo_ringlpr_dec_batch(_p(s), C.c_uint32(n), C.c_uint32(q), _p(u), _p(v), C.c_uint64(B), _p(mu))
    return mu


def fdh_ring_verify_batch(A, q, sigma, u, norm_bound):
    A = _u32(A); sigma = np.ascontiguousarray(sigma, np.int32); u = _u32(u)
    mbar, n = A.shape
    B = sigma.shape[0]
    t = np.empty((B, n), np.uint32); ok = np.empty(B, np.uint8)
    lib().qfo_fdh_ring_verify_batch(_p(A), C.c_uint32(mbar), C.c_uint32(n), C.c_uint32(q), _p(sigma),
                                    _p(u), C.c_uint64(norm_bound), C.c_uint64(B), _p(t), _p(ok))
    return t, ok


def sha256(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().qfo_sha256(data, C.c_size_t(len(data)), out)
    return bytes(out)


def hash_to_zq(s: str, q: int) -> int:
    b = s.encode()
    return int(lib().qfo_hash_to_zq(b, C.c_size_t(len(b)), C.c_uint32(q)))


def hash_to_mat_zq(s: str, rows: int, cols: int, q: int):
    b = s.encode()
    out = np.empty((rows, cols), np.uint32)
    lib().qfo_hash_to_mat_zq(b, C.c_size_t(len(b)), C.c_uint32(rows), C.c_uint32(cols), C.c_uint32(q), _p(out))
    return out


def regev_cycle(n, m, q, W, seed, msg):
    return int(lib().qfo_regev_cycle(C.c_uint32(n), C.c_uint32(m), C.c_uint32(q), C.c_uint32(W),
                                     C.c_uint64(seed), C.c_uint32(msg)))


def num_threads():
    return int(lib().qfo_num_threads())
