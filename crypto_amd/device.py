"""Thin object layer over the C ABI: contexts, keys, rings.

PyTorch is used only as plumbing here -- device allocations (`torch.empty(..., device=cuda)`),
the current HIP stream, and (in `crypto_amd.sharding`) `torch.distributed`.  All arithmetic
happens inside libqfgpu.so.  Z_q buffers are carried as `torch.int32` tensors holding the bit
pattern of the `uint32_t` residues (torch has no full uint32 support); use `to_host_u32`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import QfError, X_BITS, X_BITS_KT, X_I32, X_PLANES, X_U32  # noqa: F401


def _check(ctx_handle, rc: int):
    if rc != 0:
        msg = _lib.load().qf_last_error(ctx_handle)
        raise QfError(rc, msg.decode() if msg else "")


def bit_stride(cols: int) -> int:
    """Row stride in bytes of a bit-packed batch operand (multiple of 16 bytes)."""
    return ((cols + 127) // 128) * 16


def pack_bits(bits: np.ndarray, stride: Optional[int] = None) -> np.ndarray:
    """(B, cols) array of 0/1 -> (B, stride) uint8, LSB-first, zero padded."""
    bits = np.asarray(bits, dtype=np.uint8)
    B, cols = bits.shape
    stride = stride or bit_stride(cols)
    packed = np.packbits(bits, axis=1, bitorder="little")
    out = np.zeros((B, stride), np.uint8)
    out[:, :packed.shape[1]] = packed
    return out


def unpack_bits(packed: np.ndarray, cols: int) -> np.ndarray:
    return np.unpackbits(np.asarray(packed, np.uint8), axis=1, bitorder="little")[:, :cols]


class Context:
    """One device, one stream (include/qfgpu.h: qf_ctx)."""

    def __init__(self, device: int = 0, stream: Optional[torch.cuda.Stream] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("crypto_amd needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._stream = stream or torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        rc = self.lib.qf_ctx_create(device, C.c_void_p(self._stream.cuda_stream), C.byref(h))
        if rc != 0:
            raise QfError(rc, "qf_ctx_create failed")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.qf_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffers -------------------------------------------------------------------------
    def empty_u32(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.int32, device=self.device)

    def empty_u8(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.uint8, device=self.device)

    def to_dev_u32(self, a) -> torch.Tensor:
        a = np.ascontiguousarray(np.asarray(a, dtype=np.uint32))
        return torch.from_numpy(a.view(np.int32)).to(self.device)

    def to_dev_i32(self, a) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int32))).to(self.device)

    def to_dev_u8(self, a) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.uint8))).to(self.device)

    # Results of 4 KiB .. 1 MiB come back through one pinned staging buffer per context: a pageable
    # `.cpu()` of a 200 KB key image costs ~45 us, the pinned copy ~15 us (below 4 KiB the two are
    # equal and `.cpu()` has less Python around it).
    _PIN_BYTES = 1 << 20

    def _to_host(self, t: torch.Tensor) -> np.ndarray:
        t = t.detach()
        nbytes = t.numel() * t.element_size()
        if t.is_cuda and self._stream != torch.cuda.current_stream(self.device):
            self._stream.synchronize()                        # `.cpu()` below only orders behind torch's current stream
        if nbytes < 4096 or nbytes > self._PIN_BYTES or not t.is_contiguous() or not t.is_cuda:
            return t.cpu().numpy()
        pin = getattr(self, "_pin", None)
        if pin is None:
            pin = self._pin = torch.empty(self._PIN_BYTES, dtype=torch.uint8).pin_memory()
        stage = pin[:nbytes].view(t.dtype).view(t.shape)      # one staging buffer: copy, wait, copy out
        with torch.cuda.stream(self._stream):                 # behind the library's kernels: the CONTEXT's stream,
            stage.copy_(t, non_blocking=True)                 # which need not be torch's current one
        self._stream.synchronize()
        return stage.numpy().copy()

    def to_host_u32(self, t: torch.Tensor) -> np.ndarray:
        return self._to_host(t).view(np.uint32)

    def to_host_u8(self, t: torch.Tensor) -> np.ndarray:
        return self._to_host(t)

    def sync(self):
        _check(self.h, self.lib.qf_sync(self.h))

    def set_timing(self, on: bool):
        _check(self.h, self.lib.qf_set_timing(self.h, int(on)))

    def last_kernel_ms(self, which: str) -> float:
        ms = C.c_float()
        _check(self.h, self.lib.qf_last_kernel_ms(self.h, which.encode(), C.byref(ms)))
        return ms.value

    # ---- Z_q matrices --------------------------------------------------------------------
    def key_upload(self, A, q: int, transpose: bool = False) -> "Key":
        """A: (rows, cols) numpy array or device int32 tensor of residues."""
        h = C.c_void_p()
        if isinstance(A, torch.Tensor):
            assert A.is_cuda and A.dtype == torch.int32 and A.is_contiguous() and A.dim() == 2
            rc = self.lib.qf_key_upload_dev(self.h, A.data_ptr(), A.shape[0], A.shape[1], q, int(transpose), C.byref(h))
        else:
            A = np.ascontiguousarray(np.asarray(A, dtype=np.uint32))
            if A.ndim != 2:
                raise QfError(_lib.QF_ERR_DIM, "key must be a matrix")
            rc = self.lib.qf_key_upload_host(self.h, A.ctypes.data, A.shape[0], A.shape[1], q, int(transpose), C.byref(h))
        _check(self.h, rc)
        return Key(self, h)

    def matmul(self, key: "Key", X: torch.Tensor, kind: int, B: int, addend: Optional[torch.Tensor] = None,
               msg: Optional[torch.Tensor] = None, x_digits: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = self.empty_u32(B, key.rows)
        if kind == X_PLANES:
            # digit planes of recode_planes for this key and batch: x_digits = their digit count
            need = self.lib.qf_operand_planes_bytes(key.h, B, x_digits)
            if B and (need == 0 or X.element_size() != 1 or not X.is_contiguous() or X.numel() < need):
                raise QfError(_lib.QF_ERR_DIM, "digit planes: a contiguous int8 buffer of qf_operand_planes_bytes(key, B, x_digits) bytes expected")
            stride = 0
        elif kind == X_BITS_KT:
            # k-tile-major bits (bits_to_kt): (ceil(cols/128), >= B, 16) uint8, planes X.stride(0) bytes apart
            if B and (X.dim() != 3 or X.shape[2] != 16 or X.stride(1) != 16 or X.stride(2) != 1 or X.element_size() != 1
                      or X.shape[0] < (key.cols + 127) // 128 or X.shape[1] < B):
                raise QfError(_lib.QF_ERR_DIM, "k-tile-major bits: (ceil(cols/128), >= B, 16) uint8 planes expected")
            stride = X.stride(0) if B else 0
        else:
            stride = X.stride(0) * X.element_size() if X.dim() == 2 else 0
            if B and X.dim() != 2:
                raise QfError(_lib.QF_ERR_DIM, "batch operand must be 2-D (message-major)")
        if addend is not None and tuple(addend.shape) != (B, key.rows):
            raise QfError(_lib.QF_ERR_DIM, "addend must be B x rows")
        rc = self.lib.qf_matmul_modq_dev(self.h, key.h, X.data_ptr() if B else None, kind, stride, x_digits, B,
                                         addend.data_ptr() if addend is not None else None,
                                         msg.data_ptr() if msg is not None else None, out.data_ptr())
        _check(self.h, rc)
        return out

    def recode_planes(self, key: "Key", X: torch.Tensor, kind: int, B: int, digits: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A wide operand (B x cols int32 rows, X_U32 / X_I32) -> the signed-digit planes kernel (a) consumes (operand kind
        X_PLANES, x_digits = digits).  Raises QfError(QF_ERR_RANGE) if an entry does not fit the promised digits."""
        assert X.is_cuda and X.dim() == 2 and X.element_size() == 4 and X.stride(1) == 1
        need = self.lib.qf_operand_planes_bytes(key.h, B, digits)
        if need == 0 and B:
            raise QfError(_lib.QF_ERR_ARG, "recode_planes: digits must be 1..digits of q")
        if out is None:
            out = torch.empty((need,), dtype=torch.int8, device=X.device)
        assert out.numel() >= need and out.element_size() == 1 and out.is_contiguous()
        _check(self.h, self.lib.qf_recode_planes_dev(self.h, key.h, X.data_ptr() if B else None, kind, X.stride(0) * 4, digits, B,
                                                     out.data_ptr() if B else None))
        return out

    def bits_to_kt(self, X: torch.Tensor, B: int, cols: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Bit rows (B, stride) -> k-tile-major planes (ceil(cols/128), B, 16): the operand form X_BITS_KT of matmul."""
        assert X.is_cuda and X.dim() == 2 and X.element_size() == 1 and X.stride(1) == 1
        nt = (cols + 127) // 128
        if out is None:
            out = torch.empty((nt, B, 16), dtype=torch.uint8, device=X.device)
        assert out.dim() == 3 and out.shape[0] >= nt and out.shape[1] >= B and out.shape[2] == 16 and out.stride(1) == 16
        _check(self.h, self.lib.qf_bits_to_kt_dev(self.h, X.data_ptr() if B else None, X.stride(0), B, cols,
                                                  out.data_ptr() if B else None, out.stride(0)))
        return out

    def matvec(self, key: "Key", x) -> np.ndarray:
        x = np.ascontiguousarray(np.asarray(x, dtype=np.uint32)).reshape(-1)
        if x.size != key.cols:
            raise QfError(_lib.QF_ERR_DIM, f"vector of length {x.size} for a key with {key.cols} columns")
        y = np.empty(key.rows, np.uint32)
        _check(self.h, self.lib.qf_matvec_modq_host(self.h, key.h, x.ctypes.data, y.ctypes.data))
        return y

    def vecmat(self, key: "Key", x) -> np.ndarray:
        x = np.ascontiguousarray(np.asarray(x, dtype=np.uint32)).reshape(-1)
        if x.size != key.rows:
            raise QfError(_lib.QF_ERR_DIM, f"vector of length {x.size} for a key with {key.rows} rows")
        y = np.empty(key.cols, np.uint32)
        _check(self.h, self.lib.qf_vecmat_modq_host(self.h, key.h, x.ctypes.data, y.ctypes.data))
        return y

    def vecmat_u32(self, A: torch.Tensor, q: int, x: torch.Tensor, addend: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x^t*A + addend^t mod q for a plain (rows, cols) device matrix -- the one-off product of key
        generation (regev.rs:366-368); everything stays on the device."""
        assert A.is_cuda and A.dtype == torch.int32 and A.is_contiguous() and A.dim() == 2
        rows, cols = A.shape
        if x.numel() != rows or (addend is not None and addend.numel() != cols):
            raise QfError(_lib.QF_ERR_DIM, "vecmat: x must have `rows` entries and addend `cols`")
        if out is None:
            out = self.empty_u32(cols)
        _check(self.h, self.lib.qf_vecmat_u32_dev(self.h, A.data_ptr(), rows, cols, q, x.data_ptr(),
                                                  addend.data_ptr() if addend is not None else None, out.data_ptr()))
        return out

    def pack_messages(self, messages):
        """UTF-8 bytes of all strings back to back + B+1 byte offsets, both on the device
        (the operand layout of qf_hash_to_mat_zq_sha256_dev)."""
        raw = [m.encode() if isinstance(m, str) else bytes(m) for m in messages]
        offs = np.zeros(len(raw) + 1, np.uint64)
        if raw:
            offs[1:] = np.cumsum([len(r) for r in raw])
        blob = np.frombuffer(b"".join(raw) or b"\0", np.uint8).copy()
        return self.to_dev_u8(blob), torch.from_numpy(offs.view(np.int64)).to(self.device), len(raw)

    def hash_packed(self, packed, rows: int, cols: int, q: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        d_blob, d_offs, B = packed
        if out is None:
            out = self.empty_u32(B, rows * cols)
        if B:
            _check(self.h, self.lib.qf_hash_to_mat_zq_sha256_dev(self.h, d_blob.data_ptr(), d_offs.data_ptr(), B, rows, cols, q,
                                                                 out.data_ptr()))
        return out

    def hash_to_mat_zq_sha256(self, messages, rows: int, cols: int, q: int) -> torch.Tensor:
        """hash/sha256.rs:110-133 for a list of strings at once: (B, rows*cols) residues on the device
        (rows = n, cols = 1: the ring target of HashMatPolynomialRingZq{1,1}, sha256.rs:249-256)."""
        return self.hash_packed(self.pack_messages(messages), rows, cols, q)

    def dot_decode(self, w: torch.Tensor, q: int, Cm: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = Cm.shape[0]
        n = w.numel()
        if Cm.dim() != 2 or Cm.shape[1] != n + 1:
            raise QfError(_lib.QF_ERR_DIM, "ciphertexts must be B x (n+1)")
        if out is None:
            out = self.empty_u8(B)
        _check(self.h, self.lib.qf_dot_decode_dev(self.h, w.data_ptr(), n, q, Cm.data_ptr(), B, out.data_ptr()))
        return out

    # ---- rings ---------------------------------------------------------------------------
    def ring(self, n: int, q: int) -> "Ring":
        h = C.c_void_p()
        _check(self.h, self.lib.qf_ring_create(self.h, n, q, C.byref(h)))
        return Ring(self, h)

    # ---- counter-mode generator -----------------------------------------------------------
    # `seed` is a 64-bit integer (SplitMix64: tests, benches, oracle parity) or a (key32, nonce) pair
    # (ChaCha20 under a 256-bit key: what an object constructed WITHOUT a seed draws from).
    @staticmethod
    def _keyed(seed):
        if isinstance(seed, tuple):
            key, nonce = seed
            key = bytes(key)
            if len(key) != 32:
                raise QfError(_lib.QF_ERR_ARG, "generator key must be 32 bytes")
            return key, int(nonce) & 0xFFFFFFFFFFFFFFFF
        return None

    def sample_uniform(self, seed, stream, first, count, q, out=None):
        out = out if out is not None else self.empty_u32(count)
        k = self._keyed(seed)
        if k:
            rc = self.lib.qf_sample_uniform_keyed_dev(self.h, k[0], k[1], stream, first, count, q, out.data_ptr())
        else:
            rc = self.lib.qf_sample_uniform_dev(self.h, seed, stream, first, count, q, out.data_ptr())
        _check(self.h, rc)
        return out

    def sample_bits(self, seed, stream, first_row, rows, cols, stride=None, out=None):
        stride = stride or bit_stride(cols)
        out = out if out is not None else self.empty_u8(rows, stride)
        k = self._keyed(seed)
        if k:
            rc = self.lib.qf_sample_bits_keyed_dev(self.h, k[0], k[1], stream, first_row, rows, cols, stride, out.data_ptr())
        else:
            rc = self.lib.qf_sample_bits_dev(self.h, seed, stream, first_row, rows, cols, stride, out.data_ptr())
        _check(self.h, rc)
        return out

    def sample_bits_kt(self, seed, stream, first_row, rows, cols, out=None):
        """The bit rows of sample_bits, written k-tile-major: (ceil(cols/128), rows, 16) uint8 (operand kind X_BITS_KT)."""
        nt = (cols + 127) // 128
        if out is None:
            out = torch.empty((nt, rows, 16), dtype=torch.uint8, device=self.device)
        assert out.dim() == 3 and out.shape[0] >= nt and out.shape[1] >= rows and out.shape[2] == 16 and out.stride(1) == 16
        k = self._keyed(seed)
        if k:
            rc = self.lib.qf_sample_bits_kt_keyed_dev(self.h, k[0], k[1], stream, first_row, rows, cols, out.stride(0), out.data_ptr())
        else:
            rc = self.lib.qf_sample_bits_kt_dev(self.h, seed, stream, first_row, rows, cols, out.stride(0), out.data_ptr())
        _check(self.h, rc)
        return out

    def sample_noise(self, seed, stream, first, count, q, W, out=None):
        out = out if out is not None else self.empty_u32(count)
        k = self._keyed(seed)
        if k:
            rc = self.lib.qf_sample_noise_keyed_dev(self.h, k[0], k[1], stream, first, count, q, W, out.data_ptr())
        else:
            rc = self.lib.qf_sample_noise_dev(self.h, seed, stream, first, count, q, W, out.data_ptr())
        _check(self.h, rc)
        return out

    def sample_discrete_gauss(self, seed, stream, first, count, q, n, s, out=None):
        """D_{Z,s} (center 0) mod q, the distribution of MatZq::sample_discrete_gauss(.., q, n, 0, s)
        (regev.rs:356-363, lpr.rs:342-360): drawn on the device."""
        out = out if out is not None else self.empty_u32(count)
        k = self._keyed(seed)
        if k:
            rc = self.lib.qf_sample_discrete_gauss_keyed_dev(self.h, k[0], k[1], stream, first, count, q, int(n), float(s), out.data_ptr())
        else:
            rc = self.lib.qf_sample_discrete_gauss_dev(self.h, seed, stream, first, count, q, int(n), float(s), out.data_ptr())
        _check(self.h, rc)
        return out


class Key:
    """A MatZq resident on the device (qf_key)."""

    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h
        r, c, q = C.c_uint32(), C.c_uint32(), C.c_uint32()
        ctx.lib.qf_key_dims(h, C.byref(r), C.byref(c), C.byref(q))
        self.rows, self.cols, self.q = r.value, c.value, q.value

    def close(self):
        if self.h:
            self.ctx.lib.qf_key_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass


class Ring:
    """R_q = Z_q[X]/(X^n+1) (qf_ring); replaces utils/common_moduli.rs new_anticyclic."""

    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h
        n, q, ntt = C.c_uint32(), C.c_uint32(), C.c_int()
        ctx.lib.qf_ring_info(h, C.byref(n), C.byref(q), C.byref(ntt))
        self.n, self.q, self.uses_ntt = n.value, q.value, bool(ntt.value)

    def key(self, polys) -> "RingKey":
        h = C.c_void_p()
        if isinstance(polys, torch.Tensor):
            assert polys.is_cuda and polys.dtype == torch.int32 and polys.is_contiguous()
            p2 = polys.reshape(-1, self.n)
            rc = self.ctx.lib.qf_rkey_upload_dev(self.ctx.h, self.h, p2.data_ptr(), p2.shape[0], C.byref(h))
            npoly = p2.shape[0]
        else:
            p = np.ascontiguousarray(np.asarray(polys, dtype=np.uint32)).reshape(-1, self.n)
            rc = self.ctx.lib.qf_rkey_upload_host(self.ctx.h, self.h, p.ctypes.data, p.shape[0], C.byref(h))
            npoly = p.shape[0]
        _check(self.ctx.h, rc)
        return RingKey(self, h, npoly)

    def polymul(self, a: torch.Tensor, b: torch.Tensor, out=None) -> torch.Tensor:
        if a.shape != b.shape or a.shape[-1] != self.n:
            raise QfError(_lib.QF_ERR_DIM, "polymul operands must both be B x n")
        B = a.numel() // self.n
        out = out if out is not None else torch.empty_like(a)
        _check(self.ctx.h, self.ctx.lib.qf_polymul_dev(self.ctx.h, self.h, a.data_ptr(), b.data_ptr(), B, out.data_ptr()))
        return out

    def ringlpr_enc(self, pk: "RingKey", r, e1, e2, mu_bits, u=None, v=None):
        B = r.shape[0]
        u = u if u is not None else torch.empty_like(r)
        v = v if v is not None else torch.empty_like(r)
        _check(self.ctx.h, self.ctx.lib.qf_ringlpr_enc_dev(self.ctx.h, self.h, pk.h, r.data_ptr(), e1.data_ptr(),
                                                          e2.data_ptr(), mu_bits.data_ptr(), B, u.data_ptr(), v.data_ptr()))
        return u, v

    def ringlpr_dec(self, sk: "RingKey", u, v, out=None):
        B = u.shape[0]
        out = out if out is not None else self.ctx.empty_u8(B, (self.n + 7) // 8)
        _check(self.ctx.h, self.ctx.lib.qf_ringlpr_dec_dev(self.ctx.h, self.h, sk.h, u.data_ptr(), v.data_ptr(), B, out.data_ptr()))
        return out

    def fdh_verify(self, a: "RingKey", sigma: torch.Tensor, u: torch.Tensor, norm_bound: int, want_t: bool = False):
        B = sigma.shape[0]
        if tuple(sigma.shape) != (B, a.npoly, self.n) or tuple(u.shape) != (B, self.n):
            raise QfError(_lib.QF_ERR_DIM, "sigma must be B x mbar x n and u must be B x n")
        ok = self.ctx.empty_u8(B)
        t = self.ctx.empty_u32(B, self.n) if want_t else None
        _check(self.ctx.h, self.ctx.lib.qf_fdh_ring_verify_dev(self.ctx.h, self.h, a.h, sigma.data_ptr(), u.data_ptr(),
                                                               norm_bound, B, t.data_ptr() if want_t else None, ok.data_ptr()))
        return (ok, t) if want_t else ok

    def close(self):
        if self.h:
            self.ctx.lib.qf_ring_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass


class RingKey:
    def __init__(self, ring: Ring, h, npoly: int):
        self.ring, self.h, self.npoly = ring, h, npoly

    def close(self):
        if self.h:
            self.ring.ctx.lib.qf_rkey_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ring.ctx.h:
                self.close()
        except Exception:
            pass
