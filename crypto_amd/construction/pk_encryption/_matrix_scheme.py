"""Shared plumbing of the three matrix schemes (Regev, DualRegev, LPR): device key caching,
single-message products on kernel (c), batched products on kernel (a), batched decryption."""
from __future__ import annotations

import numpy as np

from ...device import X_BITS, X_U32, bit_stride, pack_bits
from ...math import MatZ, MatZq
from ...runtime import Randomness, default_context


class MatrixSchemeMixin:
    def _init_runtime(self, ctx=None, seed=None, noise_model=None):
        self._ctx = ctx
        self.rng = Randomness(seed, ctx)   # seed None -> a fresh os.urandom seed per object
        set_noise_model(self, noise_model)

    @property
    def ctx(self):
        return self._ctx or default_context()

    def _dev_key(self, mat: MatZq, transpose: bool = False):
        """Upload (once) and cache the digit planes of a public matrix."""
        tag = (id(self.ctx), transpose)
        key = mat._dev.get(tag)
        if key is None or key.h is None:
            key = self.ctx.key_upload(mat.entries, mat.q, transpose=transpose)
            mat._dev[tag] = key
        return key

    def _check_pk(self, pk: MatZq, rows: int, cols: int):
        # the reference panics on mismatching dimensions / moduli inside qfall-math
        if not isinstance(pk, MatZq):
            raise TypeError("public key must be a MatZq")
        if pk.q != self.q:
            raise ValueError("mismatching modulus between scheme and key")
        if pk.entries.shape != (rows, cols):
            raise ValueError(f"mismatching matrix dimensions: expected {rows}x{cols}, got {pk.entries.shape}")

    def _decode_batch(self, w: np.ndarray, ciphers) -> np.ndarray:
        """bits = decision rule applied to <w, c[:-1]> + c[-1] for each row of `ciphers`."""
        c = ciphers if not isinstance(ciphers, np.ndarray) else self.ctx.to_dev_u32(ciphers)
        bits = self.ctx.dot_decode(self.ctx.to_dev_u32(w), self.q, c)
        return self.ctx.to_host_u8(bits)


def set_noise_model(scheme, noise_model):
    """noise_model: None / "discrete_gauss" (default: D_{Z, alpha*q} with security parameter n, as the
    reference samples chi -- regev.rs:356-363, lpr.rs:342-360, ring_lpr.rs:374-388) or "irwin_hall"
    (the bounded stand-in of the benchmark generator, SURVEY 8d: explicit opt-in)."""
    if noise_model == "irwin_hall":
        scheme.rng.gauss = None
    elif noise_model in (None, "discrete_gauss"):
        scheme.rng.gauss = (int(scheme.n), float(scheme.alpha * getattr(scheme, "q_int", scheme.q)))
    else:
        raise ValueError("noise_model must be 'irwin_hall' or 'discrete_gauss'")


def message_bit(message) -> int:
    """message.modulo(2): Euclidean remainder, also for negative values (regev.rs:398,598-612)."""
    return int(message) % 2
