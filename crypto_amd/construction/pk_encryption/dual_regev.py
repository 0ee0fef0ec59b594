"""Dual Regev public-key encryption -- mirror of construction/pk_encryption/dual_regev.rs."""
from __future__ import annotations

import numpy as np

from ...device import X_U32
from ...math import MatZ, MatZq, gaussian_width
from ._matrix_scheme import message_bit
from .regev import Regev


class DualRegev(Regev):
    """Same public parameters and checks as Regev (dual_regev.rs:122-301 repeats regev.rs);
    pk = [A | u] with u = A*x, sk = x in {0,1}^m (a MatZ)."""

    @classmethod
    def secure128(cls, **kw):
        return cls(230, 5313, 7764299, 0.0011, **kw)      # dual_regev.rs (same set as Regev)

    @classmethod
    def default(cls, **kw):
        return cls(13, 154, 1427, 0.01, **kw)

    def gen(self):
        """dual_regev.rs:347-361."""
        n, m, q = self.n, self.m, self.q
        A = self.rng.uniform(1, n * m, q).reshape(n, m)
        x = self.rng.bits(2, m)
        key_a = self.ctx.key_upload(A, q)
        u = self.ctx.matvec(key_a, x)
        key_a.close()
        return MatZq(np.hstack([A, u[:, None]]), q), MatZ(x.astype(np.int64)[:, None])

    def enc(self, pk: MatZq, message, s=None, e=None) -> MatZq:
        """dual_regev.rs:385-413: c^t = s^t*pk + e^t, message into the last entry."""
        self._check_pk(pk, self.n, self.m + 1)
        q = self.q
        if s is None:
            s = self.rng.uniform(3, self.n, q)
        if e is None:
            e = self.rng.noise(4, self.m + 1, q, self.noise_width)
        c = (self.ctx.vecmat(self._dev_key(pk), np.asarray(s, np.uint32)).astype(np.uint64) + np.asarray(e, np.uint64)) % q
        c[-1] = (int(c[-1]) + message_bit(message) * (q // 2)) % q
        return MatZq(c.astype(np.uint32)[:, None], q)

    def dec(self, sk: MatZ, cipher: MatZq) -> int:
        """dual_regev.rs:437-450."""
        return int(self.dec_batch(sk, cipher.entries.reshape(1, -1))[0])

    def enc_batch(self, pk: MatZq, messages, s=None, e=None):
        """Batched: C = S * pk + E with the key uploaded transposed (kernel a at full width)."""
        self._check_pk(pk, self.n, self.m + 1)
        ctx, q = self.ctx, self.q
        msg = np.array([message_bit(v) for v in np.asarray(messages).reshape(-1)], np.uint8)
        B = msg.size
        if s is None:
            s = ctx.sample_uniform(self.rng.seed, 3, self.rng._take(3, B * self.n), B * self.n, q).view(B, self.n)
        elif isinstance(s, np.ndarray):
            s = ctx.to_dev_u32(s)
        if e is None:
            cnt = B * (self.m + 1)
            e = self.rng.noise_dev(4, cnt, q, self.noise_width).view(B, self.m + 1)
        elif isinstance(e, np.ndarray):
            e = ctx.to_dev_u32(e)
        return ctx.matmul(self._dev_key(pk, transpose=True), s, X_U32, B, addend=e, msg=ctx.to_dev_u8(msg))

    def dec_batch(self, sk: MatZ, ciphers) -> np.ndarray:
        x = sk.entries.reshape(-1)
        if x.size != self.m:
            raise ValueError("mismatching secret key dimension")
        w = np.mod(-x, self.q).astype(np.uint32)
        return self._decode_batch(w, ciphers)
