"""LPR public-key encryption -- mirror of construction/pk_encryption/lpr.rs."""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from ...device import X_U32
from ...math import MatZq, MathError, as_fraction, gaussian_width
from ._matrix_scheme import MatrixSchemeMixin, message_bit
from .base import GenericMultiBitEncryption, largest_prime_at_most


def _factor(n: int) -> int:
    """lpr.rs:178-184: the experience factor ~ max(log n - 4, 1)."""
    return 1 if n <= 20 else 2 if n <= 40 else 3 if n <= 80 else 4 if n <= 160 else 5


class LPR(MatrixSchemeMixin, GenericMultiBitEncryption):
    def __init__(self, n, q, alpha, ctx=None, seed=None, noise_model=None):
        self.n, self.q = int(n), int(q)
        if self.q <= 1:
            raise ValueError("modulus must be > 1")
        self.alpha = as_fraction(alpha)
        self._init_runtime(ctx, seed, noise_model)

    new = classmethod(lambda cls, n, q, alpha, **kw: cls(n, q, alpha, **kw))

    @classmethod
    def new_from_n(cls, n: int, **kw) -> "LPR":
        """lpr.rs:113-139."""
        n = int(n)
        if n < 10:
            raise ValueError("Choose n >= 10 as this function does not return parameters ensuring "
                             "proper correctness of the scheme otherwise.")
        upper = None
        while True:
            q, alpha = cls.gen_new_public_parameters(n, upper)
            out = cls(n, q, alpha, **kw)
            try:
                out.check_correctness()
                out.check_security()
                return out
            except MathError:
                upper = q - 1

    @staticmethod
    def gen_new_public_parameters(n: int, upper=None):
        """lpr.rs:164-191."""
        ub = n ** 3
        q = largest_prime_at_most(ub if upper is None else upper, -(-ub // 2))
        alpha = Fraction(1 / (_factor(n) * math.sqrt(n) * math.log2(n) ** 3))
        return q, alpha

    def check_correctness(self):
        """lpr.rs:221-248."""
        if self.n <= 1:
            raise MathError("n must be chosen bigger than 1.")
        if float(self.alpha) > 1 / (_factor(self.n) * math.sqrt(self.n) * math.log2(self.n) ** 3):
            raise MathError("Correctness is not guaranteed as alpha >= 1 / (sqrt(n) * log^3 n)")

    def check_security(self):
        """lpr.rs:272-284."""
        if self.q * float(self.alpha) < 2 * math.sqrt(self.n):
            raise MathError("Security is not guaranteed as q * alpha < 2 * sqrt(n)")

    @classmethod
    def secure128(cls, **kw):
        return cls(500, 76859609, 0.000005, **kw)   # lpr.rs:290-292

    @classmethod
    def default(cls, **kw):
        return cls(10, 983, 0.0072, **kw)           # lpr.rs:306-312

    @property
    def noise_width(self) -> int:
        return gaussian_width(self.alpha * self.q)

    @property
    def noise_digits(self) -> int:
        """signed 8-bit digits that cover the noise model's range (4W, or ceil(s log2 n) for D_{Z,s})."""
        bound = self.rng.noise_bound(self.noise_width)
        d = 1
        while 127 * (256 ** d - 1) // 255 < bound:
            d += 1
        return d

    def gen(self):
        """lpr.rs:338-370: A <- Z_q^{n x n}, s,e <- chi^n, b^t = s^t A + e^t, pk = [A ; b^t]."""
        return self._lwe_keypair(self.n, self.n, secret_from_chi=True)

    def enc(self, pk: MatZq, message, r=None, e=None) -> MatZq:
        """lpr.rs:394-430: c = pk*r + e + [0..0, msg*floor(q/2)]^t."""
        self._check_pk(pk, self.n + 1, self.n)
        q, W = self.q, self.noise_width
        if r is None:
            r = self.rng.noise(3, self.n, q, W)
        if e is None:
            e = self.rng.noise(4, self.n + 1, q, W)
        c = (self.ctx.matvec(self._dev_key(pk), np.asarray(r, np.uint32)).astype(np.uint64) + np.asarray(e, np.uint64)) % q
        c[-1] = (int(c[-1]) + message_bit(message) * (q // 2)) % q
        return MatZq(c.astype(np.uint32)[:, None], q)

    def dec(self, sk: MatZq, cipher: MatZq) -> int:
        """lpr.rs:454-468."""
        return int(self.dec_batch(sk, cipher.entries.reshape(1, -1))[0])

    def enc_batch(self, pk: MatZq, messages, r=None, e=None):
        """B encryptions on kernel (a).  r: (B, n), e: (B, n+1) device int32 tensors or numpy."""
        self._check_pk(pk, self.n + 1, self.n)
        ctx, q, W = self.ctx, self.q, self.noise_width
        msg = np.array([message_bit(v) for v in np.asarray(messages).reshape(-1)], np.uint8)
        B = msg.size
        if r is None:
            r = self.rng.noise_dev(3, B * self.n, q, W).view(B, self.n)
        elif isinstance(r, np.ndarray):
            r = ctx.to_dev_u32(r)
        if e is None:
            e = self.rng.noise_dev(4, B * (self.n + 1), q, W).view(B, self.n + 1)
        elif isinstance(e, np.ndarray):
            e = ctx.to_dev_u32(e)
        return ctx.matmul(self._dev_key(pk), r, X_U32, B, addend=e, msg=ctx.to_dev_u8(msg), x_digits=self.noise_digits)

    def dec_batch(self, sk: MatZq, ciphers) -> np.ndarray:
        s = sk.entries.reshape(-1).astype(np.int64)
        if s.size != self.n:
            raise ValueError("mismatching secret key dimension")
        return self._decode_batch(((self.q - s) % self.q).astype(np.uint32), ciphers)
