"""RingLPR public-key encryption -- mirror of construction/pk_encryption/ring_lpr.rs.
All products are negacyclic R_q products on kernel (b)."""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from ...math import MathError, PolynomialRingZq, as_fraction, gaussian_width
from ...runtime import Randomness, default_context
from ...utils.common_moduli import new_anticyclic
from .base import PKEncryptionScheme, largest_prime_at_most
from .lpr import _factor
from ._matrix_scheme import set_noise_model


class RingLPR(PKEncryptionScheme):
    def __init__(self, n, q, alpha, ctx=None, seed=None, noise_model=None):
        """RingLPR::new (ring_lpr.rs:90-104): modulus (X^n + 1) mod q."""
        self.n, self.q_int = int(n), int(q)
        self.q = new_anticyclic(self.n, self.q_int)
        self.alpha = as_fraction(alpha)
        self._ctx = ctx
        self.rng = Randomness(seed, ctx)
        self._ring = None
        set_noise_model(self, noise_model)

    new = classmethod(lambda cls, n, q, alpha, **kw: cls(n, q, alpha, **kw))

    @property
    def ctx(self):
        return self._ctx or default_context()

    @property
    def ring(self):
        if self._ring is None or self._ring.h is None:
            self._ring = self.ctx.ring(self.n, self.q_int)
        return self._ring

    @classmethod
    def new_from_n(cls, n: int, **kw) -> "RingLPR":
        """ring_lpr.rs:118-144."""
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
            except MathError as err:
                if "power of 2" in str(err):
                    raise
                upper = q - 1

    @staticmethod
    def gen_new_public_parameters(n: int, upper=None):
        """ring_lpr.rs:169-197."""
        ub = n ** 3
        q = largest_prime_at_most(ub if upper is None else upper, -(-ub // 2))
        alpha = Fraction(1 / (_factor(n) * math.sqrt(n) * math.log2(n) ** 3))
        return q, alpha

    def check_correctness(self):
        """ring_lpr.rs:228-270."""
        if self.n <= 1:
            raise MathError("n must be chosen bigger than 1.")
        if self.n & (self.n - 1):
            raise MathError("n is not a perfect power of 2, which is required for the correctness of this scheme.")
        if float(self.alpha) > 1 / (_factor(self.n) * math.sqrt(self.n) * math.log2(self.n) ** 3):
            raise MathError("Correctness is not guaranteed as alpha >= 1 / (sqrt(n) * log^3 n)")

    def check_security(self):
        """ring_lpr.rs:294-306."""
        if self.q_int * float(self.alpha) < 2 * math.sqrt(self.n):
            raise MathError("Security is not guaranteed as q * alpha < 2 * sqrt(n)")

    @classmethod
    def secure128(cls, **kw):
        return cls(512, 92897729, 0.000005, **kw)   # ring_lpr.rs:312-314

    @classmethod
    def default(cls, **kw):
        return cls(16, 2399, 0.0039, **kw)          # ring_lpr.rs:343-345

    @property
    def noise_width(self) -> int:
        return gaussian_width(self.alpha * self.q_int)

    def _poly(self, coeffs) -> PolynomialRingZq:
        return PolynomialRingZq(coeffs, self.n, self.q_int)

    def mu_bits(self, message) -> np.ndarray:
        """ring_lpr.rs:424-427,318-329: bits of |message| mod 2^n, bit i -> coefficient i, packed."""
        mu = abs(int(message)) % (1 << self.n)
        return np.frombuffer(mu.to_bytes((self.n + 7) // 8, "little"), np.uint8).copy()

    def gen(self):
        """ring_lpr.rs:370-395: a <- R_q, s,e <- chi, b = a*s + e; pk = (a, b), sk = s."""
        n, q, W = self.n, self.q_int, self.noise_width
        ctx = self.ctx
        a = self.rng.uniform(1, n, q)
        s = self.rng.noise(2, n, q, W)
        e = self.rng.noise(4, n, q, W)
        prod = ctx.to_host_u32(self.ring.polymul(ctx.to_dev_u32(a[None]), ctx.to_dev_u32(s[None])))[0]
        b = ((prod.astype(np.uint64) + e) % q).astype(np.uint32)
        return (self._poly(a), self._poly(b)), self._poly(s)

    def enc_batch(self, pk, messages, r=None, e1=None, e2=None):
        """ring_lpr.rs:422-461 for B messages: returns device tensors (u, v), each (B, n)."""
        ctx, n, q, W = self.ctx, self.n, self.q_int, self.noise_width
        mu = np.stack([self.mu_bits(m) for m in messages])
        B = mu.shape[0]

        def draw(stream, given):
            if given is None:
                return self.rng.noise_dev(stream, B * n, q, W).view(B, n)
            return ctx.to_dev_u32(given) if isinstance(given, np.ndarray) else given
        r, e1, e2 = draw(3, r), draw(4, e1), draw(4, e2)
        pk_key = self.ring.key(np.stack([pk[0].coeffs, pk[1].coeffs]))
        return self.ring.ringlpr_enc(pk_key, r, e1, e2, ctx.to_dev_u8(mu))

    def enc(self, pk, message, r=None, e1=None, e2=None):
        rr = None if r is None else np.asarray(r, np.uint32)[None]
        ee1 = None if e1 is None else np.asarray(e1, np.uint32)[None]
        ee2 = None if e2 is None else np.asarray(e2, np.uint32)[None]
        u, v = self.enc_batch(pk, [message], rr, ee1, ee2)
        u, v = self.ctx.to_host_u32(u)[0], self.ctx.to_host_u32(v)[0]
        return self._poly(u), self._poly(v)

    def dec_batch(self, sk, u, v) -> list:
        """ring_lpr.rs:489-508 for B ciphertexts -> list of Python ints."""
        ctx = self.ctx
        u = ctx.to_dev_u32(u) if isinstance(u, np.ndarray) else u
        v = ctx.to_dev_u32(v) if isinstance(v, np.ndarray) else v
        sk_key = self.ring.key(sk.coeffs[None])
        packed = ctx.to_host_u8(self.ring.ringlpr_dec(sk_key, u, v))
        return [int.from_bytes(row.tobytes(), "little") for row in packed]

    def dec(self, sk, cipher) -> int:
        return self.dec_batch(sk, cipher[0].coeffs[None], cipher[1].coeffs[None])[0]
