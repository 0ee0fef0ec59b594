"""Regev public-key encryption -- mirror of construction/pk_encryption/regev.rs.

Arithmetic runs on the GPU library: `enc` is one HBM-streaming mat-vec (kernel c, regev.rs:404),
`enc_batch` is the int8-MFMA mod-q GEMM (kernel a), `dec`/`dec_batch` the dot-product decoder
(regev.rs:438-452).  Randomness comes from the seeded generator instead of the reference's
unseeded samplers; pass explicit randomness to reproduce a ciphertext bit for bit.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from ...device import X_BITS, X_BITS_KT, bit_stride, pack_bits
from ...math import MatZq, MathError, as_fraction, gaussian_width, log2_ceil
from ._matrix_scheme import MatrixSchemeMixin, message_bit
from .base import GenericMultiBitEncryption, largest_prime_at_most


class Regev(MatrixSchemeMixin, GenericMultiBitEncryption):
    def __init__(self, n, m, q, alpha, ctx=None, seed=None, noise_model=None):
        """Regev::new (regev.rs:87-99).  Parameters are not checked (as in the reference)."""
        self.n, self.m, self.q = int(n), int(m), int(q)
        if self.q <= 1:
            raise ValueError("modulus must be > 1")
        self.alpha = as_fraction(alpha)
        self._init_runtime(ctx, seed, noise_model)

    new = classmethod(lambda cls, n, m, q, alpha, **kw: cls(n, m, q, alpha, **kw))

    # ---- parameter generation (regev.rs:122-201) ------------------------------------------
    @classmethod
    def new_from_n(cls, n: int, **kw) -> "Regev":
        n = int(n)
        if n < 10:
            raise ValueError("Choose n >= 10 as this function does not return parameters ensuring "
                             "proper correctness of the scheme otherwise.")
        upper = None
        while True:
            m, q, alpha = cls.gen_new_public_parameters(n, upper)
            out = cls(n, m, q, alpha, **kw)
            try:
                out.check_correctness()
                out.check_security()
                return out
            except MathError:
                upper = q - 1  # next prime below (the reference re-samples a random prime)

    @staticmethod
    def gen_new_public_parameters(n: int, upper=None):
        power = 5 if 2 <= n <= 4 else 4 if n == 5 else 3
        ub = n ** power
        lb = -(-ub // 2)
        q = largest_prime_at_most(ub if upper is None else upper, lb)
        m = (n + 1) * log2_ceil(q)                     # (n+1) * ceil(log2 q), regev.rs:193
        alpha = Fraction(1 / (2 * math.sqrt(n) * math.log2(n) ** 2))
        return m, q, alpha

    def check_correctness(self):
        """regev.rs:229-253.  Raises MathError where the reference returns Err."""
        if self.n <= 1:
            raise MathError("n must be chosen bigger than 1.")
        if float(self.alpha) > 1 / (math.sqrt(self.n) * math.log2(self.n)):
            raise MathError("Correctness is not guaranteed as alpha >= 1 / (sqrt(n) * log n)")
        if 20 * math.sqrt(self.m) * float(self.alpha) > self.q:
            raise MathError("Correctness is not guaranteed as 5 * sqrt(m) * alpha > q/4")

    def check_security(self):
        """regev.rs:277-295."""
        if self.q * float(self.alpha) < 2 * math.sqrt(self.n):
            raise MathError("Security is not guaranteed as q * alpha < 2 * sqrt(n)")
        if self.m <= math.ceil((self.n + 1) * math.log2(self.q)):
            raise MathError("Security is not guaranteed as m <= (n + 1) log q")

    @classmethod
    def secure128(cls, **kw):
        return cls(230, 5313, 7764299, 0.0011, **kw)      # regev.rs:301-303

    @classmethod
    def default(cls, **kw):
        return cls(13, 154, 1427, 0.01, **kw)             # regev.rs:317-324

    # ---- the scheme --------------------------------------------------------------------------
    @property
    def noise_width(self) -> int:
        return gaussian_width(self.alpha * self.q)

    def gen(self):
        """regev.rs:350-374: A <- Z_q^{n x m}, s <- Z_q^n, e <- chi^m, b^t = s^t A + e^t, pk = [A ; b^t]."""
        return self._lwe_keypair(self.n, self.m, secret_from_chi=False)

    def enc(self, pk: MatZq, message, x=None) -> MatZq:
        """regev.rs:396-414: c = pk*x + [0..0, (message mod 2)*floor(q/2)]^t, x <- {0,1}^m."""
        self._check_pk(pk, self.n + 1, self.m)
        msg = message_bit(message)
        if x is None:
            x = self.rng.bits(3, self.m)
        c = self.ctx.matvec(self._dev_key(pk), np.asarray(x, np.uint32))
        c[-1] = (int(c[-1]) + msg * (self.q // 2)) % self.q
        return MatZq.from_canonical(c.reshape(-1, 1), self.q)

    def dec(self, sk: MatZq, cipher: MatZq) -> int:
        """regev.rs:438-452."""
        return int(self.dec_batch(sk, cipher.entries.reshape(1, -1))[0])

    # ---- batched forms (the GPU hot path) ------------------------------------------------
    def enc_batch(self, pk: MatZq, messages, x_bits=None):
        """B encryptions at once.  messages: B ints/bits; x_bits: optional (B, stride) packed
        bit rows on the device (or numpy), else drawn from stream 3.  Returns a device tensor
        (B, n+1) holding one ciphertext per row."""
        self._check_pk(pk, self.n + 1, self.m)
        ctx = self.ctx
        msg = np.array([message_bit(v) for v in np.asarray(messages).reshape(-1)], np.uint8)
        B = msg.size
        if x_bits is None:
            # x drawn here (regev.rs:401): the same bits as sample_bits, written k-tile-major -- the form kernel (a) reads fastest
            first = self.rng._take(3, B)
            x_kt = ctx.sample_bits_kt(self.rng.seed, 3, first, B, self.m)
            return ctx.matmul(self._dev_key(pk), x_kt, X_BITS_KT, B, msg=ctx.to_dev_u8(msg))
        elif isinstance(x_bits, np.ndarray):
            x_bits = ctx.to_dev_u8(x_bits if x_bits.shape[1] == bit_stride(self.m) else pack_bits(x_bits))
        return ctx.matmul(self._dev_key(pk), x_bits, X_BITS, B, msg=ctx.to_dev_u8(msg))

    def dec_batch(self, sk: MatZq, ciphers) -> np.ndarray:
        s = sk.entries.reshape(-1).astype(np.int64)
        if s.size != self.n:
            raise ValueError("mismatching secret key dimension")
        w = ((self.q - s) % self.q).astype(np.uint32)
        return self._decode_batch(w, ciphers)
