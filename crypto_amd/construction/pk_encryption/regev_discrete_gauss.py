"""Regev / dual Regev with discrete Gaussian regularity -- mirrors of
construction/pk_encryption/regev_discrete_gauss.rs and dual_regev_discrete_gauss.rs.

Same kernels as Regev / DualRegev: enc of the first is K*e for the (n+1) x m matrix K = [A ; p^t]
with a short Gaussian e (SampleD over Z^m with the identity basis = m independent D_{Z,r} draws),
enc of the dual is s^t*[A | u] + x^t.  chi and SampleD are drawn on the device with
qf_sample_discrete_gauss_dev (seeded; the reference's RNG is not)."""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from ...device import X_U32
from ...math import MatZq, MathError, as_fraction
from ._matrix_scheme import MatrixSchemeMixin, message_bit
from .base import GenericMultiBitEncryption, largest_prime_at_most


def _digits_for(bound: int) -> int:
    d = 1
    while 127 * (256 ** d - 1) // 255 < bound:
        d += 1
    return d


class _GaussianRegularityBase(MatrixSchemeMixin, GenericMultiBitEncryption):
    def __init__(self, n, m, q, r, alpha, ctx=None, seed=None):
        """::new (regev_discrete_gauss.rs:91-105, dual_regev_discrete_gauss.rs:91-105); parameters unchecked."""
        self.n, self.m, self.q = int(n), int(m), int(q)
        if self.q <= 1:
            raise ValueError("modulus must be > 1")
        self.r, self.alpha = as_fraction(r), as_fraction(alpha)
        self._init_runtime(ctx, seed)

    new = classmethod(lambda cls, n, m, q, r, alpha, **kw: cls(n, m, q, r, alpha, **kw))

    @classmethod
    def new_from_n(cls, n: int, **kw):
        """regev_discrete_gauss.rs:128-159 (dual: same).  "random prime" -> largest prime of the range, then
        the next smaller one while the checks fail."""
        n = int(n)
        if n <= 1:
            raise ValueError("n must be chosen bigger than 1.")
        upper = None
        while True:
            try:
                m, q, r, alpha = cls.gen_new_public_parameters(n, upper)
            except MathError:
                # n = 31..36: no prime of [n^3/2, n^3] satisfies q*alpha >= n -- the reference's loop
                # (regev_discrete_gauss.rs:144-156) would sample forever; report it instead
                raise MathError(f"no prime q in the range new_from_n({n}) draws from passes check_correctness and check_security")
            out = cls(n, m, q, r, alpha, **kw)
            try:
                out.check_correctness()
                out.check_security()
                return out
            except MathError:
                upper = q - 1

    @staticmethod
    def gen_new_public_parameters(n: int, upper=None):
        """regev_discrete_gauss.rs:161-199: q prime in [n^p/2, n^p], m = ceil(2(n+1) lg q), r = log2 m,
        alpha = 1/(sqrt(m) log2^2 m).  (`q.log(10)`: the reference's lg is the base-10 logarithm here.)"""
        power = 9 if n == 2 else 8 if n == 3 else 7 if n <= 5 else 6 if n <= 8 else 5 if n <= 12 else 4 if n <= 30 else 3
        ub = n ** power
        q = largest_prime_at_most(ub if upper is None else upper, -(-ub // 2))
        m = math.ceil(2 * (n + 1) * math.log10(q))
        r = Fraction(math.log2(m))
        alpha = Fraction(1 / (math.sqrt(m) * math.log2(m) ** 2))
        return m, q, r, alpha

    def check_correctness(self):
        """regev_discrete_gauss.rs:222-246."""
        if self.n <= 1:
            raise MathError("n must be chosen bigger than 1.")
        if self.q < 5 * float(self.r) * self.m:
            raise MathError("Correctness is not guaranteed as q < 5rm, but q >= 5rm is required.")
        if float(self.alpha) > 1 / (float(self.r) * math.sqrt(self.m) * math.sqrt(math.log2(self.n))):
            raise MathError("Correctness is not guaranteed as alpha > 1/(r*sqrt(m)*w(sqrt(log n))")

    def check_security(self):
        """regev_discrete_gauss.rs:270-297."""
        if self.q * float(self.alpha) < self.n:
            raise MathError("Security is not guaranteed as q * alpha < n, but q * alpha >= n is required.")
        if self.m < 2 * (self.n + 1) * math.log10(self.q):
            raise MathError("Security is not guaranteed as m < 2(n + 1) lg (q)")
        if float(self.r) < math.sqrt(math.log2(self.m)):
            raise MathError("Security is not guaranteed as r < sqrt( log m )")

    @classmethod
    def secure128(cls, **kw):
        return cls(350, 5248, 29892991, 12.357, 0.00009, **kw)     # regev_discrete_gauss.rs:312-314

    @classmethod
    def default(cls, **kw):
        return cls(2, 16, 443, 4, Fraction(1, 64), **kw)           # regev_discrete_gauss.rs:329-338

    # ---- sampling ---------------------------------------------------------------------------
    def _chi(self, stream, count):
        """x <- chi = D_{Z, alpha*q} with security parameter n (regev_discrete_gauss.rs:376-384)."""
        first = self.rng._take(stream, count)
        return self.ctx.sample_discrete_gauss(self.rng.seed, stream, first, count, self.q, self.n, float(self.alpha * self.q))

    def _sample_d(self, stream, count):
        """MatZq::sample_d_common(m, q, n, r): SampleD over Z^m (identity basis), center 0 = independent
        D_{Z,r} coordinates (regev_discrete_gauss.rs:419)."""
        first = self.rng._take(stream, count)
        return self.ctx.sample_discrete_gauss(self.rng.seed, stream, first, count, self.q, self.n, float(self.r))

    @property
    def _e_digits(self) -> int:
        return _digits_for(math.ceil(float(self.r) * math.log2(self.n)))


class RegevWithDiscreteGaussianRegularity(_GaussianRegularityBase):
    """pk = (A, p = A^t s + x), sk = s; cipher = (u = A e, c = p^t e + msg*floor(q/2))."""

    def gen(self):
        """regev_discrete_gauss.rs:369-393."""
        n, m, q = self.n, self.m, self.q
        s = self.rng.uniform(2, n, q)
        A = self.rng.uniform(1, n * m, q).reshape(n, m)
        x = self.ctx.to_host_u32(self._chi(4, m))
        key_a = self.ctx.key_upload(A, q)
        p = (self.ctx.vecmat(key_a, s).astype(np.uint64) + x) % q
        key_a.close()
        return (MatZq(A, q), MatZq(p.astype(np.uint32)[:, None], q)), MatZq(s[:, None], q)

    def _stacked(self, pk) -> MatZq:
        A, p = pk
        if not isinstance(A, MatZq) or not isinstance(p, MatZq):
            raise TypeError("public key must be (MatZq, MatZq)")
        if A.q != self.q or p.q != self.q:
            raise ValueError("mismatching modulus between scheme and key")
        if A.entries.shape != (self.n, self.m) or p.entries.shape != (self.m, 1):
            raise ValueError("mismatching matrix dimensions")
        K = getattr(A, "_stacked_with", None)
        if K is None or K[0] is not p:
            K = (p, MatZq(np.vstack([A.entries, p.entries.T]), self.q))
            A._stacked_with = K
        return K[1]

    def enc(self, pk, message, e=None):
        """regev_discrete_gauss.rs:413-430: returns (u, c)."""
        K = self._stacked(pk)
        if e is None:
            e = self.ctx.to_host_u32(self._sample_d(3, self.m))
        y = self.ctx.matvec(self._dev_key(K), np.asarray(e, np.uint32))
        c = (int(y[-1]) + message_bit(message) * (self.q // 2)) % self.q
        return MatZq(y[:-1, None], self.q), c

    def dec(self, sk: MatZq, cipher) -> int:
        """regev_discrete_gauss.rs:451-462: c - <s, u>."""
        u, c = cipher
        row = np.concatenate([u.entries.reshape(-1).astype(np.uint32), np.array([int(c) % self.q], np.uint32)])
        return int(self.dec_batch(sk, row[None, :])[0])

    def enc_batch(self, pk, messages, e=None):
        """B encryptions on kernel (a): rows (u_b ; c_b) of a (B, n+1) device tensor."""
        K = self._stacked(pk)
        ctx = self.ctx
        msg = np.array([message_bit(v) for v in np.asarray(messages).reshape(-1)], np.uint8)
        B = msg.size
        if e is None:
            e = self._sample_d(3, B * self.m).view(B, self.m)
        elif isinstance(e, np.ndarray):
            e = ctx.to_dev_u32(e)
        return ctx.matmul(self._dev_key(K), e, X_U32, B, msg=ctx.to_dev_u8(msg), x_digits=self._e_digits)

    def dec_batch(self, sk: MatZq, ciphers) -> np.ndarray:
        s = sk.entries.reshape(-1).astype(np.int64)
        if s.size != self.n:
            raise ValueError("mismatching secret key dimension")
        return self._decode_batch(((self.q - s) % self.q).astype(np.uint32), ciphers)


class DualRegevWithDiscreteGaussianRegularity(_GaussianRegularityBase):
    """pk = (A, u = A e), sk = e short; cipher = (p = A^t s + x, c = u^t s + x' + msg*floor(q/2))."""

    def gen(self):
        """dual_regev_discrete_gauss.rs:369-383."""
        n, m, q = self.n, self.m, self.q
        A = self.rng.uniform(1, n * m, q).reshape(n, m)
        e = self.ctx.to_host_u32(self._sample_d(2, m))
        key_a = self.ctx.key_upload(A, q)
        u = self.ctx.matvec(key_a, e)
        key_a.close()
        return (MatZq(A, q), MatZq(u[:, None], q)), MatZq(e[:, None], q)

    def _joined(self, pk) -> MatZq:
        A, u = pk
        if not isinstance(A, MatZq) or not isinstance(u, MatZq):
            raise TypeError("public key must be (MatZq, MatZq)")
        if A.q != self.q or u.q != self.q:
            raise ValueError("mismatching modulus between scheme and key")
        if A.entries.shape != (self.n, self.m) or u.entries.shape != (self.n, 1):
            raise ValueError("mismatching matrix dimensions")
        K = getattr(A, "_joined_with", None)
        if K is None or K[0] is not u:
            K = (u, MatZq(np.hstack([A.entries, u.entries]), self.q))
            A._joined_with = K
        return K[1]

    def enc(self, pk, message, s=None, x=None):
        """dual_regev_discrete_gauss.rs:404-431: returns (p, c)."""
        K = self._joined(pk)
        q = self.q
        if s is None:
            s = self.rng.uniform(3, self.n, q)
        if x is None:
            x = self.ctx.to_host_u32(self._chi(4, self.m + 1))
        y = (self.ctx.vecmat(self._dev_key(K), np.asarray(s, np.uint32)).astype(np.uint64) + np.asarray(x, np.uint64)) % q
        c = (int(y[-1]) + message_bit(message) * (q // 2)) % q
        return MatZq(y[:-1].astype(np.uint32)[:, None], q), c

    def dec(self, sk: MatZq, cipher) -> int:
        """dual_regev_discrete_gauss.rs:455-466: c - <e, p>."""
        p, c = cipher
        row = np.concatenate([p.entries.reshape(-1).astype(np.uint32), np.array([int(c) % self.q], np.uint32)])
        return int(self.dec_batch(sk, row[None, :])[0])

    def enc_batch(self, pk, messages, s=None, x=None):
        """Batched: rows (p_b ; c_b) of a (B, m+1) device tensor = S*[A | u] + X."""
        K = self._joined(pk)
        ctx, q = self.ctx, self.q
        msg = np.array([message_bit(v) for v in np.asarray(messages).reshape(-1)], np.uint8)
        B = msg.size
        if s is None:
            s = ctx.sample_uniform(self.rng.seed, 3, self.rng._take(3, B * self.n), B * self.n, q).view(B, self.n)
        elif isinstance(s, np.ndarray):
            s = ctx.to_dev_u32(s)
        if x is None:
            x = self._chi(4, B * (self.m + 1)).view(B, self.m + 1)
        elif isinstance(x, np.ndarray):
            x = ctx.to_dev_u32(x)
        return ctx.matmul(self._dev_key(K, transpose=True), s, X_U32, B, addend=x, msg=ctx.to_dev_u8(msg))

    def dec_batch(self, sk: MatZq, ciphers) -> np.ndarray:
        e = sk.entries.reshape(-1).astype(np.int64)
        if e.size != self.m:
            raise ValueError("mismatching secret key dimension")
        return self._decode_batch(((self.q - e) % self.q).astype(np.uint32), ciphers)
