"""Classical GPV preimage-sampleable function -- verify half of primitive/psf/gpv.rs.

`f_a` (gpv.rs:189-192: A*sigma) runs on kernel (c) for one signature and on kernel (a) for a batch;
`check_domain` (gpv.rs:218-223) is an exact integer norm test.  `trap_gen`, `samp_d`, `samp_p` are
sequential Gaussian sampling over Q: out of scope, NotImplementedError."""
from __future__ import annotations

import numpy as np

from ...device import X_I32
from ...math import MatZ, MatZq, as_fraction, log2_ceil
from ...runtime import default_context


class GadgetParameters:
    """sample/g_trapdoor/gadget_parameters.rs:86-108 (init_default): base 2, k = ceil(log2 q),
    m_bar = n*k + ceil(log2 n)^2; A is n x (m_bar + n*k)."""

    def __init__(self, n: int, q: int):
        self.n, self.q = int(n), int(q)
        if self.n < 1:
            raise ValueError("n must be at least 1")
        self.base = 2
        self.k = log2_ceil(self.q)
        self.m_bar = self.n * self.k + log2_ceil(self.n) ** 2

    init_default = classmethod(lambda cls, n, q: cls(n, q))


class PSFGPV:
    def __init__(self, gp: GadgetParameters, s, ctx=None):
        self.gp, self.s = gp, as_fraction(s)
        self._ctx = ctx

    @property
    def ctx(self):
        return self._ctx or default_context()

    @property
    def m(self) -> int:
        return self.gp.n * self.gp.k + self.gp.m_bar        # gpv.rs:219

    def trap_gen(self):
        raise NotImplementedError("trapdoor generation is key-generation-time rational arithmetic (out of scope)")

    def samp_d(self):
        raise NotImplementedError("SampleD is sequential Gaussian sampling (out of scope)")

    def samp_p(self, a, td, u):
        raise NotImplementedError("preimage sampling is sequential Gaussian sampling over Q (out of scope)")

    def check_domain(self, sigma) -> bool:
        """gpv.rs:218-223: column vector of m entries with ||sigma||^2 <= s^2 * m (exact)."""
        if not isinstance(sigma, MatZ) or not sigma.is_column_vector() or sigma.get_num_rows() != self.m:
            return False
        norm = sum(int(v) * int(v) for v in sigma.entries[:, 0])
        return norm <= self.s * self.s * self.m

    def _key(self, a: MatZq):
        if not isinstance(a, MatZq) or a.q != self.gp.q or a.entries.shape != (self.gp.n, self.m):
            raise ValueError("mismatching parity-check matrix")
        tag = ("psf", id(self.ctx))
        key = a._dev.get(tag)
        if key is None or key.h is None:
            key = a._dev[tag] = self.ctx.key_upload(a.entries, a.q)
        return key

    def f_a(self, a: MatZq, sigma: MatZ) -> MatZq:
        """gpv.rs:189-192: asserts the domain, returns A*sigma."""
        assert self.check_domain(sigma), "sigma is not in the domain"
        x = (sigma.entries[:, 0].astype(object) % self.gp.q).astype(np.uint32)
        return MatZq(self.ctx.matvec(self._key(a), x)[:, None], self.gp.q)

    def f_a_batch_dev(self, a: MatZq, sigmas):
        """A*sigma_b for B in-domain signatures at once (kernel a, signed operand): (B, n) device tensor."""
        arr = np.stack([s.entries[:, 0] for s in sigmas]).astype(np.int64)
        if np.abs(arr).max(initial=0) >= 2 ** 31:
            raise ValueError("signature entries exceed 32 bits")
        return self.ctx.matmul(self._key(a), self.ctx.to_dev_i32(arr.astype(np.int32)), X_I32, arr.shape[0])
