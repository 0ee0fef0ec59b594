"""SIS hash -- mirror of construction/hash/sis.rs."""
from __future__ import annotations

import math

import numpy as np

from ...device import X_BITS, X_U32, bit_stride, pack_bits
from ...math import MatZq, MathError
from ...runtime import Randomness, default_context


class SISHash:
    def __init__(self, key: MatZq, ctx=None):
        self.key = key
        self._ctx = ctx

    @property
    def ctx(self):
        return self._ctx or default_context()

    @classmethod
    def gen(cls, n, m, q, ctx=None, seed=None) -> "SISHash":
        """sis.rs:79-93: uniform key A in Z_q^{n x m}; MathError if n <= 0."""
        n, m, q = int(n), int(m), int(q)
        if n < 1:
            raise MathError("n must be chosen bigger than 0.")
        rng = Randomness(seed, ctx)
        return cls(MatZq(rng.uniform(1, n * m, q).reshape(n, m), q), ctx)

    def check_security(self):
        """sis.rs:114-136."""
        n, m, q = self.key.get_num_rows(), self.key.get_num_columns(), self.key.q
        m_bar = math.ceil(n * math.log2(q))
        if m < m_bar:
            raise MathError("m was chosen smaller than n log q")
        if q <= math.ceil(math.sqrt(m_bar)):
            raise MathError("q was chosen smaller than ceil(sqrt(n log q))")

    def _dev_key(self):
        tag = (id(self.ctx), False)
        k = self.key._dev.get(tag)
        if k is None or k.h is None:
            k = self.ctx.key_upload(self.key.entries, self.key.q)
            self.key._dev[tag] = k
        return k

    def hash(self, value: MatZq) -> MatZq:
        """sis.rs:161-167: key * value; panics unless value is a column vector with matching
        modulus and length."""
        if not value.is_column_vector():
            raise ValueError("The hashed value has to be a column vector!")
        if value.q != self.key.q:
            raise ValueError("mismatching modulus")
        if value.get_num_rows() != self.key.get_num_columns():
            raise ValueError("mismatching matrix dimensions")
        d = self.ctx.matvec(self._dev_key(), value.entries.reshape(-1))
        return MatZq(d[:, None], self.key.q)

    def hash_batch(self, values, binary: bool = False):
        """Hash B inputs at once on kernel (a).  values: (B, m) residues (numpy / device int32), or
        with binary=True a packed bit matrix (B, bit_stride(m)).  Returns a device tensor (B, n)."""
        ctx, m = self.ctx, self.key.get_num_columns()
        if binary:
            if isinstance(values, np.ndarray):
                values = ctx.to_dev_u8(values if values.shape[1] == bit_stride(m) and values.dtype == np.uint8
                                       else pack_bits(values))
            return ctx.matmul(self._dev_key(), values, X_BITS, values.shape[0])
        if isinstance(values, np.ndarray):
            if values.ndim != 2 or values.shape[1] != m:
                raise ValueError("mismatching matrix dimensions")
            values = ctx.to_dev_u32(values)
        return ctx.matmul(self._dev_key(), values, X_U32, values.shape[0])
