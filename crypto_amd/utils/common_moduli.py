"""utils/common_moduli.rs: the moduli X^n + 1 and X^n - 1 over Z_q."""
from __future__ import annotations

import numpy as np


class ModulusPolynomialRingZq:
    """Coefficients of the modulus polynomial (degree n, n+1 coefficients) mod q."""

    def __init__(self, coeffs, q: int):
        self.coeffs = np.asarray(coeffs, dtype=np.int64) % q
        self.q = int(q)

    def get_degree(self): return len(self.coeffs) - 1
    def get_q(self): return self.q

    def is_anticyclic(self):
        c = self.coeffs
        return c[0] == 1 % self.q and c[-1] == 1 % self.q and not c[1:-1].any()


def new_anticyclic(n: int, q: int) -> ModulusPolynomialRingZq:
    """common_moduli.rs:41-48: X^n + 1 mod q (n >= 1, q > 1)."""
    if n < 1:
        raise ValueError("n must be at least 1")
    if q < 2:
        raise ValueError("modulus must be > 1")
    c = np.zeros(n + 1, np.int64)
    c[0] = 1
    c[n] = 1
    return ModulusPolynomialRingZq(c, q)


def new_cyclic(n: int, q: int) -> ModulusPolynomialRingZq:
    """common_moduli.rs:72-79: X^n - 1 mod q."""
    if n < 1:
        raise ValueError("n must be at least 1")
    if q < 2:
        raise ValueError("modulus must be > 1")
    c = np.zeros(n + 1, np.int64)
    c[0] = -1
    c[n] = 1
    return ModulusPolynomialRingZq(c, q)
