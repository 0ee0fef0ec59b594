"""Host-side value types at the drop-in boundary.

The reference exchanges qfall-math objects (MatZq, MatZ, PolynomialRingZq, MatPolyOverZ ...);
here the same roles are played by small numpy-backed classes with the same text formats
("[[a, b],[c, d]] mod q", "len  c0 c1 ...").  They carry data only -- all arithmetic on them is
done by the GPU library through `crypto_amd.device`.
"""
from __future__ import annotations

import math
import re
from fractions import Fraction

import numpy as np


class MathError(ValueError):
    """Counterpart of qfall_math::error::MathError (returned by the reference's check_* methods)."""


def is_prime(n: int) -> bool:
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def log2_ceil(q: int) -> int:
    """Z::log_ceil(2)."""
    return (q - 1).bit_length()


def as_fraction(x) -> Fraction:
    if isinstance(x, Fraction):
        return x
    if isinstance(x, float):
        return Fraction(x)  # exact binary value, as Q::from(f64) does
    return Fraction(x)


class MatZq:
    """Matrix over Z_q: canonical residues as uint32, row-major."""

    def __init__(self, entries, q: int):
        if q < 2:
            raise ValueError("modulus must be > 1")
        a = np.asarray(entries)
        if a.ndim != 2:
            raise ValueError("MatZq needs a 2-D array")
        self.q = int(q)
        if a.dtype == object:
            arr = np.array([[int(v) % self.q for v in row] for row in a], dtype=np.uint32).reshape(a.shape)
        elif a.dtype.kind == "i":
            arr = np.mod(a.astype(np.int64), self.q).astype(np.uint32)
        else:
            arr = (a.astype(np.uint64) % self.q).astype(np.uint32)
        self.entries = np.ascontiguousarray(arr)
        self._dev = {}

    @classmethod
    def from_canonical(cls, entries: np.ndarray, q: int) -> "MatZq":
        """Adopts a 2-D uint32 array whose entries are already canonical residues (what the device
        library returns): no pass over the data."""
        self = cls.__new__(cls)
        assert entries.dtype == np.uint32 and entries.ndim == 2
        self.q = int(q)
        self.entries = np.ascontiguousarray(entries)
        self._dev = {}
        return self

    # -- qfall-math style accessors
    def get_num_rows(self): return self.entries.shape[0]
    def get_num_columns(self): return self.entries.shape[1]
    def get_mod(self): return self.q
    def is_column_vector(self): return self.entries.shape[1] == 1
    def is_row_vector(self): return self.entries.shape[0] == 1
    def get_entry(self, i, j): return int(self.entries[i, j])
    def transpose(self): return MatZq(self.entries.T, self.q)

    def __eq__(self, other):
        return isinstance(other, MatZq) and self.q == other.q and np.array_equal(self.entries, other.entries)

    def __repr__(self):
        rows = ",".join("[" + ", ".join(str(int(v)) for v in r) + "]" for r in self.entries)
        return f"[{rows}] mod {self.q}"

    @staticmethod
    def from_str(s: str) -> "MatZq":
        body, q = s.rsplit("mod", 1)
        rows = re.findall(r"\[([^\[\]]*)\]", body)
        return MatZq(np.array([[int(v) for v in r.split(",")] for r in rows], dtype=object), int(q))


class MatZ:
    """Integer matrix (small signed entries on this path), int64."""

    def __init__(self, entries):
        a = np.asarray(entries, dtype=np.int64)
        if a.ndim != 2:
            raise ValueError("MatZ needs a 2-D array")
        self.entries = np.ascontiguousarray(a)

    def get_num_rows(self): return self.entries.shape[0]
    def get_num_columns(self): return self.entries.shape[1]
    def is_column_vector(self): return self.entries.shape[1] == 1

    def __eq__(self, other):
        return isinstance(other, MatZ) and np.array_equal(self.entries, other.entries)

    @staticmethod
    def from_str(s: str) -> "MatZ":
        rows = re.findall(r"\[([^\[\]]*)\]", s)
        return MatZ([[int(v) for v in r.split(",")] for r in rows])


def parse_poly(s: str):
    """'len  c0 c1 ...' (two spaces) -> list of coefficients; '0' -> []."""
    parts = s.split()
    if not parts or parts[0] == "0":
        return []
    k = int(parts[0])
    coeffs = [int(v) for v in parts[1:]]
    assert len(coeffs) == k
    return coeffs


def format_poly(coeffs) -> str:
    c = [int(v) for v in coeffs]
    while c and c[-1] == 0:
        c.pop()
    return "0" if not c else f"{len(c)}  " + " ".join(str(v) for v in c)


class PolynomialRingZq:
    """Element of Z_q[X]/(X^n+1): n canonical coefficients, degree-ascending."""

    def __init__(self, coeffs, n: int, q: int):
        c = np.zeros(n, dtype=object)
        cc = list(coeffs)
        if len(cc) > n:
            raise ValueError("more coefficients than the ring degree; reduce first")
        for i, v in enumerate(cc):
            c[i] = int(v) % q
        self.n, self.q = int(n), int(q)
        self.coeffs = c.astype(np.uint32)

    def get_coeff(self, i): return int(self.coeffs[i])

    def __eq__(self, other):
        return (isinstance(other, PolynomialRingZq) and (self.n, self.q) == (other.n, other.q)
                and np.array_equal(self.coeffs, other.coeffs))

    def __repr__(self):
        return f"{format_poly(self.coeffs)} / X^{self.n}+1 mod {self.q}"


class MatPolyOverZ:
    """Column vector of integer polynomials (FDH signatures): int64 array (rows, n)."""

    def __init__(self, polys, n: int):
        a = np.zeros((len(polys), n), dtype=np.int64)
        for i, p in enumerate(polys):
            p = list(p)
            a[i, :len(p)] = p
        self.entries, self.n = a, int(n)

    def get_num_rows(self): return self.entries.shape[0]
    def is_column_vector(self): return True

    def __eq__(self, other):
        return isinstance(other, MatPolyOverZ) and np.array_equal(self.entries, other.entries)

    @staticmethod
    def from_str(s: str, n: int) -> "MatPolyOverZ":
        inner = s.strip()
        inner = inner[inner.index("[[") + 2: inner.rindex("]]")]
        # a row "[[p0, p1, ...]]" or a column "[[p0],[p1],...]" (the reference's signatures are columns)
        return MatPolyOverZ([parse_poly(p.strip(" []")) for p in inner.split(",")], n)


def gaussian_width(s: Fraction | float) -> int:
    """Half-width W of the seeded Gaussian stand-in for Gaussian parameter s:
    sigma = s/sqrt(2*pi), value = sum of 4 uniform integers in [-W,W], W = round(sigma*sqrt(3)/2)."""
    sigma = float(s) / math.sqrt(2 * math.pi)
    return int(round(sigma * math.sqrt(3) / 2))
