"""SHA-256 based hashes -- mirror of construction/hash/sha256.rs (host side, hashlib).
These feed FDH verification (the string-hash step); the lattice arithmetic that
consumes their output runs on the GPU."""
from __future__ import annotations

import hashlib

import numpy as np

from ...math import MatZq


def sha256(string: str) -> str:
    """sha256.rs:37-42."""
    return hashlib.sha256(string.encode()).hexdigest()


def hash_to_zq_sha256(string: str, modulus: int) -> int:
    """sha256.rs:65-83."""
    modulus = int(modulus)
    if modulus <= 1:
        raise ValueError("modulus must be > 1")
    hexs = "".join(sha256(f"{i} {modulus} {string}") for i in range(modulus.bit_length() // 128 + 1))
    return int(hexs, 16) % modulus


def hash_to_mat_zq_sha256(string: str, num_rows: int, num_cols: int, modulus: int) -> MatZq:
    """sha256.rs:110-133."""
    if num_rows <= 0 or num_cols <= 0:
        raise ValueError("number of rows and columns must be positive")
    new_string = f"{num_rows} {num_cols} {string}"
    ent = np.array([[hash_to_zq_sha256(f"{i} {j} {new_string}", modulus) for j in range(num_cols)]
                    for i in range(num_rows)], dtype=object)
    return MatZq(ent, modulus)


class HashMatZq:
    """sha256.rs:160-194."""

    def __init__(self, modulus: int, rows: int, cols: int):
        self.modulus, self.rows, self.cols = int(modulus), int(rows), int(cols)

    def hash(self, m: str) -> MatZq:
        return hash_to_mat_zq_sha256(m, self.rows, self.cols, self.modulus)


class HashMatPolynomialRingZq:
    """sha256.rs:220-257: hash into a rows x cols matrix over R_q.  On the hot path rows = cols = 1
    (fdh/gpv_ring.rs:78-82), where coefficient r of the polynomial is entry (r, 0) of the
    (n x 1) hash matrix."""

    def __init__(self, modulus, rows: int = 1, cols: int = 1):
        self.modulus, self.rows, self.cols = modulus, int(rows), int(cols)

    def hash(self, m: str) -> np.ndarray:
        if (self.rows, self.cols) != (1, 1):
            raise NotImplementedError("only the 1x1 target used by FDH::init_gpv_ring is on the hot path")
        n, q = self.modulus.get_degree(), self.modulus.get_q()
        return hash_to_mat_zq_sha256(m, n, 1, q).entries.reshape(-1)

    def hash_batch_dev(self, ctx, messages):
        """The same hash for many strings on the device (qf_hash_to_mat_zq_sha256_dev): (B, n) residues
        resident in HBM, ready to be the `u` operand of the verification kernel."""
        if (self.rows, self.cols) != (1, 1):
            raise NotImplementedError("only the 1x1 target used by FDH::init_gpv_ring is on the hot path")
        return ctx.hash_to_mat_zq_sha256(messages, self.modulus.get_degree(), 1, self.modulus.get_q())
