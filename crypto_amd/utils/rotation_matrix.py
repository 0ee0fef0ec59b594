"""utils/rotation_matrix.rs: negacyclic rotation matrices (host-side structure building only)."""
from __future__ import annotations

import numpy as np


def rot_minus(vec) -> np.ndarray:
    """rotation_matrix.rs:45-67: column j holds the coefficients of a * X^j mod X^n + 1.
    Accepts a column or a row vector; panics (ValueError) on a matrix."""
    a = np.asarray(vec, dtype=object)
    if a.ndim == 2:
        if a.shape[1] == 1:
            a = a[:, 0]
        elif a.shape[0] == 1:
            a = a[0]
        else:
            raise ValueError("The input must be a vector.")
    n = len(a)
    out = np.zeros((n, n), dtype=object)
    for i in range(n):
        for j in range(n):
            k = i + j
            if k >= n:
                out[k % n, j] = -a[i]
            else:
                out[k, j] = a[i]
    return out


def rot_minus_matrix(matrix) -> np.ndarray:
    """rotation_matrix.rs:89-100: [rot(col_0) | rot(col_1) | ...]."""
    m = np.asarray(matrix, dtype=object)
    return np.concatenate([rot_minus(m[:, c]) for c in range(m.shape[1])], axis=1)
