"""Exact big-integer Python restatement of the qFALL-crypto hot path (small cases only).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import anything under oracle/.  This module exists to (1) pin the conventions against the
reference's own known-answer tests and (2) cross-check the C restatement (qf_oracle.c) with an
independent implementation that uses unbounded Python ints.

Each function cites the reference file:line it follows (paths relative to /root/reference/src).
"""
from __future__ import annotations

import hashlib
from fractions import Fraction

MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------------------- conventions
def ring_dist(a: int, x: int, q: int) -> int:
    """Zq::distance -- wrap-around distance (regev.rs:447)."""
    d = abs(a % q - x % q)
    return min(d, q - d)


def decide_bit(r: int, q: int) -> int:
    """regev.rs:445-451: 1 iff dist(r,0) > dist(r, floor(q/2)); ties -> 0."""
    return 1 if ring_dist(r, 0, q) > ring_dist(r, q // 2, q) else 0


# ----------------------------------------------------------------------------- generator
def sm64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK64
    return x ^ (x >> 31)


def rand64(seed: int, stream: int, idx: int) -> int:
    return sm64(((seed ^ (stream << 48)) + idx) & MASK64)


def rand_below(seed, stream, idx, bound):
    return (rand64(seed, stream, idx) * bound) >> 64


def chacha20_block(key32: bytes, counter: int, nonce: int):
    """ChaCha20 block, 20 rounds, 64-bit counter in state words 12-13 and 64-bit nonce in 14-15 (the keyed
    generator of include/qfgpu.h; same 16-word state as RFC 7539 2.3, whose test vector 2.3.2 pins it)."""
    M = 0xFFFFFFFF
    rot = lambda v, c: ((v << c) | (v >> (32 - c))) & M
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574]
    st += [int.from_bytes(key32[4 * i:4 * i + 4], "little") for i in range(8)]
    st += [counter & M, (counter >> 32) & M, nonce & M, (nonce >> 32) & M]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rot(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rot(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rot(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rot(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M for a, b in zip(x, st)]


def rand64_keyed(key32: bytes, nonce: int, stream: int, idx: int) -> int:
    o = chacha20_block(key32, idx, nonce ^ (stream << 48))
    return o[0] | (o[1] << 32)


def sample_uniform(seed, stream, first, count, q):
    return [rand_below(seed, stream, first + i, q) for i in range(count)]


def sample_noise(seed, stream, first, count, q, W):
    out = []
    for i in range(count):
        v = sum(rand_below(seed, stream, (first + i) * 4 + t, 2 * W + 1) - W for t in range(4))
        out.append(v % q)
    return out


def sample_bits(seed, stream, first_row, rows, cols):
    """bit j of row r = bit (j & 63) of value((first_row + r) * ceil(cols/64) + (j >> 6))"""
    w64 = (cols + 63) // 64
    return [[(rand64(seed, stream, (first_row + r) * w64 + (j >> 6)) >> (j & 63)) & 1 for j in range(cols)]
            for r in range(rows)]


# ----------------------------------------------------------------------------- Z_q matrices
def matvec(A, x, q):
    """regev.rs:404, lpr.rs:420, sis.rs:166: A*x mod q."""
    return [sum(a * b for a, b in zip(row, x)) % q for row in A]


def vecmat(x, A, q):
    """regev.rs:367: x^t*A mod q."""
    cols = len(A[0])
    return [sum(x[i] * A[i][j] for i in range(len(A))) % q for j in range(cols)]


def regev_enc(pk, x, msg, q):
    """regev.rs:396-414 with the randomness x supplied by the caller."""
    c = matvec(pk, x, q)
    c[-1] = (c[-1] + (msg % 2) * (q // 2)) % q
    return c


def lpr_enc(pk, r, e, msg, q):
    """lpr.rs:394-430 with r, e supplied."""
    c = [(ci + ei) % q for ci, ei in zip(matvec(pk, r, q), e)]
    c[-1] = (c[-1] + (msg % 2) * (q // 2)) % q
    return c


def regev_dec(s, c, q):
    """regev.rs:438-452 / lpr.rs:454-468: <[-s;1], c>."""
    r = (sum((-si) * ci for si, ci in zip(s, c[:-1])) + c[-1]) % q
    return decide_bit(r, q)


def dual_regev_enc(pk, s, e, msg, q):
    """dual_regev.rs:385-413: c^t = s^t*pk + e^t, message on the last entry."""
    c = [(ci + ei) % q for ci, ei in zip(vecmat(s, pk, q), e)]
    c[-1] = (c[-1] + (msg % 2) * (q // 2)) % q
    return c


def dual_regev_dec(x, c, q):
    """dual_regev.rs:437-450: c^t * [-x;1]."""
    r = (c[-1] - sum(xi * ci for xi, ci in zip(x, c[:-1]))) % q
    return decide_bit(r, q)


# ----------------------------------------------------------------------------- R_q
def rot_minus(a):
    """utils/rotation_matrix.rs:45-67 (signed integers, no modulus)."""
    n = len(a)
    out = [[0] * n for _ in range(n)]
    for i in range(n):
        for j in range(n):
            k = i + j
            if k >= n:
                out[k % n][j] = -a[i]
            else:
                out[k][j] = a[i]
    return out


def rot_minus_matrix(mat):
    """utils/rotation_matrix.rs:89-100: [rot(col_0) | rot(col_1) | ...] of a matrix."""
    rows, cols = len(mat), len(mat[0])
    blocks = [rot_minus([mat[r][c] for r in range(rows)]) for c in range(cols)]
    return [sum((blk[r] for blk in blocks), []) for r in range(rows)]


def polymul_negacyclic(a, b, q):
    """&PolynomialRingZq * &PolynomialRingZq (ring_lpr.rs:391): a*b mod (X^n+1, q)."""
    n = len(a)
    c = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k >= n:
                c[k - n] -= ai * bj
            else:
                c[k] += ai * bj
    return [v % q for v in c]


def ringlpr_enc(a, b, r, e1, e2, mu_bits, q):
    """ring_lpr.rs:422-461."""
    u = [(x + y) % q for x, y in zip(polymul_negacyclic(a, r, q), e1)]
    v = [(x + y + m * (q // 2)) % q
         for x, y, m in zip(polymul_negacyclic(b, r, q), e2, mu_bits)]
    return u, v


def ringlpr_dec(s, u, v, q):
    """ring_lpr.rs:489-508."""
    su = polymul_negacyclic(s, u, q)
    w = [(x - y) % q for x, y in zip(v, su)]
    return [1 if ring_dist(c, q // 2, q) < ring_dist(c, 0, q) else 0 for c in w]


def message_to_mu_bits(message: int, n: int):
    """ring_lpr.rs:424-427,318-329: |message| mod 2^n, bit i -> coefficient i."""
    mu = abs(message) % (1 << n)
    return [(mu >> i) & 1 for i in range(n)]


def fdh_ring_f_a(A, sigma, q):
    """psf/gpv_ring.rs:235-239: reduce sigma into R_q, then a*sigma (1 x mbar times mbar x 1)."""
    n = len(A[0])
    t = [0] * n
    for aj, sj in zip(A, sigma):
        p = polymul_negacyclic(aj, [c % q for c in sj], q)
        t = [(x + y) % q for x, y in zip(t, p)]
    return t


def fdh_check_domain(sigma, s: Fraction, mbar: int):
    """psf/gpv_ring.rs:266-275: ||iota(sigma)||^2 <= s^2 * (n*mbar), exact rationals."""
    n = len(sigma[0])
    norm = sum(c * c for sj in sigma for c in sj)
    return len(sigma) == mbar and Fraction(norm) <= s * s * (n * mbar)


def norm_bound(s: Fraction, n: int, mbar: int) -> int:
    """floor(s^2 * n * mbar): integer form of the bound above (norm is an integer)."""
    v = s * s * (n * mbar)
    return v.numerator // v.denominator


# ----------------------------------------------------------------------------- hashing
def sha256_hex(s: str) -> str:
    """hash/sha256.rs:37-42."""
    return hashlib.sha256(s.encode()).hexdigest()


def hash_to_zq(s: str, q: int) -> int:
    """hash/sha256.rs:65-83."""
    hexs = ""
    for i in range(q.bit_length() // 128 + 1):
        hexs += sha256_hex(f"{i} {q} {s}")
    return int(hexs, 16) % q


def hash_to_mat_zq(s: str, rows: int, cols: int, q: int):
    """hash/sha256.rs:110-133."""
    return [[hash_to_zq(f"{i} {j} {rows} {cols} {s}", q) for j in range(cols)]
            for i in range(rows)]


def hash_to_ring(s: str, n: int, q: int):
    """hash/sha256.rs:249-256 for rows=cols=1 (fdh/gpv_ring.rs:78-82): coefficient r of the
    single polynomial is entry (r,0) of the (n x 1) hash matrix."""
    return [row[0] for row in hash_to_mat_zq(s, n, 1, q)]


# ----------------------------------------------------------------------------- parameters
def log2_ceil(q: int) -> int:
    """Z::log_ceil(2) (gadget_parameters.rs:175)."""
    return (q - 1).bit_length()
