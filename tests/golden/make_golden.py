"""Regenerates tests/golden/*.json.

Two kinds of fixtures:
  reference_kats.json -- DATA transcribed from the reference's own tests (inputs + expected
      outputs only): rot_minus (utils/rotation_matrix.rs:110-138), the fixed ring trapdoor of
      sample/g_trapdoor/short_basis_ring.rs:357-378 (a = [1 | a_bar | g^t - (a_bar*r + e)],
      gadget_ring.rs:71-77), SHA-256 KATs (construction/hash/sha256.rs:35,60,104,270-332),
      default / secure128 parameter sets.  The reference cannot be executed here (Rust, no
      toolchain), so these are copied values, not regenerated ones.
  oracle_vectors.json -- small seeded input/output vectors produced by the exact-integer Python
      oracle (oracle/pyoracle.py).  They pin the C oracle and the GPU path to each other across
      rounds; they are NOT reference outputs (scheme-level products have no KAT upstream).
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402

SEED = 0x5146414C4C

reference_kats = {
    "rot_minus": {"vec": [1, 5, -1, 9],
                  "expected": [[1, -9, 1, -5], [5, 1, -9, 1], [-1, 5, 1, -9], [9, -1, 5, 1]]},
    "rot_minus_matrix": {"mat": [[1, 5, -1, 9], [18446744073709551615, 1, 2, 3]],
                         "expected": [[1, -18446744073709551615, 5, -1, -1, -2, 9, -3],
                                      [18446744073709551615, 1, 1, 5, 2, -1, 3, 9]]},
    "ring_trapdoor_n4_q16": {
        "n": 4, "q": 16,
        "a": [[1], [2, 8, 8, 12], [11, 10, 7, 13], [9, 6, 6, 12], [6, 11, 1, 6], [3, 10, 2, 9]],
        "r": [[-1, 7, 6, -8], [0, -2, 4], [0, 3, -4, 1], [6, 4, -1, 3]],
        "e": [[-4, 8, -3, 7], [1, -2, 2, 4], [-6, 7, -5], [-7, 10, -12, -15]]},
    # sample/g_trapdoor/short_basis_classical.rs:281-303 (fixed trapdoor, n=2, q=8) with the is_trapdoor
    # identity of gadget_classical.rs:300-322 / g_trapdoor.rs:168-184:  A * [R ; I] = G = I_2 (x) (1,2,4)  mod 8
    "classical_trapdoor_n2_q8": {
        "q": 8,
        "A": [[2, 6, 2, 5, 3, 0, 1, 1, 1, 6, 5, 0, 6], [6, 0, 3, 1, 5, 6, 2, 7, 0, 3, 7, 7, 0]],
        "R": [[0, 1, 0, 1, 1, 0], [-1, 1, 0, 0, 0, -1], [-1, 0, -1, -1, -1, 0], [-1, 1, 0, 0, 0, 1],
              [-1, -1, 0, 1, 0, 1], [-1, 0, 0, -1, 0, 1], [0, -1, 0, 0, 0, 0]],
        "G": [[1, 2, 4, 0, 0, 0], [0, 0, 0, 1, 2, 4]],
        # working_sa_r_identity, short_basis_classical.rs:339-356: gen_sa_r(params, tag = I, A) = [0 | I ; S' | W]
        # with G*W = -A*[I_7 | 0] mod 8 (compute_w :114-119) -- the reference's literal 13 x 13 output
        "sa_r_identity": [
            [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0],
            [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0],
            [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0, 2, 0, 0, 0, 1, 1, 0, 1], [0, 0, 0, 0, 2, -1, 1, 1, 1, 1, 0, 0, 1],
            [0, 0, 0, 2, -1, 0, 1, 0, 1, 0, 1, 0, 1], [0, 0, 2, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0], [0, 2, -1, 0, 0, 0, 1, 0, 0, 1, 1, 1, 1],
            [2, -1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1]]},
    # working_sa_r, short_basis_ring.rs:404-439: gen_sa_r for the fixed ring trapdoor above (n = 4, q = 16, k = 4),
    # reduced by X^4 + 1 and coefficient-embedded (24 x 24; row 4*i + c = coefficient c of polynomial row i).  Columns
    # 16.. hold pd (x) [I_2 ; w]: w = base-2 digits of (-1)*a_0 and (-1)*a_1 in R_16 (compute_w :128-139), then
    # every X^s * w reduced by X^4 + 1; columns 0..15 hold pd (x) S'' (monomials 2*X^s and -X^s)
    "ring_sa_r_n4_q16": {"n": 4, "q": 16, "k": 4, "matrix": [
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1],
        [0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 1, 0],
        [0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 1, 1],
        [0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, -1, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, -1, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, -1],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, -1, 0, 0, 1, 0, 0, 0, 0, 1, 1],
        [2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, -1, 0, -1],
        [0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, -1],
        [0, 0, 0, 0, 0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 1, 1, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, -1, 0, 0, 0, 0, 0, 1, 0, 1, 1, 1]]},
    "sha256": {"Hello World!": "7f83b1657ff1fc53b92dc18148a1d65dfc2d4b1fa3d677284addd200126d9069",
               "qfall": "eb6ed1369a670050bd04b24036e8c29144b0f6b10166dc9c8b4987a6026c715f"},
    "hash_to_zq": [["Hello World!", 7, 2], ["Hello World!", 256, 150], ["qfall", 16, 12]],
    "hash_to_mat_zq": [["Hello World!", 2, 2, 7, [[6, 3], [5, 2]]],
                       ["Hello World!", 2, 2, 256, [[159, 26], [249, 141]]],
                       ["qfall", 2, 2, 16, [[3, 12], [9, 12]]]],
    "parameter_sets": {
        "regev_default": [13, 154, 1427, 0.01], "regev_secure128": [230, 5313, 7764299, 0.0011],
        "lpr_default": [10, 983, 0.0072], "lpr_secure128": [500, 76859609, 0.000005],
        "ring_lpr_default": [16, 2399, 0.0039], "ring_lpr_secure128": [512, 92897729, 0.000005]},
}


def regev_case(n, m, q, W, B):
    pk = [po.sample_uniform(SEED, 1, i * m, m, q) for i in range(n + 1)]
    x = po.sample_bits(SEED, 3, 0, B, m)
    msg = [v for v in po.sample_uniform(SEED, 5, 0, B, 2)]
    c = [po.regev_enc(pk, x[b], msg[b], q) for b in range(B)]
    s = po.sample_uniform(SEED, 2, 0, n, q)
    return {"n": n, "m": m, "q": q, "B": B, "msg": msg, "cipher": c,
            "dec_with_stream2_secret": [po.regev_dec(s, cb, q) for cb in c]}


def lpr_case(n, q, W, B):
    pk = [po.sample_uniform(SEED, 1, i * n, n, q) for i in range(n + 1)]
    r = [po.sample_noise(SEED, 3, b * n, n, q, W) for b in range(B)]
    e = [po.sample_noise(SEED, 4, b * (n + 1), n + 1, q, W) for b in range(B)]
    msg = po.sample_uniform(SEED, 5, 0, B, 2)
    return {"n": n, "q": q, "W": W, "B": B, "msg": msg,
            "cipher": [po.lpr_enc(pk, r[b], e[b], msg[b], q) for b in range(B)]}


def ring_case(n, q, W, B):
    a = po.sample_uniform(SEED, 1, 0, n, q)
    s = po.sample_noise(SEED, 2, 0, n, q, W)
    e = po.sample_noise(SEED, 4, 0, n, q, W)
    b = [(x + y) % q for x, y in zip(po.polymul_negacyclic(a, s, q), e)]
    out = {"n": n, "q": q, "W": W, "B": B, "b": b, "u": [], "v": [], "dec": []}
    for i in range(B):
        r = po.sample_noise(SEED, 3, i * n, n, q, W)
        e1 = po.sample_noise(SEED, 4, n + i * n, n, q, W)
        e2 = po.sample_noise(SEED, 4, n + B * n + i * n, n, q, W)
        mu = po.sample_bits(SEED, 5, i, 1, n)[0]
        u, v = po.ringlpr_enc(a, b, r, e1, e2, mu, q)
        out["u"].append(u); out["v"].append(v); out["dec"].append(po.ringlpr_dec(s, u, v, q))
    return out


def polymul_case(n, q, B):
    a = [po.sample_uniform(SEED, 6, i * n, n, q) for i in range(B)]
    b = [po.sample_uniform(SEED, 6, (B + i) * n, n, q) for i in range(B)]
    return {"n": n, "q": q, "B": B, "c": [po.polymul_negacyclic(a[i], b[i], q) for i in range(B)]}


oracle_vectors = {
    "regev": [regev_case(13, 154, 1427, 5, 4), regev_case(4, 30, 997, 2, 3)],
    "lpr": [lpr_case(10, 983, 3, 4)],
    "ring_lpr": [ring_case(16, 2399, 4, 3), ring_case(8, 257, 1, 2)],
    "polymul": [polymul_case(16, 2399, 3), polymul_case(4, 16, 2), polymul_case(64, 257, 2), polymul_case(32, 1073479681, 2)],
}

with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
    json.dump(reference_kats, f, indent=1)
with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
    json.dump(oracle_vectors, f, separators=(",", ":"))
print("wrote golden fixtures")
