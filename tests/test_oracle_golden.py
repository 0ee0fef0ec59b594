"""CPU: pins the oracle (C restatement + exact Python) to the reference's own known-answer tests
and to the committed seeded vectors (tests/golden/, see make_golden.py for provenance)."""
import json
import os

import numpy as np
import pytest

from oracle import coracle as co
from oracle import pyoracle as po

G = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(G, "reference_kats.json")))
VEC = json.load(open(os.path.join(G, "oracle_vectors.json")))
SEED = 0x5146414C4C


def test_rot_minus_kat():
    """utils/rotation_matrix.rs:110-121."""
    k = KATS["rot_minus"]
    assert po.rot_minus(k["vec"]) == k["expected"]
    assert co.rot_minus_i64(k["vec"]).tolist() == k["expected"]


def test_rot_minus_matrix_kat():
    """utils/rotation_matrix.rs:125-138 (entries up to u64::MAX: exact Python ints)."""
    k = KATS["rot_minus_matrix"]
    assert po.rot_minus_matrix(k["mat"]) == k["expected"]


def test_negacyclic_product_equals_rot_minus_times_vector():
    """Rot^-(a)*b == a*b mod (X^n+1): ties the product convention to the rot_minus KAT."""
    rng = np.random.default_rng(1)
    for n, q in [(4, 16), (8, 257), (16, 2399)]:
        a = [int(v) for v in rng.integers(0, q, n)]
        b = [int(v) for v in rng.integers(0, q, n)]
        rot = po.rot_minus(a)
        via_rot = [sum(rot[i][j] * b[j] for j in range(n)) % q for i in range(n)]
        assert via_rot == po.polymul_negacyclic(a, b, q)
        assert via_rot == co.polymul_batch(np.array([a]), np.array([b]), q)[0].tolist()
        assert np.array_equal(co.rot_minus_modq(a, q), np.array(rot, dtype=object) % q)


def test_fixed_ring_trapdoor_kat():
    """short_basis_ring.rs:357-378 with gadget_ring.rs:71-77: a_{2+i} = 2^i - (a_bar*r_i + e_i)
    in Z_16[X]/(X^4+1) -- a known answer for the negacyclic product from the reference's tests."""
    k = KATS["ring_trapdoor_n4_q16"]
    n, q = k["n"], k["q"]
    pad = lambda p: list(p) + [0] * (n - len(p))
    a = [pad(p) for p in k["a"]]
    assert a[0] == [1, 0, 0, 0]
    for i in range(4):
        r = [c % q for c in pad(k["r"][i])]
        e = pad(k["e"][i])
        ar_py = po.polymul_negacyclic(a[1], r, q)
        ar_c = co.polymul_batch(np.array([a[1]]), np.array([r]), q)[0].tolist()
        assert ar_py == ar_c
        g = [1 << i, 0, 0, 0]
        assert [(g[j] - ar_py[j] - e[j]) % q for j in range(n)] == a[2 + i]
    # and as an FDH-style f_a: a * [e ; r ; I] = g^t  (gadget_ring.rs:189-210 is_trapdoor)
    for i in range(4):
        col = [pad(k["e"][i]), pad(k["r"][i])] + [[1 if (j == i and t == 0) else 0 for t in range(n)] for j in range(4)]
        t = po.fdh_ring_f_a(a, col, q)
        assert t == [(1 << i) % q, 0, 0, 0]
        tc, _ = co.fdh_ring_verify_batch(np.array(a), q, np.array([col], dtype=np.int32), np.zeros((1, n), np.uint32), 10 ** 6)
        assert tc[0].tolist() == t


def test_sha256_kats():
    """construction/hash/sha256.rs:35,60,104,270-332."""
    for s, h in KATS["sha256"].items():
        assert po.sha256_hex(s) == h and co.sha256(s.encode()).hex() == h
    for s, q, v in KATS["hash_to_zq"]:
        assert po.hash_to_zq(s, q) == v and co.hash_to_zq(s, q) == v
    for s, r, c, q, m in KATS["hash_to_mat_zq"]:
        assert po.hash_to_mat_zq(s, r, c, q) == m and co.hash_to_mat_zq(s, r, c, q).tolist() == m


def test_sha256_padding_lengths():
    import hashlib
    for ln in [0, 1, 55, 56, 63, 64, 65, 119, 120, 200]:
        d = bytes((i * 7 + 3) & 255 for i in range(ln))
        assert co.sha256(d) == hashlib.sha256(d).digest()


def test_generator_c_vs_python():
    q = 1073741789
    assert co.sample_uniform(SEED, 1, 3, 50, q).tolist() == po.sample_uniform(SEED, 1, 3, 50, q)
    assert co.sample_noise(SEED, 4, 9, 40, q, 300).tolist() == po.sample_noise(SEED, 4, 9, 40, q, 300)
    bits = co.sample_bits(SEED, 3, 2, 3, 77, 16)
    ref = po.sample_bits(SEED, 3, 2, 3, 77)
    assert np.unpackbits(bits, axis=1, bitorder="little")[:, :77].tolist() == ref
    assert not np.unpackbits(bits, axis=1, bitorder="little")[:, 77:].any()


def test_keyed_generator_chacha20():
    """The keyed generator (qf_sample_*_keyed_dev: what an object constructed without a seed draws from) is
    ChaCha20 in counter mode.  Pinned by the block test vector of RFC 7539 section 2.3.2 (key 00..1f, counter 1,
    nonce 00:00:00:09:00:00:00:4a:00:00:00:00 = state words 12..15 = 1, 0x09000000, 0x4a000000, 0), in the C
    oracle and in the independent Python restatement; then the keyed samplers C vs Python."""
    key = bytes(range(32))
    want = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
            0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    ctr, nonce = 1 | (0x09000000 << 32), 0x4a000000
    assert co.chacha20_block(key, ctr, nonce).tolist() == want
    assert po.chacha20_block(key, ctr, nonce) == want
    k2 = bytes((7 * i + 3) & 255 for i in range(32))
    for stream, idx in ((1, 0), (2, 5), (4, (1 << 40) + 9)):
        assert co.rand64_keyed(k2, 0x1234, stream, idx) == po.rand64_keyed(k2, 0x1234, stream, idx)
    q = 1073741789
    v = [po.rand64_keyed(k2, 9, 1, 3 + i) for i in range(50)]
    assert co.sample_uniform_keyed(k2, 9, 1, 3, 50, q).tolist() == [(x * q) >> 64 for x in v]
    W = 300
    nz = co.sample_noise_keyed(k2, 9, 4, 2, 10, q, W).tolist()
    for i in range(10):
        t = sum(((po.rand64_keyed(k2, 9, 4, (2 + i) * 4 + j) * (2 * W + 1)) >> 64) - W for j in range(4))
        assert nz[i] == t % q
    bits = np.unpackbits(co.sample_bits_keyed(k2, 9, 3, 2, 3, 77, 16), axis=1, bitorder="little")
    for r in range(3):
        for j in range(77):
            assert bits[r, j] == (po.rand64_keyed(k2, 9, 3, (2 + r) * 2 + (j >> 6)) >> (j & 63)) & 1
    assert not bits[:, 77:].any()
    # the keyed and the seeded Gaussian share the inversion table: equal draws give equal samples
    g = co.sample_discrete_gauss_keyed(k2, 9, 4, 0, 2000, 983, 10, 7.0).astype(np.int64)
    g[g > 491] -= 983
    assert np.abs(g).max() <= 24 and abs(g.mean()) < 0.5 and abs(g.var() / (49 / (2 * np.pi)) - 1) < 0.15


@pytest.mark.parametrize("case", VEC["regev"])
def test_regev_vectors(case):
    n, m, q, B = case["n"], case["m"], case["q"], case["B"]
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m)
    xb = co.sample_bits(SEED, 3, 0, B, m, ((m + 127) // 128) * 16)
    msg = np.array(case["msg"], np.uint8)
    c = co.matmul_batch(pk, q, xb, co.X_BITS, B, msg=msg)
    assert c.tolist() == case["cipher"]
    s = co.sample_uniform(SEED, 2, 0, n, q)
    assert co.dot_decode_batch(s, q, c).tolist() == case["dec_with_stream2_secret"]


@pytest.mark.parametrize("case", VEC["lpr"])
def test_lpr_vectors(case):
    n, q, W, B = case["n"], case["q"], case["W"], case["B"]
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * n, q).reshape(n + 1, n)
    r = co.sample_noise(SEED, 3, 0, B * n, q, W).reshape(B, n)
    e = co.sample_noise(SEED, 4, 0, B * (n + 1), q, W).reshape(B, n + 1)
    c = co.matmul_batch(pk, q, r, co.X_U32, B, addend=e, msg=np.array(case["msg"], np.uint8))
    assert c.tolist() == case["cipher"]


@pytest.mark.parametrize("case", VEC["ring_lpr"])
def test_ring_lpr_vectors(case):
    n, q, W, B = case["n"], case["q"], case["W"], case["B"]
    a = co.sample_uniform(SEED, 1, 0, n, q)
    s = co.sample_noise(SEED, 2, 0, n, q, W)
    b = np.array(case["b"], np.uint32)
    r = co.sample_noise(SEED, 3, 0, B * n, q, W).reshape(B, n)
    e1 = co.sample_noise(SEED, 4, n, B * n, q, W).reshape(B, n)
    e2 = co.sample_noise(SEED, 4, n + B * n, B * n, q, W).reshape(B, n)
    mu = co.sample_bits(SEED, 5, 0, B, n, (n + 7) // 8)
    u, v = co.ringlpr_enc_batch(a, b, q, r, e1, e2, mu)
    assert u.tolist() == case["u"] and v.tolist() == case["v"]
    dec = np.unpackbits(co.ringlpr_dec_batch(s, q, u, v), axis=1, bitorder="little")[:, :n]
    assert dec.tolist() == case["dec"]


@pytest.mark.parametrize("case", VEC["polymul"])
def test_polymul_vectors(case):
    n, q, B = case["n"], case["q"], case["B"]
    a = co.sample_uniform(SEED, 6, 0, B * n, q).reshape(B, n)
    b = co.sample_uniform(SEED, 6, B * n, B * n, q).reshape(B, n)
    assert co.polymul_batch(a, b, q).tolist() == case["c"]
    if co.find_psi(n, q):
        assert co.polymul_batch(a, b, q, use_ntt=True).tolist() == case["c"]


def test_cpu_ntt_equals_schoolbook():
    n, q = 1024, 1073479681
    a = co.sample_uniform(SEED, 6, 0, 2 * n, q).reshape(2, n)
    b = co.sample_uniform(SEED, 6, 5000, 2 * n, q).reshape(2, n)
    assert np.array_equal(co.polymul_batch(a, b, q, use_ntt=True), co.polymul_batch(a, b, q))
    with pytest.raises(ValueError):
        co.polymul_batch(a, b, 1073741789, use_ntt=True)


def test_decision_rule_and_round_trip_semantics():
    """regev.rs:445-451 ties -> 0; message mod 2 incl. negatives (regev.rs:598-612)."""
    q = 1427
    assert po.decide_bit(0, q) == 0 and po.decide_bit(q // 2, q) == 1
    for r in range(q):
        assert co.lib().qfo_decide_bit(r, q) == po.decide_bit(r, q)
    assert po.decide_bit(4, 16) == 0  # equidistant from 0 and 8 -> 0
    assert (-(2 ** 63)) % 2 == 0 and (2 ** 63 - 1) % 2 == 1
    assert po.message_to_mu_bits(65536, 16) == [0] * 16  # ring_lpr.rs:655-667
    assert co.regev_cycle(50, 867, 124991, 96, SEED, 1) == 1
    assert co.regev_cycle(13, 154, 1427, 5, SEED, 0) == 0


def test_multithreaded_variants_agree():
    n, m, q, B = 20, 100, 7993, 37
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m)
    xb = co.sample_bits(SEED, 3, 0, B, m, 16)
    c1 = co.matmul_batch(pk, q, xb, co.X_BITS, B)
    assert np.array_equal(c1, co.matmul_batch(pk, q, xb, co.X_BITS, B, mt=True))
    s = co.sample_uniform(SEED, 2, 0, n, q)
    assert np.array_equal(co.dot_decode_batch(s, q, c1), co.dot_decode_batch(s, q, c1, mt=True))


def test_fixed_classical_trapdoor_kat():
    """short_basis_classical.rs:281-303 with the is_trapdoor identity (gadget_classical.rs:300-322):
    A * [R ; I] = I_2 (x) (1,2,4) mod 8 -- a known answer for the Z_q matrix product (the arithmetic
    of kernels a/c) from the reference's own fixed test data."""
    k = KATS["classical_trapdoor_n2_q8"]
    q = k["q"]
    A = np.array(k["A"], np.uint32)
    T = np.vstack([np.array(k["R"], np.int64), np.eye(6, dtype=np.int64)])          # 13 x 6
    G = np.array(k["G"], np.uint32)
    for j in range(6):
        col = (T[:, j] % q).astype(np.uint32)
        assert co.matvec(A, col, q).tolist() == G[:, j].tolist()
        assert po.matvec(A.tolist(), col.tolist(), q) == G[:, j].tolist()
    cols = np.ascontiguousarray(T.T).astype(np.int32)                               # 6 "messages" of 13 entries
    assert np.array_equal(co.matmul_batch(A, q, cols, co.X_I32, 6), G.T)


def ring_sa_r_products(kats):
    """The ring products the reference's literal gen_sa_r output pins (short_basis_ring.rs:404-439; fixture
    ring_sa_r_n4_q16): (factor a, factor b, a*b mod (X^4+1, 16)) triples read off the 24 x 24 matrix.
      * every X^s * w_j[b] for the base-2 digit polynomials w of -a_0, -a_1 (columns 16..23),
      * every X^s * (entry of S'') for the monomials 2 and -1 (columns 0..15),
      * (-1) * a_j itself, recomposed from its digits: sum_b 2^b w_j[b] = -a_j  (compute_w :128-139)."""
    k, t = kats["ring_sa_r_n4_q16"], kats["ring_trapdoor_n4_q16"]
    n, q, kk = k["n"], k["q"], k["k"]
    M = np.array(k["matrix"], np.int64)
    poly = lambda i, col: M[n * i:n * i + n, col]           # coefficients of polynomial row i in column col
    mono = lambda s: np.eye(n, dtype=np.int64)[s]
    out = []
    for j in range(2):
        digits = [poly(2 + b, 16 + j) for b in range(kk)]
        for b in range(kk):
            for s in range(n):
                out.append((digits[b] % q, mono(s), poly(2 + b, 16 + 2 * s + j) % q))
        a_j = np.array(list(t["a"][j]) + [0] * (n - len(t["a"][j])), np.int64)
        neg = sum((1 << b) * digits[b] for b in range(kk)) % q
        out.append((a_j % q, (q - 1) * mono(0), neg))        # (-1) * a_j in R_16
    for s in range(n):
        for tcol in range(kk):
            for i in range(kk):
                out.append((poly(2 + i, tcol) % q, mono(s), poly(2 + i, kk * s + tcol) % q))
    return n, q, out


def test_reference_gen_sa_r_ring_output():
    """working_sa_r (short_basis_ring.rs:404-439): the literal matrix the reference prints -- every ring
    product it contains, through both oracles."""
    n, q, cases = ring_sa_r_products(KATS)
    assert len(cases) == 2 * (4 * 4 + 1) + 4 * 4 * 4
    a = np.array([c[0] for c in cases], np.uint32)
    b = np.array([c[1] for c in cases], np.uint32)
    want = np.array([c[2] for c in cases], np.uint32)
    assert np.array_equal(co.polymul_batch(a, b, q), want)
    for x, y, w in cases:
        assert po.polymul_negacyclic([int(v) for v in x], [int(v) for v in y], q) == [int(v) for v in w]
    assert any((w != 0).any() and (w > 8).any() for _, _, w in cases)      # the -1 entries are really exercised


def test_reference_gen_sa_r_classical_output():
    """working_sa_r_identity (short_basis_classical.rs:339-356): the literal 13 x 13 matrix [0 | I ; S' | W];
    its W block satisfies G*W = -A*[I_7 | 0] mod 8 (compute_w :114-119) -- two Z_q matrix products."""
    k = KATS["classical_trapdoor_n2_q8"]
    q, A, G = k["q"], np.array(k["A"], np.uint32), np.array(k["G"], np.uint32)
    M = np.array(k["sa_r_identity"], np.int64)
    assert np.array_equal(M[:7, 6:], np.eye(7, dtype=np.int64)) and not M[:7, :6].any()
    W = M[7:, 6:]                                                        # 6 x 7
    E = -np.vstack([np.eye(7, dtype=np.int64), np.zeros((6, 7), np.int64)])   # -[I_7 ; 0]: 13 x 7
    lhs = co.matmul_batch(G, q, np.ascontiguousarray(W.T).astype(np.int32), co.X_I32, 7)   # rows: (G*W_j)^t
    rhs = co.matmul_batch(A, q, np.ascontiguousarray(E.T).astype(np.int32), co.X_I32, 7)   # rows: (-A e_j)^t
    assert np.array_equal(lhs, rhs)
    assert np.array_equal(rhs.T, (-A[:, :7].astype(np.int64)) % q)
    for j in range(7):
        assert po.matvec(G.tolist(), [int(v) % q for v in W[:, j]], q) == [int(v) for v in rhs[j]]
    # S' = I_2 (x) S_k with G*S' = 0 mod 8 (the gadget's own short basis, compute_s :89-111)
    S = M[7:, :6]
    assert not co.matmul_batch(G, q, np.ascontiguousarray(S.T).astype(np.int32), co.X_I32, 6).any()


def test_oracle_discrete_gauss_distribution():
    """The oracle's inversion sampler draws D_{Z,s}: support, mean, variance s^2/(2 pi).  (Parity with the
    reference's sampler is distributional only -- its RNG is unseeded, qfall-math is not in the tree.)"""
    q, n, s, N = 1073741789, 1024, 1073741789 * 6.25e-6, 200000
    x = co.sample_discrete_gauss(SEED, 4, 0, N, q, n, s).astype(np.int64)
    x[x > q // 2] -= q
    sigma = s / np.sqrt(2 * np.pi)
    assert np.abs(x).max() <= int(np.ceil(s * np.log2(n)))
    assert abs(x.mean()) < 5 * sigma / np.sqrt(N)
    assert abs(x.var() / sigma ** 2 - 1) < 0.02
    # consecutive windows of one stream are consistent (counter mode)
    assert np.array_equal(co.sample_discrete_gauss(SEED, 4, 100, 50, q, n, s), co.sample_discrete_gauss(SEED, 4, 0, 150, q, n, s)[100:])
    with pytest.raises(ValueError):
        co.sample_discrete_gauss(SEED, 4, 0, 1, q, 1024, 1e7)      # support beyond 2^21


def _replay():
    from crypto_amd.math import MatZ, MatZq, parse_poly
    path = os.path.join(os.path.dirname(__file__), "golden", "replay", "fixtures.txt")
    f = dict(l.split("=", 1) for l in open(path).read().splitlines() if "=" in l)
    return f, MatZq, MatZ, parse_poly


def test_replay_fixtures_match_the_c_oracle():
    """tests/golden/replay/fixtures.txt (qfall-math's text formats, for replay through the real reference:
    tests/golden/make_replay.py) parsed with the host layer's from_str and recomputed by the C oracle."""
    f, MatZq, MatZ, parse_poly = _replay()
    pk, x, c = MatZq.from_str(f["regev_pk"]), MatZ.from_str(f["regev_x"]), MatZq.from_str(f["regev_c"])
    q = pk.q
    y = co.matvec(pk.entries.astype(np.uint32), x.entries[:, 0].astype(np.uint32), q).astype(np.int64)
    y[-1] = (y[-1] + q // 2) % q
    assert y.tolist() == [int(v) for v in c.entries[:, 0]]
    A, v, d = MatZq.from_str(f["sis_key"]), MatZq.from_str(f["sis_value"]), MatZq.from_str(f["sis_digest"])
    assert co.matvec(A.entries.astype(np.uint32), v.entries[:, 0].astype(np.uint32), A.q).tolist() == [int(t) for t in d.entries[:, 0]]
    pk, r, e, c = (MatZq.from_str(f[k]) for k in ("lpr_pk", "lpr_r", "lpr_e", "lpr_c"))
    q = pk.q
    y = (co.matvec(pk.entries.astype(np.uint32), r.entries[:, 0].astype(np.uint32), q).astype(np.int64) + e.entries[:, 0].astype(np.int64)) % q
    y[-1] = (y[-1] + q // 2) % q
    assert y.tolist() == [int(t) for t in c.entries[:, 0]]
    n, q = int(f["ring_n"]), int(f["ring_q"])
    pad = lambda p: np.array([p + [0] * (n - len(p))], np.uint32)
    a, b, cc = (pad(parse_poly(f[k])) for k in ("ring_a", "ring_b", "ring_c"))
    assert np.array_equal(co.polymul_batch(a, b, q), cc)


def test_replay_fixtures_round3_products_match_the_c_oracle():
    """The fixtures added in round 3 (every remaining product of SURVEY 8a: Regev::dec with the tie of the decision
    rule, DualRegev enc/dec, RingLPR enc/dec, PSFGPVRing::f_a at n = 8 / q = 512, hash_to_mat_zq_sha256): the
    expected values were written by the exact Python oracle; here the C restatement recomputes every one."""
    from crypto_amd.math import MatPolyOverZ
    f, MatZq, MatZ, parse_poly = _replay()
    u32 = lambda m: np.ascontiguousarray(m.entries.astype(np.uint32))
    # Regev::dec
    sk = MatZq.from_str(f["regev_dec_sk"])
    q = sk.q
    want = [int(b) for b in f["regev_dec_bits"].split()]
    Cm = np.stack([u32(MatZq.from_str(f[f"regev_dec_c{k}"]))[:, 0] for k in range(int(f["regev_dec_count"]))])
    assert co.dot_decode_batch(u32(sk)[:, 0], q, Cm).tolist() == want
    s_ = [int(v) for v in u32(sk)[:, 0]]
    inner = lambda c: (sum((-a) * int(b) for a, b in zip(s_, c[:-1])) + int(c[-1])) % q
    assert inner(Cm[2]) == (q + q // 2) // 2 and want[2] == 0          # the tie decides 0
    assert [inner(Cm[0]), inner(Cm[1])] == [0, q // 4]
    # DualRegev enc / dec
    pk, s, e = (MatZq.from_str(f[k]) for k in ("dualregev_pk", "dualregev_s", "dualregev_e"))
    x = MatZ.from_str(f["dualregev_sk"])
    q = pk.q
    pkT = np.ascontiguousarray(u32(pk).T)
    for msg in (0, 1):
        c = co.matmul_batch(pkT, q, u32(s).T.copy(), co.X_U32, 1, addend=u32(e).T.copy(), msg=np.array([msg], np.uint8))
        assert c[0].tolist() == [int(t) for t in MatZq.from_str(f[f"dualregev_c{msg}"]).entries[:, 0]]
        assert co.dualregev_decode_batch(x.entries[:, 0].astype(np.int32), q, c).tolist() == [msg]
    # RingLPR enc / dec
    n, q = int(f["ringlpr_n"]), int(f["ringlpr_q"])
    P = lambda k: np.array(parse_poly(f[k]) + [0] * (n - len(parse_poly(f[k]))), np.uint32)
    mu_bits = (P("ringlpr_mu_q_half") != 0).astype(np.uint8)
    packed = np.packbits(mu_bits[None, :], axis=1, bitorder="little")
    u, v = co.ringlpr_enc_batch(P("ringlpr_a"), P("ringlpr_b"), q, P("ringlpr_r")[None], P("ringlpr_e1")[None], P("ringlpr_e2")[None], packed)
    assert np.array_equal(u[0], P("ringlpr_u")) and np.array_equal(v[0], P("ringlpr_v"))
    as_int = lambda b: int.from_bytes(bytes(b), "little")
    assert as_int(co.ringlpr_dec_batch(P("ringlpr_sk"), q, u, v)[0]) == int(f["ringlpr_message"])
    assert as_int(co.ringlpr_dec_batch(P("ringlpr_sk"), q, np.zeros((1, n), np.uint32), P("ringlpr_tie_v")[None])[0]) == int(f["ringlpr_tie_message"])
    assert (int(f["ringlpr_tie_message"]) >> 11) & 1 == 0 and P("ringlpr_tie_v")[11] == (q + q // 2) // 2    # the tie coefficient decides 0
    # f_a
    n, q = int(f["fa_n"]), int(f["fa_q"])
    A = MatPolyOverZ.from_str(f["fa_a"], n).entries.astype(np.uint32)
    sig = MatPolyOverZ.from_str(f["fa_sigma"], n).entries.astype(np.int32)
    t_want = MatPolyOverZ.from_str(f["fa_t"], n).entries.astype(np.uint32)
    assert A.shape == (11, n) and sig.shape == (11, n)
    t, ok = co.fdh_ring_verify_batch(A, q, sig[None], t_want, 100 * 100 * n * 11)
    assert np.array_equal(t, t_want) and ok.tolist() == [1]
    # hash_to_mat_zq_sha256
    assert co.hash_to_mat_zq(f["hash_msg"], 8, 1, 512).tolist() == MatZq.from_str(f["hash_8x1_512"]).entries.astype(np.int64).tolist()
    assert co.hash_to_mat_zq(f["hash_msg"], 3, 2, 1073479681).tolist() == MatZq.from_str(f["hash_3x2_big"]).entries.astype(np.int64).tolist()


def test_replay_rust_test_covers_every_fixture_key():
    """replay_test.rs (uncompiled here) must read every key make_replay.py writes, and nothing else."""
    import re
    f, *_ = _replay()
    rs = open(os.path.join(os.path.dirname(__file__), "golden", "replay", "replay_test.rs")).read()
    used = set(re.findall(r'f\["([a-z0-9_]+)"\]', rs)) | set(re.findall(r'&f, "([a-z0-9_]+)"', rs))
    fmt = re.findall(r'format!\("([a-z_]+)\{', rs)                     # regev_dec_c{k}, dualregev_c{msg}
    covered = {k for k in f if k in used or any(k.startswith(p) for p in fmt)}
    assert not set(f) - covered - {"regev_dec_count"}, sorted(set(f) - covered)
    assert used <= set(f), sorted(used - set(f))
