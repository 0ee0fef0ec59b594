"""GPU: scheme-level tests that read like the reference's own (#[cfg(test)] modules cited per
test), driven through the Python mirror of the trait API; every ciphertext / digest is also
compared bit for bit with the CPU oracle on the same seeded randomness."""
import numpy as np
import pytest

from oracle import coracle as co
from oracle import pyoracle as po
from crypto_amd.construction.pk_encryption import LPR, DualRegev, Regev, RingLPR
from crypto_amd.construction.hash import SISHash
from crypto_amd.construction.signature import FDH
from crypto_amd.math import MatPolyOverZ, MatZq
from crypto_amd.primitive.psf import GadgetParametersRing, PSFGPVRing
from crypto_amd.device import QfError, unpack_bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make", [Regev.default, lambda **k: Regev.new_from_n(50, **k),
                                  LPR.default, lambda **k: LPR.new_from_n(50, **k),
                                  DualRegev.default, lambda **k: DualRegev.new_from_n(20, **k)])
def test_cycle_zero_and_one(ctx, make):
    """regev.rs:547-595, dual_regev.rs:545-593, lpr.rs:564-616: gen -> enc -> dec == msg."""
    scheme = make(ctx=ctx)
    pk, sk = scheme.gen()
    for msg in (0, 1, 0, 1):
        assert scheme.dec(sk, scheme.enc(pk, msg)) == msg


def test_regev_modulo_two_semantics(ctx):
    """regev.rs:598-612: the message is reduced mod 2, also for i64::MAX / i64::MIN."""
    r = Regev.default(ctx=ctx)
    pk, sk = r.gen()
    assert r.dec(sk, r.enc(pk, 2 ** 63 - 1)) == 1
    assert r.dec(sk, r.enc(pk, -(2 ** 63))) == 0
    assert r.dec(sk, r.enc(pk, 3)) == 1


@pytest.mark.parametrize("cls", [Regev, LPR, DualRegev])
def test_multi_bit(ctx, cls):
    """regev.rs:622-669: LSB-first multi-bit encryption of |message|."""
    s = cls.default(ctx=ctx)
    pk, sk = s.gen()
    for msg in (15, 3, 0, 255, -9):
        assert s.dec_multiple_bits(sk, s.enc_multiple_bits(pk, msg)) == abs(msg)


def test_regev_single_and_batch_agree_with_oracle(ctx):
    r = Regev.new_from_n(20, ctx=ctx)
    pk, sk = r.gen()
    x = r.rng.bits(3, r.m)
    c = r.enc(pk, 1, x=x)
    assert c.entries[:, 0].tolist() == po.regev_enc(pk.entries.tolist(), x.tolist(), 1, r.q)
    B = 70
    msgs = np.arange(B) % 2
    xb = co.sample_bits(7, 3, 0, B, r.m, ((r.m + 127) // 128) * 16)
    cb = ctx.to_host_u32(r.enc_batch(pk, msgs, x_bits=xb))
    assert np.array_equal(cb, co.matmul_batch(pk.entries, r.q, xb, co.X_BITS, B, msg=msgs.astype(np.uint8)))
    assert r.dec_batch(sk, cb).tolist() == msgs.tolist()
    assert r.dec_batch(sk, cb).tolist() == co.dot_decode_batch(sk.entries[:, 0], r.q, cb).tolist()


def test_regev_gen_on_device_matches_oracle(ctx):
    """regev.rs:350-374 with the seeded generator: gen draws A, s, e and forms b^t = s^t A + e^t on the
    device; the oracle redoes it from the same seed.  The key planes cached by gen serve the first enc."""
    seed = 0xC0FFEE
    r = Regev.new_from_n(30, ctx=ctx, seed=seed, noise_model="irwin_hall")
    pk, sk = r.gen()
    n, m, q = r.n, r.m, r.q
    A = co.sample_uniform(seed, 1, 0, n * m, q).reshape(n, m)
    s_ = co.sample_uniform(seed, 2, 0, n, q)
    e = co.sample_noise(seed, 4, 0, m, q, r.noise_width)
    b = ((co.vecmat(s_, A, q).astype(np.uint64) + e) % q).astype(np.uint32)
    assert np.array_equal(pk.entries, np.vstack([A, b[None, :]]))
    assert np.array_equal(sk.entries[:, 0], s_)
    assert (id(ctx), False) in pk._dev
    x = r.rng.bits(3, m)
    assert r.enc(pk, 1, x=x).entries[:, 0].tolist() == po.regev_enc(pk.entries.tolist(), x.tolist(), 1, q)
    pk2, sk2 = r.gen()                      # the streams advance: a second key pair is a different one
    assert not np.array_equal(pk2.entries, pk.entries) and r.dec(sk2, r.enc(pk2, 1)) == 1


def test_lpr_batch_round_trip_and_oracle(ctx):
    l = LPR.new_from_n(64, ctx=ctx)
    pk, sk = l.gen()
    B = 100
    msgs = (np.arange(B) * 7 // 3) % 2
    W = l.noise_width
    rr = co.sample_noise(9, 3, 0, B * l.n, l.q, W).reshape(B, l.n)
    ee = co.sample_noise(9, 4, 0, B * (l.n + 1), l.q, W).reshape(B, l.n + 1)
    c = ctx.to_host_u32(l.enc_batch(pk, msgs, r=rr, e=ee))
    assert np.array_equal(c, co.matmul_batch(pk.entries, l.q, rr, co.X_U32, B, addend=ee, msg=msgs.astype(np.uint8)))
    assert l.dec_batch(sk, c).tolist() == msgs.tolist()
    c2 = l.enc_batch(pk, msgs)                     # device-generated randomness
    assert l.dec_batch(sk, c2).tolist() == msgs.tolist()


def test_dual_regev_batch_round_trip_and_oracle(ctx):
    d = DualRegev.new_from_n(16, ctx=ctx)
    pk, sk = d.gen()
    B = 40
    msgs = np.arange(B) % 2
    ss = co.sample_uniform(9, 3, 0, B * d.n, d.q).reshape(B, d.n)
    ee = co.sample_noise(9, 4, 0, B * (d.m + 1), d.q, d.noise_width).reshape(B, d.m + 1)
    c = ctx.to_host_u32(d.enc_batch(pk, msgs, s=ss, e=ee))
    ref = co.matmul_batch(np.ascontiguousarray(pk.entries.T), d.q, ss, co.X_U32, B, addend=ee, msg=msgs.astype(np.uint8))
    assert np.array_equal(c, ref)
    assert d.dec_batch(sk, c).tolist() == msgs.tolist()
    assert d.dec_batch(sk, c).tolist() == co.dualregev_decode_batch(sk.entries[:, 0], d.q, c).tolist()


def test_key_dimension_and_modulus_mismatch_panic(ctx):
    """qfall-math panics on mismatching dimensions / moduli; we raise."""
    r = Regev.default(ctx=ctx)
    pk, _ = r.gen()
    with pytest.raises(ValueError):
        Regev(13, 150, 1427, 0.01, ctx=ctx).enc(pk, 1)
    with pytest.raises(ValueError):
        Regev(13, 154, 1429, 0.01, ctx=ctx).enc(pk, 1)
    key = ctx.key_upload(pk.entries, r.q)
    with pytest.raises(QfError):
        ctx.matvec(key, np.zeros(10, np.uint32))
    with pytest.raises(QfError):
        ctx.key_upload(pk.entries, 2 ** 31)          # q >= 2^31
    with pytest.raises(QfError):
        ctx.key_upload(pk.entries, 1)
    # empty batch: no launch, empty result
    out = ctx.matmul(key, ctx.empty_u8(0, 32), 0, 0)
    assert tuple(out.shape) == (0, 14)


@pytest.mark.parametrize("make,messages", [
    (RingLPR.default, [0, 1, 2, 15, 212, 65535, 255, 13]),                 # ring_lpr.rs:608-630 (n=16)
    (lambda **k: RingLPR.new_from_n(64, **k), [0, 1, 2 ** 63 - 1, 2 ** 64 - 1, 12345678901234567])])  # :633-652
def test_ring_lpr_cycle(ctx, make, messages):
    s = make(ctx=ctx)
    pk, sk = s.gen()
    for m in messages:
        assert s.dec(sk, s.enc(pk, m)) == m
    u, v = s.enc_batch(pk, messages)
    assert s.dec_batch(sk, u, v) == messages


def test_ring_lpr_message_wraps_mod_2n(ctx):
    """ring_lpr.rs:655-667: enc(65536) at n=16 decrypts to 0."""
    s = RingLPR.default(ctx=ctx)
    pk, sk = s.gen()
    assert s.dec(sk, s.enc(pk, 65536)) == 0
    assert s.dec(sk, s.enc(pk, -3)) == 3


def test_ring_lpr_n1024_ntt_friendly(ctx):
    """BASELINE config 5: n=1024, q = 2^30 - 2^18 + 1."""
    # alpha is taken 64x below new_from_n(1024)'s value: at n=1024 the reference's own alpha
    # (ring_lpr.rs:183-191, "found out by experience" for small n) gives decryption noise of
    # about q/3 and does not round-trip; the arithmetic parity below does not depend on alpha.
    s = RingLPR(1024, 1073479681, 1 / (5 * 32 * 1000 * 64), ctx=ctx, noise_model="irwin_hall")
    assert s.ring.uses_ntt
    pk, sk = s.gen()
    msgs = [0, 1, 2 ** 1024 - 1, 3 ** 600, 2 ** 1023 + 12345]
    u, v = s.enc_batch(pk, msgs)
    assert s.dec_batch(sk, u, v) == [m % 2 ** 1024 for m in msgs]
    ur, vr = co.ringlpr_enc_batch(pk[0].coeffs, pk[1].coeffs, s.q_int,
                                  co.sample_noise(s.rng.seed, 3, 0, 5 * 1024, s.q_int, s.noise_width).reshape(5, 1024),
                                  co.sample_noise(s.rng.seed, 4, 1024, 5 * 1024, s.q_int, s.noise_width).reshape(5, 1024),
                                  co.sample_noise(s.rng.seed, 4, 6 * 1024, 5 * 1024, s.q_int, s.noise_width).reshape(5, 1024),
                                  np.stack([s.mu_bits(m) for m in msgs]))
    assert np.array_equal(ctx.to_host_u32(u), ur) and np.array_equal(ctx.to_host_u32(v), vr)


def test_sis_hash(ctx):
    """sis.rs:237-278: dimensions and panics; digest compared with the oracle."""
    h = SISHash.gen(5, 18, 11, ctx=ctx)
    h.check_security()
    v = MatZq(co.sample_uniform(3, 6, 0, 18, 11)[:, None], 11)
    d = h.hash(v)
    assert d.get_num_rows() == 5 and d.get_num_columns() == 1 and d.q == 11
    assert d.entries[:, 0].tolist() == po.matvec(h.key.entries.tolist(), v.entries[:, 0].tolist(), 11)
    with pytest.raises(ValueError):
        h.hash(MatZq(np.zeros((1, 18), np.uint32), 11))       # row vector
    with pytest.raises(ValueError):
        h.hash(MatZq(np.zeros((17, 1), np.uint32), 11))       # wrong length
    with pytest.raises(ValueError):
        h.hash(MatZq(np.zeros((18, 1), np.uint32), 13))       # wrong modulus
    h2 = SISHash.gen(32, 32 * 23, 8380417, ctx=ctx)
    vals = co.sample_uniform(3, 6, 0, 50 * 736, 8380417).reshape(50, 736)
    assert np.array_equal(ctx.to_host_u32(h2.hash_batch(vals)), co.matmul_batch(h2.key.entries, 8380417, vals, co.X_U32, 50))
    bits = co.sample_bits(3, 3, 0, 50, 736, 96)
    assert np.array_equal(ctx.to_host_u32(h2.hash_batch(bits, binary=True)), co.matmul_batch(h2.key.entries, 8380417, bits, co.X_BITS, 50))


def test_psf_f_a_fixed_trapdoor_kat(ctx):
    """gadget_ring.rs:189-210 with the fixed trapdoor of short_basis_ring.rs:357-378:
    f_a(a, [e_i; r_i; unit_i]) == 2^i  in Z_16[X]/(X^4+1)."""
    a = [[1, 0, 0, 0], [2, 8, 8, 12], [11, 10, 7, 13], [9, 6, 6, 12], [6, 11, 1, 6], [3, 10, 2, 9]]
    r = [[-1, 7, 6, -8], [0, -2, 4], [0, 3, -4, 1], [6, 4, -1, 3]]
    e = [[-4, 8, -3, 7], [1, -2, 2, 4], [-6, 7, -5], [-7, 10, -12, -15]]
    psf = PSFGPVRing(GadgetParametersRing.init_default(4, 16), 1000, ctx=ctx)
    assert psf.mbar == 6
    sig = [MatPolyOverZ([e[i], r[i]] + [[1] if j == i else [] for j in range(4)], 4) for i in range(4)]
    t = psf.f_a_batch(a, sig)
    assert t.tolist() == [[1, 0, 0, 0], [2, 0, 0, 0], [4, 0, 0, 0], [8, 0, 0, 0]]


def test_psf_check_domain(ctx):
    """psf/gpv_ring.rs:347-437: shape and norm conditions."""
    psf = PSFGPVRing(GadgetParametersRing.init_default(8, 512), 3, ctx=ctx)
    n, mbar = 8, psf.mbar
    bound = psf.norm_bound()                                   # 9 * 88 = 792
    ok = MatPolyOverZ([[3] * n] * mbar, n)                     # norm^2 = 9*88 = 792 -> inside (<=)
    assert psf.check_domain(ok)
    too_long = MatPolyOverZ([[3] * n] * (mbar - 1) + [[3] * (n - 1) + [4]], n)   # 799 > 792
    assert not psf.check_domain(too_long)
    assert not psf.check_domain(MatPolyOverZ([[1] * n] * (mbar - 1), n))         # wrong length
    assert bound == 792


def test_fdh_ring_vfy(ctx):
    """signature/fdh.rs:115-123 + fdh/gpv_ring.rs:99-123 (N=8, q=512): a valid preimage verifies,
    a tampered one, a wrong message and an out-of-domain one do not.  (Preimages are built by
    solving for sigma_0 since a_0 = 1; the trapdoor sampler is out of scope.)"""
    n, q = 8, 512
    fdh = FDH.init_gpv_ring(n, q, 10 ** 4, ctx=ctx)
    mbar = fdh.psf.mbar
    pk = co.sample_uniform(5, 1, 0, mbar * n, q).reshape(mbar, n)
    pk[0] = 0; pk[0, 0] = 1
    msgs = [f"Hello World! {i}" for i in range(10)]
    sigs = []
    for i, m in enumerate(msgs):
        tail = (co.sample_uniform(5, 7, i * 1000, (mbar - 1) * n, 7).astype(np.int64) - 3).reshape(mbar - 1, n)
        part = po.fdh_ring_f_a(pk[1:].tolist(), tail.tolist(), q)
        u = po.hash_to_ring(m, n, q)
        s0 = [((u[k] - part[k] + q // 2) % q) - q // 2 for k in range(n)]
        sigs.append(MatPolyOverZ([s0] + tail.tolist(), n))
    assert fdh.vfy_batch(msgs, sigs, pk) == [True] * 10
    assert fdh.vfy(msgs[0], sigs[0], pk)
    assert not fdh.vfy(msgs[0], sigs[1], pk)                       # signature of another message
    bad = MatPolyOverZ(sigs[2].entries.copy(), n); bad.entries[3, 1] += 1
    assert not fdh.vfy(msgs[2], bad, pk)
    small = FDH.init_gpv_ring(n, q, 1, ctx=ctx)                    # s = 1: every sigma is too long
    assert not small.vfy(msgs[0], sigs[0], pk)
    # oracle agreement incl. t
    u_all = np.array([po.hash_to_ring(m, n, q) for m in msgs], np.uint32)
    sg = np.stack([s.entries for s in sigs]).astype(np.int32)
    _, ok_ref = co.fdh_ring_verify_batch(pk, q, sg, u_all, fdh.psf.norm_bound())
    assert ok_ref.tolist() == [1] * 10


@pytest.mark.parametrize("make", [Regev.default, lambda **k: Regev.new_from_n(50, **k), LPR.default,
                                  lambda **k: LPR.new_from_n(64, **k), lambda **k: DualRegev.new_from_n(20, **k),
                                  lambda **k: RingLPR.new_from_n(64, **k)])
def test_cycle_with_discrete_gaussian_noise(ctx, make):
    """The same round trips with chi = D_{Z, alpha*q} drawn on the device, the distribution the reference
    samples (regev.rs:356-363, lpr.rs:342-360,399-417, ring_lpr.rs:430-452) instead of the benchmark
    generator's stand-in: single messages and a batch decrypt to what was encrypted."""
    scheme = make(ctx=ctx, noise_model="discrete_gauss")
    assert scheme.rng.gauss is not None
    pk, sk = scheme.gen()
    if isinstance(scheme, RingLPR):
        msgs = [0, 1, 5, (1 << scheme.n) - 1]
        for m in msgs:
            assert scheme.dec(sk, scheme.enc(pk, m)) == m
        return
    for msg in (0, 1, 1, 0):
        assert scheme.dec(sk, scheme.enc(pk, msg)) == msg
    msgs = np.arange(300) % 2
    assert scheme.dec_batch(sk, scheme.enc_batch(pk, msgs)).tolist() == msgs.tolist()


def test_lpr_noise_digits_follow_the_model(ctx):
    s = LPR.new_from_n(1024, ctx=ctx, noise_model="irwin_hall")
    assert s.rng.gauss is None and s.noise_digits == 2
    assert LPR.new_from_n(1024, ctx=ctx).rng.gauss is not None      # the default is the reference's D_{Z, alpha*q}
    g = LPR.new_from_n(1024, ctx=ctx, noise_model="discrete_gauss")
    import math
    assert g.rng.noise_bound(g.noise_width) == math.ceil(float(g.alpha * g.q) * 10)
    assert g.noise_digits in (2, 3)
    with pytest.raises(ValueError):
        LPR.default(ctx=ctx, noise_model="binomial")


from crypto_amd.construction.pk_encryption import (DualRegevWithDiscreteGaussianRegularity as DualRegevDG,
                                                    RegevWithDiscreteGaussianRegularity as RegevDG)


@pytest.mark.parametrize("make", [RegevDG.default, lambda **k: RegevDG.new_from_n(3, **k), lambda **k: RegevDG.new_from_n(10, **k),
                                  RegevDG.secure128, DualRegevDG.default, lambda **k: DualRegevDG.new_from_n(5, **k),
                                  lambda **k: DualRegevDG.new_from_n(10, **k), DualRegevDG.secure128])
def test_gaussian_regularity_variants_cycle(ctx, make):
    """regev_discrete_gauss.rs / dual_regev_discrete_gauss.rs test_encryption modules (cycle_zero_*, cycle_one_*):
    gen -> enc -> dec == msg, single messages, multi-bit and a batch; the batch is also checked against the oracle."""
    s = make(ctx=ctx)
    pk, sk = s.gen()
    for msg in (0, 1, 1, 0):
        assert s.dec(sk, s.enc(pk, msg)) == msg
    assert s.dec_multiple_bits(sk, s.enc_multiple_bits(pk, 13)) == 13
    B = 200
    msgs = (np.arange(B) * 7 % 3 == 0).astype(np.int64)
    C = s.enc_batch(pk, msgs)
    assert s.dec_batch(sk, C).tolist() == msgs.tolist()
    # parity of the batch with the oracle on the same seeded draws
    if isinstance(s, RegevDG):
        K = np.vstack([pk[0].entries, pk[1].entries.T]).astype(np.uint32)
        pos = s.rng.pos[3] - B * s.m
        e = co.sample_discrete_gauss(s.rng.seed, 3, pos, B * s.m, s.q, s.n, float(s.r)).reshape(B, s.m)
        ref = co.matmul_batch(K, s.q, e, co.X_U32, B, msg=msgs.astype(np.uint8))
    else:
        K = np.ascontiguousarray(np.hstack([pk[0].entries, pk[1].entries]).astype(np.uint32).T)
        S = co.sample_uniform(s.rng.seed, 3, s.rng.pos[3] - B * s.n, B * s.n, s.q).reshape(B, s.n)
        X = co.sample_discrete_gauss(s.rng.seed, 4, s.rng.pos[4] - B * (s.m + 1), B * (s.m + 1), s.q, s.n,
                                     float(s.alpha * s.q)).reshape(B, s.m + 1)
        ref = co.matmul_batch(K, s.q, S, co.X_U32, B, addend=X, msg=msgs.astype(np.uint8))
    assert np.array_equal(ctx.to_host_u32(C), ref)


def test_gaussian_regularity_parameter_rules():
    """regev_discrete_gauss.rs test_pp_generation: new_from_n yields correct + secure sets, n <= 1 panics,
    the fixed sets pass their own checks."""
    for cls in (RegevDG, DualRegevDG):
        for n in (2, 3, 5, 8, 12, 25):
            p = cls.new_from_n(n, ctx=object())
            p.check_correctness(); p.check_security()
            assert p.m >= 2 * (n + 1) * np.log10(p.q)
        cls.default(ctx=object()).check_correctness()
        cls.secure128(ctx=object()).check_correctness()
        cls.secure128(ctx=object()).check_security()
        with pytest.raises(ValueError):
            cls.new_from_n(1, ctx=object())


def test_classical_gpv_psf_and_fdh_verify(ctx):
    """primitive/psf/gpv.rs:189-223 and signature/fdh.rs:115-123 with fdh/gpv.rs:40-60: f_a = A*sigma,
    check_domain, vfy = in domain && f_a == H(m).  The trapdoor sampler is out of scope, so A's last
    column is solved for so that a chosen short sigma (last entry 1) IS a preimage of H(m)."""
    from crypto_amd.construction.hash.sha256 import hash_to_mat_zq_sha256
    from crypto_amd.math import MatZ
    from crypto_amd.primitive.psf import PSFGPV, GadgetParameters
    n, q = 4, 64
    s = ((n * 6) ** 0.5 + 1) * 2 * np.log2(2 * n * 6)            # the reference test's choice (fdh/gpv.rs test)
    fdh = FDH.init_gpv(n, q, s, ctx=ctx)
    psf = fdh.psf
    assert isinstance(psf, PSFGPV) and psf.gp.k == 6 and psf.gp.m_bar == 4 * 6 + 4 and psf.m == 52
    rng = np.random.default_rng(5)
    msgs = ["Hello World!", "qfall", "", "x" * 100]
    sigmas, A = [], rng.integers(0, q, (n, psf.m)).astype(np.int64)
    sig = rng.integers(-3, 4, psf.m).astype(np.int64)
    sig[-1] = 1
    u = hash_to_mat_zq_sha256(msgs[0], n, 1, q).entries[:, 0].astype(np.int64)
    A[:, -1] = (u - A[:, :-1] @ sig[:-1]) % q
    Aq = MatZq(A, q)
    good = MatZ(sig[:, None])
    assert psf.check_domain(good)
    t = psf.f_a(Aq, good)
    assert t.entries[:, 0].tolist() == po.matvec(A.tolist(), [int(v) % q for v in sig], q) == u.tolist()
    assert fdh.vfy(msgs[0], good, Aq)
    assert not fdh.vfy(msgs[1], good, Aq)                               # other message
    big = MatZ((sig * 1000)[:, None])
    assert not psf.check_domain(big) and not fdh.vfy(msgs[0], big, Aq)  # out of domain
    assert not psf.check_domain(MatZ(sig[None, :])) and not psf.check_domain(MatZ(sig[:-1, None]))
    with pytest.raises(AssertionError):
        psf.f_a(Aq, big)
    res = fdh.vfy_batch(msgs, [good, good, big, good], Aq)
    assert res == [True, False, False, False]
    with pytest.raises(NotImplementedError):
        psf.samp_p(None, None, None)
