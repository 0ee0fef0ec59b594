"""GPU parity at BASELINE.json's FULL sizes.  The oracle cannot recompute these batches in seconds, so each
configuration is checked through size-independent properties of the arithmetic (enc -> dec round trips,
linearity of the product in its batch operand, commutativity / distributivity / the X-shift rule of the ring
product) over the whole batch, plus a bit-exact oracle comparison of a sample of its rows."""
import numpy as np
import pytest
import torch

from oracle import coracle as co
from crypto_amd.device import X_BITS, X_BITS_KT, X_U32

pytestmark = pytest.mark.gpu
SEED = 0x5146414C4C


def _i64(t):
    return t.to(torch.int64) & 0xFFFFFFFF


def _rows(B, k=48):
    return np.unique(np.concatenate([[0, 1, B // 2, B - 2, B - 1], np.random.default_rng(3).integers(0, B, k)]))


def _split_masks(ctx, stride):
    even = torch.full((stride,), 0x55, dtype=torch.uint8, device=ctx.device)
    return even, torch.full((stride,), 0xAA, dtype=torch.uint8, device=ctx.device)


@pytest.mark.parametrize("n,m,q,B", [(256, 6168, 16777213, 65536)])
def test_regev_config2(ctx, n, m, q, B):
    """BASELINE config 2: 65,536 encryptions + decryptions."""
    pk = ctx.sample_uniform(SEED, 1, 0, (n + 1) * m, q).view(n + 1, m)
    s = ctx.sample_uniform(SEED, 2, 0, n, q)
    e = ctx.sample_noise(SEED, 4, 0, m, q, 4)
    ka = ctx.key_upload(pk[:n].contiguous(), q)
    b = torch.from_numpy(ctx.vecmat(ka, ctx.to_host_u32(s)).astype(np.int64)).to(ctx.device)
    pk[n] = ((b + _i64(e)) % q).to(torch.int32)
    key = ctx.key_upload(pk, q)
    x = ctx.sample_bits(SEED, 3, 0, B, m)
    msg = (ctx.sample_uniform(SEED, 5, 0, B, 2) & 1).to(torch.uint8)
    c = ctx.matmul(key, x, X_BITS, B, msg=msg)
    w = ((q - _i64(s)) % q).to(torch.int32)
    assert torch.equal(ctx.dot_decode(w, q, c), msg)                         # round trip of every message
    ev, od = _split_masks(ctx, x.shape[1])
    c0 = ctx.matmul(key, x, X_BITS, B)
    ce, co_ = ctx.matmul(key, x & ev, X_BITS, B), ctx.matmul(key, x & od, X_BITS, B)
    assert torch.equal((_i64(ce) + _i64(co_)) % q, _i64(c0))                # linear in x
    rows = _rows(B)
    ref = co.matmul_batch(ctx.to_host_u32(pk), q, ctx.to_host_u8(x[rows]), co.X_BITS, len(rows), msg=ctx.to_host_u8(msg[rows]), mt=True)
    assert np.array_equal(ctx.to_host_u32(c[rows]), ref)
    # a batch that ends inside a message tile (B mod 256 = 93, mod 32 = 29) at the full K: every row of the partial last tile
    # against the oracle, the rest against the full batch, nothing written behind the batch
    Bp = 256 * 17 + 93
    guard = ctx.empty_u32(Bp + 1, n + 1)
    guard.fill_(-1)
    cp = ctx.matmul(key, x[:Bp], X_BITS, Bp, msg=msg[:Bp], out=guard[:Bp])
    assert torch.equal(cp, c[:Bp]) and bool((guard[Bp] == -1).all())
    last = np.arange(256 * 17, Bp)
    ref = co.matmul_batch(ctx.to_host_u32(pk), q, ctx.to_host_u8(x[last]), co.X_BITS, len(last), msg=ctx.to_host_u8(msg[last]), mt=True)
    assert np.array_equal(ctx.to_host_u32(cp[last]), ref)
    # the same bits k-tile-major (what bench.py keeps resident): drawn in that form and converted from the rows -- one image;
    # the whole batch and the batch ending inside a tile (planes of the full batch: rows behind Bp are simply not read)
    xk = ctx.sample_bits_kt(SEED, 3, 0, B, m)
    assert torch.equal(xk, ctx.bits_to_kt(x, B, m))
    assert torch.equal(ctx.matmul(key, xk, X_BITS_KT, B, msg=msg), c)
    guard.fill_(-1)
    assert torch.equal(ctx.matmul(key, xk, X_BITS_KT, Bp, msg=msg[:Bp], out=guard[:Bp]), c[:Bp]) and bool((guard[Bp] == -1).all())


@pytest.mark.timeout(300)
def test_sis_config3(ctx):
    """BASELINE config 3: 2^20 binary inputs, n = 512, m = 11,776."""
    n, q, B = 512, 8380417, 1 << 20
    m = n * 23
    A = ctx.sample_uniform(SEED, 1, 0, n * m, q).view(n, m)
    key = ctx.key_upload(A, q)
    x = ctx.sample_bits(SEED, 3, 0, B, m)
    d = ctx.matmul(key, x, X_BITS, B)
    ev, od = _split_masks(ctx, x.shape[1])
    de = ctx.matmul(key, x & ev, X_BITS, B)
    do = ctx.matmul(key, x & od, X_BITS, B)
    assert torch.equal((_i64(de) + _i64(do)) % q, _i64(d))
    del de, do
    rows = _rows(B, 24)
    ref = co.matmul_batch(ctx.to_host_u32(A), q, ctx.to_host_u8(x[rows]), co.X_BITS, len(rows), mt=True)
    assert np.array_equal(ctx.to_host_u32(d[rows]), ref)
    xk = ctx.bits_to_kt(x, B, m)                                              # the same inputs k-tile-major
    assert torch.equal(ctx.matmul(key, xk, X_BITS_KT, B), d)
    del xk
    # a collision of the hash: x and x' = x with two equal... (none to construct without a trapdoor) -- instead
    # homomorphism on Z_q inputs at the secondary size: A(v1 + v2) = A v1 + A v2
    B2 = 1 << 14
    v1 = ctx.sample_uniform(SEED, 6, 0, B2 * m, q).view(B2, m)
    v2 = ctx.sample_uniform(SEED, 6, B2 * m, B2 * m, q).view(B2, m)
    vs = ((_i64(v1) + _i64(v2)) % q).to(torch.int32)
    h1, h2, hs = (ctx.matmul(key, v, X_U32, B2) for v in (v1, v2, vs))
    assert torch.equal((_i64(h1) + _i64(h2)) % q, _i64(hs))


@pytest.mark.timeout(300)
def test_lpr_config4(ctx):
    """BASELINE config 4: 2^18 LPR encryptions + decryptions, n = 1024.  The round trip is run with a narrow
    noise (W = 40): at the width the reference's rule gives for n = 1024 (alpha*q = 6711, W = 2319) the
    decryption noise sum_i r_i e_i - s_i e_i has a standard deviation of 3.2e8 > q/4, i.e. the scheme itself
    mis-decrypts a large share of messages there -- a property of the parameter rule, not of the arithmetic
    (the bit-exact oracle comparison below uses that full width)."""
    n, q, B, W = 1024, 1073741789, 1 << 18, 40
    A = ctx.sample_uniform(SEED, 1, 0, n * n, q).view(n, n)
    s = ctx.sample_noise(SEED, 2, 0, n, q, W)
    e0 = ctx.sample_noise(SEED, 4, 0, n, q, W)
    ka = ctx.key_upload(A, q)
    b = torch.from_numpy(ctx.vecmat(ka, ctx.to_host_u32(s)).astype(np.int64)).to(ctx.device)
    pk = torch.cat([A, ((b + _i64(e0)) % q).to(torch.int32)[None, :]])
    key = ctx.key_upload(pk, q)
    r = ctx.sample_noise(SEED, 3, 0, B * n, q, W).view(B, n)
    e = ctx.sample_noise(SEED, 4, n, B * (n + 1), q, W).view(B, n + 1)
    msg = (ctx.sample_uniform(SEED, 5, 0, B, 2) & 1).to(torch.uint8)
    c = ctx.matmul(key, r, X_U32, B, addend=e, msg=msg, x_digits=2)
    w = ((q - _i64(s)) % q).to(torch.int32)
    assert torch.equal(ctx.dot_decode(w, q, c), msg)
    # the digit promise changes nothing: full-width product of the same operands
    c_full = ctx.matmul(key, r, X_U32, B, addend=e, msg=msg)
    assert torch.equal(c_full, c)
    del c_full
    # full-width noise of the reference's parameter rule: 2-digit promise == full-width product == oracle rows
    Wf = 2319
    r = ctx.sample_noise(SEED, 3, 0, B * n, q, Wf, out=r.view(-1)).view(B, n)
    e = ctx.sample_noise(SEED, 4, n, B * (n + 1), q, Wf, out=e.view(-1)).view(B, n + 1)
    c = ctx.matmul(key, r, X_U32, B, addend=e, msg=msg, x_digits=2, out=c)
    assert torch.equal(ctx.matmul(key, r, X_U32, B, addend=e, msg=msg), c)
    rows = _rows(B, 24)
    ref = co.matmul_batch(ctx.to_host_u32(pk), q, ctx.to_host_u32(r[rows]), co.X_U32, len(rows), addend=ctx.to_host_u32(e[rows]),
                          msg=ctx.to_host_u8(msg[rows]), mt=True)
    assert np.array_equal(ctx.to_host_u32(c[rows]), ref)
    # a batch that ends inside a message tile at the full K (digit-plane operands on the 16x16x64 body): the rows of the
    # partial last tile against the oracle, the rest against the full batch, nothing written behind the batch
    Bp = 256 * 9 + 93
    guard = ctx.empty_u32(Bp + 1, n + 1)
    guard.fill_(-1)
    cp = ctx.matmul(key, r[:Bp], X_U32, Bp, addend=e[:Bp], msg=msg[:Bp], x_digits=2, out=guard[:Bp])
    assert torch.equal(cp, c[:Bp]) and bool((guard[Bp] == -1).all())
    last = np.arange(256 * 9 + 60, Bp)
    ref = co.matmul_batch(ctx.to_host_u32(pk), q, ctx.to_host_u32(r[last]), co.X_U32, len(last), addend=ctx.to_host_u32(e[last]),
                          msg=ctx.to_host_u8(msg[last]), mt=True)
    assert np.array_equal(ctx.to_host_u32(cp[last]), ref)


@pytest.mark.timeout(400)
def test_ring_config5(ctx):
    """BASELINE config 5: 2^20 ring products, RingLPR on 2^18 messages, 32,768 FDH verifications (n = 1024)."""
    n, q = 1024, 1073479681
    ring = ctx.ring(n, q)
    B = 1 << 20
    a = ctx.sample_uniform(SEED, 6, 0, B * n, q).view(B, n)
    b = ctx.sample_uniform(SEED, 6, B * n, B * n, q).view(B, n)
    ab = ring.polymul(a, b)
    assert torch.equal(ring.polymul(b, a), ab)                              # commutative
    # X * a is the negacyclic shift of a: (X a) b = X (a b)
    xa = torch.cat([((q - _i64(a[:, -1:])) % q).to(torch.int32), a[:, :-1]], dim=1).contiguous()
    xab = torch.cat([((q - _i64(ab[:, -1:])) % q).to(torch.int32), ab[:, :-1]], dim=1)
    assert torch.equal(ring.polymul(xa, b), xab)
    del xa, xab
    rows = _rows(B, 12)
    assert np.array_equal(ctx.to_host_u32(ab[rows]), co.polymul_batch(ctx.to_host_u32(a[rows]), ctx.to_host_u32(b[rows]), q))
    del ab
    # distributive on a quarter of the batch
    Bq = B // 4
    c = ctx.sample_uniform(SEED, 6, 2 * B * n, Bq * n, q).view(Bq, n)
    bc = ((_i64(b[:Bq]) + _i64(c)) % q).to(torch.int32)
    lhs = ring.polymul(a[:Bq].contiguous(), bc)
    rhs = (_i64(ring.polymul(a[:Bq].contiguous(), b[:Bq].contiguous())) + _i64(ring.polymul(a[:Bq].contiguous(), c))) % q
    assert torch.equal(_i64(lhs), rhs)
    del a, b, c, bc, lhs, rhs

    # RingLPR: 2^18 messages of n bits each, encrypt -> decrypt
    B = 1 << 18
    W = 40      # narrow noise for the round trip (see test_lpr_config4); parity rows below are width-independent
    av = ctx.sample_uniform(SEED, 1, 0, n, q)
    s = ctx.sample_noise(SEED, 2, 0, n, q, W)
    e = ctx.sample_noise(SEED, 4, 0, n, q, W)
    bp = ((_i64(ring.polymul(av[None].contiguous(), s[None].contiguous())[0]) + _i64(e)) % q).to(torch.int32)
    pk = ring.key(torch.stack([av, bp]))
    sk = ring.key(s[None].contiguous())
    r = ctx.sample_noise(SEED, 3, 0, B * n, q, W).view(B, n)
    e1 = ctx.sample_noise(SEED, 4, n, B * n, q, W).view(B, n)
    e2 = ctx.sample_noise(SEED, 4, n + B * n, B * n, q, W).view(B, n)
    mu = ctx.sample_bits(SEED, 5, 0, B, n, n // 8)
    u, v = ring.ringlpr_enc(pk, r, e1, e2, mu)
    assert torch.equal(ring.ringlpr_dec(sk, u, v), mu)
    rows = _rows(B, 8)
    ur, vr = co.ringlpr_enc_batch(ctx.to_host_u32(av), ctx.to_host_u32(bp), q, ctx.to_host_u32(r[rows]), ctx.to_host_u32(e1[rows]),
                                  ctx.to_host_u32(e2[rows]), ctx.to_host_u8(mu[rows]))
    assert np.array_equal(ctx.to_host_u32(u[rows]), ur) and np.array_equal(ctx.to_host_u32(v[rows]), vr)
    del r, e1, e2, u, v

    # FDH: 32,768 signatures of k+2 = 32 polynomials; t is additive in sigma, the verdict follows t == u and the norm
    B, mbar = 1 << 15, 32
    A = ctx.sample_uniform(SEED, 1, 5000, mbar * n, q).view(mbar, n)
    ak = ring.key(A)
    s1 = (_i64(ctx.sample_uniform(SEED, 7, 0, B * mbar * n, 201)) - 100).to(torch.int32).view(B, mbar, n)
    s2 = (_i64(ctx.sample_uniform(SEED, 7, B * mbar * n, B * mbar * n, 201)) - 100).to(torch.int32).view(B, mbar, n)
    zero = torch.zeros(B, n, dtype=torch.int32, device=ctx.device)
    big = 1 << 62
    _, t1 = ring.fdh_verify(ak, s1, zero, big, want_t=True)
    _, t2 = ring.fdh_verify(ak, s2, zero, big, want_t=True)
    ok12, t12 = ring.fdh_verify(ak, s1 + s2, ((_i64(t1) + _i64(t2)) % q).to(torch.int32), big, want_t=True)
    assert torch.equal(_i64(t12), (_i64(t1) + _i64(t2)) % q) and bool(ok12.all())
    norms = (s1.to(torch.int64) ** 2).sum(dim=(1, 2))
    bound = int(norms.median())
    ok, _ = ring.fdh_verify(ak, s1, t1, bound, want_t=True)
    assert torch.equal(ok.to(torch.bool), norms <= bound)                   # exact domain test over the batch
    u_bad = t1.clone(); u_bad[::2, 7] = ((_i64(u_bad[::2, 7]) + 1) % q).to(torch.int32)
    ok, _ = ring.fdh_verify(ak, s1, u_bad, big, want_t=True)
    assert not bool(ok[::2].any()) and bool(ok[1::2].all())
    rows = _rows(B, 4)
    t_ref, _ = co.fdh_ring_verify_batch(ctx.to_host_u32(A), q, s1[rows].cpu().numpy(), np.zeros((len(rows), n), np.uint32), big)
    assert np.array_equal(ctx.to_host_u32(t1[rows]), t_ref)


@pytest.mark.timeout(300)
def test_dual_regev_secure128(ctx):
    """DualRegev at the reference's `secure128` shape (dual_regev.rs:301-303: n = 230, m = 5313, q = 7,764,299,
    alpha = 0.0011), 8,192 messages.  The batched shape is the opposite of every other config: the key is the
    transposed pk, (m+1) = 5,314 rows x n = 230 columns, the batch operand s is uniform in Z_q (3 digits) --
    167 row blocks over two k-tiles.  enc (dual_regev.rs:403-410) against the oracle on sampled rows, the full
    round trip through dec (dual_regev.rs:437-450: c^t * [-x ; 1]) with the reference's own noise width
    (W = round(alpha*q/sqrt(2 pi) * sqrt(3)/2) = 2951: decryption noise sigma 1.8e5 against q/4 = 1.9e6), and
    linearity of the product in s."""
    n, m, q, B, W = 230, 5313, 7764299, 8192, 2951
    A = ctx.sample_uniform(SEED, 1, 0, n * m, q).view(n, m)
    xb = ctx.sample_bits(SEED, 2, 0, 1, m)                                   # sk = x in {0,1}^m
    x_h = np.unpackbits(ctx.to_host_u8(xb), axis=1, bitorder="little")[0, :m].astype(np.uint32)
    ka = ctx.key_upload(A, q)
    u = ctx.matvec(ka, x_h)                                                 # u = A x (dual_regev.rs:354)
    assert np.array_equal(u, co.matvec(ctx.to_host_u32(A), x_h, q))
    pk_h = np.hstack([ctx.to_host_u32(A), u[:, None]])                      # n x (m+1)
    key = ctx.key_upload(pk_h, q, transpose=True)
    assert (key.rows, key.cols) == (m + 1, n)
    s = ctx.sample_uniform(SEED, 3, 0, B * n, q).view(B, n)
    e = ctx.sample_noise(SEED, 4, 0, B * (m + 1), q, W).view(B, m + 1)
    msg = (ctx.sample_uniform(SEED, 5, 0, B, 2) & 1).to(torch.uint8)
    c = ctx.matmul(key, s, X_U32, B, addend=e, msg=msg)
    w = ctx.to_dev_u32((q - x_h.astype(np.int64)) % q)                      # [-x ; 1]
    assert torch.equal(ctx.dot_decode(w, q, c), msg)                        # round trip of every message
    rows = _rows(B, 40)
    pkT = np.ascontiguousarray(pk_h.T)
    ref = co.matmul_batch(pkT, q, ctx.to_host_u32(s[rows]), co.X_U32, len(rows), addend=ctx.to_host_u32(e[rows]),
                          msg=ctx.to_host_u8(msg[rows]), mt=True)
    assert np.array_equal(ctx.to_host_u32(c[rows]), ref)
    assert np.array_equal(ctx.to_host_u8(ctx.dot_decode(w, q, c[rows].contiguous())),
                          co.dualregev_decode_batch(x_h.astype(np.int32), q, ref))
    # linear in s, no addend: pk^t (s1 + s2) = pk^t s1 + pk^t s2
    s2 = ctx.sample_uniform(SEED, 3, B * n, B * n, q).view(B, n)
    ss = ((_i64(s) + _i64(s2)) % q).to(torch.int32)
    h1, h2, hs = (ctx.matmul(key, v, X_U32, B) for v in (s, s2, ss))
    assert torch.equal((_i64(h1) + _i64(h2)) % q, _i64(hs))
    # a batch that ends inside a message tile (B mod 256 = 93, mod 32 = 29) at the full K: rows of the last tile
    Bp = 4096 + 93
    cp = ctx.matmul(key, s[:Bp], X_U32, Bp, addend=e[:Bp], msg=msg[:Bp])
    assert torch.equal(cp, c[:Bp])


@pytest.mark.parametrize("n,q", [(256, 1073479681), (1024, 1073479681), (256, 92897729), (1024, 16777213)])
def test_ring_product_by_kernel_a(ctx, n, q):
    """Kernel (b) cross-checked by kernel (a) on the device, no oracle in between (SURVEY 7 "an independent GPU
    cross-check"): Rot^-(a) * b = a * b in Z_q[X]/(X^n+1) (utils/rotation_matrix.rs:45-67).  The rotation matrix
    is built on the host (column j = coefficients of a * X^j), uploaded as a Z_q key, and the batch of b's goes
    through the int8-MFMA product; the same batch through qf_polymul_dev with a repeated.  NTT-friendly q (wave
    NTT kernel) and q with 2n not dividing q-1 (exact schoolbook kernel)."""
    from crypto_amd.utils.rotation_matrix import rot_minus
    B = 512
    a = ctx.sample_uniform(SEED, 6, 0, n, q)
    b = ctx.sample_uniform(SEED, 6, n, B * n, q).view(B, n)
    a_h = ctx.to_host_u32(a).astype(np.int64)
    rot = np.mod(rot_minus(a_h).astype(np.int64), q).astype(np.uint32)     # n x n, column j = a * X^j
    assert rot[0, 1] == (q - a_h[n - 1]) % q and rot[1, 0] == a_h[1]
    key = ctx.key_upload(rot, q)
    by_a = ctx.matmul(key, b, X_U32, B)
    ring = ctx.ring(n, q)
    assert ring.uses_ntt == ((q - 1) % (2 * n) == 0)
    by_b = ring.polymul(a[None].expand(B, n).contiguous(), b)
    assert torch.equal(by_a, by_b)
