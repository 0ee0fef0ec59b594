"""GPU parity tests proper: every kernel, called through the C ABI, against the CPU oracle
(oracle/qf_oracle.c) on the same seeded inputs.  Bit-exact -- this is integer work."""
import numpy as np
import pytest
import torch

from oracle import coracle as co
from crypto_amd import _lib
from crypto_amd.device import QfError, X_BITS, X_BITS_KT, X_I32, X_PLANES, X_U32, bit_stride, pack_bits

pytestmark = pytest.mark.gpu
SEED = 0x5146414C4C


def test_sampler_matches_oracle(ctx):
    q = 1073741789
    d = ctx.to_host_u32(ctx.sample_uniform(SEED, 1, 5, 4099, q))
    assert np.array_equal(d, co.sample_uniform(SEED, 1, 5, 4099, q))
    d = ctx.to_host_u32(ctx.sample_noise(SEED, 4, 7, 3001, q, 4096))
    assert np.array_equal(d, co.sample_noise(SEED, 4, 7, 3001, q, 4096))
    st = bit_stride(867)
    d = ctx.to_host_u8(ctx.sample_bits(SEED, 3, 11, 37, 867, st))
    assert np.array_equal(d, co.sample_bits(SEED, 3, 11, 37, 867, st))


@pytest.mark.parametrize("rows,cols,stride", [(1, 1, 16), (3, 64, 16), (5, 65, 32), (70000, 129, 32), (300, 6168, 784), (9, 867, 116),
                                              (2, 4096, 512), (4, 127, 20)])
def test_sample_bits_both_kernels(ctx, rows, cols, stride):
    """Bit rows: the 16-bytes-per-thread kernel (strides that are multiples of 16: the GEMM's operand layout)
    and the dword kernel (other strides), draw boundaries at 64/128 bits, more rows than grid.y can hold."""
    d = ctx.to_host_u8(ctx.sample_bits(SEED, 3, 5, rows, cols, stride))
    assert np.array_equal(d, co.sample_bits(SEED, 3, 5, rows, cols, stride))


@pytest.mark.parametrize("rows,cols,q", [(14, 154, 1427), (51, 867, 124991), (257, 6168, 16777213),
                                          (1, 3, 7), (33, 4097, 1073741789), (5, 18, 11)])
def test_matvec_vecmat(ctx, rows, cols, q):
    A = co.sample_uniform(SEED, 1, 0, rows * cols, q).reshape(rows, cols)
    key = ctx.key_upload(A, q)
    x = co.sample_uniform(SEED, 6, 0, cols, q)
    assert np.array_equal(ctx.matvec(key, x), co.matvec(A, x, q))
    s = co.sample_uniform(SEED, 2, 0, rows, q)
    assert np.array_equal(ctx.vecmat(key, s), co.vecmat(s, A, q))


@pytest.mark.parametrize("rows,cols,q", [(50, 867, 124991), (1, 3, 7), (130, 4099, 2147483629), (65, 1024, 8380417)])
def test_vecmat_u32_with_addend(ctx, rows, cols, q):
    """qf_vecmat_u32_dev: x^t*A + e^t on a plain device matrix (key generation, regev.rs:366-368), with
    operands that are arbitrary uint32 values, into a row of a larger device matrix."""
    A = co.sample_uniform(SEED, 1, 0, rows * cols, 2 ** 32 - 1).reshape(rows, cols)
    s = co.sample_uniform(SEED, 2, 0, rows, 2 ** 32 - 1)
    e = co.sample_uniform(SEED, 4, 0, cols, 2 ** 32 - 1)
    red = lambda v: (v.astype(np.uint64) % q).astype(np.uint32)
    want0 = co.vecmat(red(s), red(A), q)
    want = ((want0.astype(np.uint64) + red(e)) % q).astype(np.uint32)
    img = ctx.empty_u32(rows + 1, cols)
    img[:rows].copy_(ctx.to_dev_u32(A))
    ctx.vecmat_u32(img[:rows], q, ctx.to_dev_u32(s), addend=ctx.to_dev_u32(e), out=img[rows])
    assert np.array_equal(ctx.to_host_u32(img[rows]), want)
    assert np.array_equal(ctx.to_host_u32(ctx.vecmat_u32(img[:rows], q, ctx.to_dev_u32(s))), want0)


@pytest.mark.parametrize("U", [1, 2, 3, 4])
@pytest.mark.parametrize("rows,cols,q", [(512, 11776, 8380417), (257, 6168, 16777213), (70, 1300, 2147483629), (33, 50, 251)])
def test_matvec_every_chunking(ctx, monkeypatch, rows, cols, q, U):
    """Kernel (c) on the digit planes with 1..4 k-steps per wave (QF_MATVEC_U): every grid shape the
    launcher can pick, partial last chunks, 1..4 digits per entry, x with arbitrary uint32 values."""
    monkeypatch.setenv("QF_MATVEC_U", str(U))
    A = co.sample_uniform(SEED, 1, 7, rows * cols, q).reshape(rows, cols)
    key = ctx.key_upload(A, q)
    x = co.sample_uniform(SEED, 6, U, cols, 2 ** 32 - 1)            # any uint32 is reduced mod q
    want = co.matvec(A, (x.astype(np.uint64) % q).astype(np.uint32), q)
    for _ in range(2):                                              # twice: the scratch must come back clean
        assert np.array_equal(ctx.matvec(key, x), want)


@pytest.mark.parametrize("n,m,q,B", [(13, 154, 1427, 70), (50, 867, 124991, 300), (256, 6168, 16777213, 513),
                                     (3, 16, 13, 5), (64, 1000, 2147483629, 40), (31, 129, 251, 33),
                                     # packed remainder block (rows mod 32 = r, r*digits <= 32): r*LA = 32 exactly with
                                     # LA = 2 and LA = 4, r*LA = 30 at LA = 3, tiles of 4 / 2 / 1 row blocks, a lone
                                     # packed block, and r*LA = 33 which must NOT pack
                                     (47, 300, 40009, 70), (39, 200, 2147483629, 33), (41, 500, 8380417, 257),
                                     (100, 640, 8380417, 300), (229, 256, 12289, 64), (7, 99, 1000003, 19),
                                     (42, 130, 8380417, 65)])
def test_matmul_bits_regev_shape(ctx, n, m, q, B):
    """Regev::enc batched (regev.rs:396-414): pk*x + message in the last entry."""
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m)
    st = bit_stride(m)
    xb = co.sample_bits(SEED, 3, 0, B, m, st)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
    key = ctx.key_upload(pk, q)
    out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, msg=ctx.to_dev_u8(msg))
    ref = co.matmul_batch(pk, q, xb, co.X_BITS, B, msg=msg)
    assert np.array_equal(ctx.to_host_u32(out), ref)


@pytest.mark.parametrize("n,m,q,B", [(13, 154, 1427, 70), (50, 867, 124991, 300), (256, 6168, 16777213, 513), (3, 16, 13, 5),
                                     (64, 1000, 2147483629, 40), (31, 129, 251, 33), (47, 300, 40009, 70), (41, 500, 8380417, 257),
                                     (100, 640, 8380417, 300), (7, 99, 1000003, 19), (511, 11776, 8380417, 1000), (1, 1, 7, 1),
                                     (256, 6168, 16777213, 256 * 3 + 93)])
def test_matmul_bits_k_tile_major(ctx, n, m, q, B):
    """The same products from k-tile-major bits (QF_X_BITS_KT): qf_bits_to_kt_dev against a host transposition, the
    product against the row-major one and the oracle; planes wider than the batch, addend + message; a plane too
    short for the batch is refused."""
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m)
    st = bit_stride(m)
    nt = (m + 127) // 128
    xb = co.sample_bits(SEED, 3, 0, B, m, st)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
    add = co.sample_uniform(SEED, 4, 0, B * (n + 1), q).reshape(B, n + 1)
    key = ctx.key_upload(pk, q)
    xd = ctx.to_dev_u8(xb)
    kt = ctx.bits_to_kt(xd, B, m)
    assert tuple(kt.shape) == (nt, B, 16)
    assert np.array_equal(kt.cpu().numpy(), np.ascontiguousarray(xb[:, :nt * 16].reshape(B, nt, 16).transpose(1, 0, 2)))
    ref = co.matmul_batch(pk, q, xb, co.X_BITS, B, msg=msg)
    out = ctx.matmul(key, kt, X_BITS_KT, B, msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), ref)
    assert torch.equal(out, ctx.matmul(key, xd, X_BITS, B, msg=ctx.to_dev_u8(msg)))
    # planes longer than the batch (garbage behind row B-1 must not matter), with an addend
    wide = torch.full((nt, B + 37, 16), 0xA5, dtype=torch.uint8, device=kt.device)
    ctx.bits_to_kt(xd, B, m, out=wide)
    ref2 = co.matmul_batch(pk, q, xb, co.X_BITS, B, addend=add, msg=msg)
    out2 = ctx.matmul(key, wide, X_BITS_KT, B, addend=ctx.to_dev_u32(add), msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out2), ref2)
    if B > 1:
        with pytest.raises(QfError):
            ctx.matmul(key, kt[:, :B - 1], X_BITS_KT, B)


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 64), (5, 65), (70000, 129), (300, 6168), (9, 867), (257, 128), (1000, 11776)])
def test_bit_sampler_k_tile_major(ctx, rows, cols):
    """qf_sample_bits_kt_dev / _keyed: the rows of the row-major sampler (= the oracle's), word (row, k-tile) at plane k-tile."""
    nt = (cols + 127) // 128
    st = bit_stride(cols)
    for seed, ref in ((SEED, co.sample_bits(SEED, 3, 11, rows, cols, st)),
                      ((bytes(range(32)), 5), co.sample_bits((bytes(range(32)), 5), 3, 11, rows, cols, st))):
        kt = ctx.sample_bits_kt(seed, 3, 11, rows, cols)
        assert tuple(kt.shape) == (nt, rows, 16)
        assert np.array_equal(kt.cpu().numpy(), np.ascontiguousarray(ref.reshape(rows, nt, 16).transpose(1, 0, 2)))
    wide = torch.full((nt, rows + 5, 16), 0x5A, dtype=torch.uint8, device=kt.device)
    ctx.sample_bits_kt(SEED, 3, 11, rows, cols, out=wide)
    assert torch.equal(wide[:, :rows], ctx.sample_bits_kt(SEED, 3, 11, rows, cols)) and bool((wide[:, rows:] == 0x5A).all())


@pytest.mark.parametrize("n,q,W,B,digits", [(10, 983, 3, 50, 0), (10, 983, 3, 50, 1), (64, 1073741789, 2000, 70, 3),
                                            (64, 1073741789, 2000, 70, 2), (40, 65521, 30000, 35, 1), (128, 8380417, 100, 64, 0)])
def test_matmul_u32_lpr_shape(ctx, n, q, W, B, digits):
    """LPR::enc batched (lpr.rs:394-430): pk*r + e + message; digits = caller's width promise
    (a broken promise must fall back to full width and stay exact)."""
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * n, q).reshape(n + 1, n)
    r = co.sample_noise(SEED, 3, 0, B * n, q, W).reshape(B, n)
    e = co.sample_noise(SEED, 4, 0, B * (n + 1), q, W).reshape(B, n + 1)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
    key = ctx.key_upload(pk, q)
    out = ctx.matmul(key, ctx.to_dev_u32(r), X_U32, B, addend=ctx.to_dev_u32(e), msg=ctx.to_dev_u8(msg), x_digits=digits)
    ref = co.matmul_batch(pk, q, r, co.X_U32, B, addend=e, msg=msg)
    assert np.array_equal(ctx.to_host_u32(out), ref)


@pytest.mark.parametrize("n,q,W,B,digits", [(10, 983, 3, 50, 1), (64, 1073741789, 2000, 70, 2), (64, 1073741789, 2000, 70, 4), (40, 65521, 30, 35, 1),
                                            (128, 8380417, 100, 64, 3), (1024, 1073741789, 2319, 300, 2), (230, 7764299, 0, 257, 3)])
def test_matmul_prerecoded_planes(ctx, n, q, W, B, digits):
    """Operand kind QF_X_PLANES: qf_recode_planes_dev once, then the product on the planes -- the ciphertexts of the
    QF_X_U32 path and the oracle (LPR::enc shape, lpr.rs:394-430; W = 0: uniform entries at full width); a promise the
    entries break is refused with QF_ERR_RANGE, and so are planes with a digit count the key does not have."""
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * n, q).reshape(n + 1, n)
    r = co.sample_noise(SEED, 3, 0, B * n, q, W).reshape(B, n) if W else co.sample_uniform(SEED, 3, 0, B * n, q).reshape(B, n)
    e = co.sample_uniform(SEED, 4, 0, B * (n + 1), q).reshape(B, n + 1)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
    key = ctx.key_upload(pk, q)
    rd, ed, md = ctx.to_dev_u32(r), ctx.to_dev_u32(e), ctx.to_dev_u8(msg)
    planes = ctx.recode_planes(key, rd, X_U32, B, digits)
    out = ctx.matmul(key, planes, X_PLANES, B, addend=ed, msg=md, x_digits=digits)
    ref = co.matmul_batch(pk, q, r, co.X_U32, B, addend=e, msg=msg)
    assert np.array_equal(ctx.to_host_u32(out), ref)
    assert torch.equal(out, ctx.matmul(key, rd, X_U32, B, addend=ed, msg=md, x_digits=digits))
    assert torch.equal(ctx.matmul(key, planes, X_PLANES, B, x_digits=digits), ctx.matmul(key, rd, X_U32, B))
    if W and digits < 4 and q > 2 ** 24:
        big = rd.clone()
        big[B // 2, n // 3] = q // 2                    # does not fit `digits` signed bytes
        with pytest.raises(QfError) as err:
            ctx.recode_planes(key, big, X_U32, B, digits)
        assert err.value.code == _lib.QF_ERR_RANGE
        assert torch.equal(ctx.matmul(key, planes, X_PLANES, B, addend=ed, msg=md, x_digits=digits), out)   # the context is fine afterwards
    with pytest.raises(QfError):
        ctx.matmul(key, planes, X_PLANES, B, x_digits=0)
    with pytest.raises(QfError):
        ctx.recode_planes(key, rd, X_U32, B, 5)


def test_matmul_full_width_and_i32(ctx):
    """SIS hash of uniform Z_q vectors (sis.rs:46,161-167) and a signed operand."""
    n, m, q, B = 32, 300, 8380417, 90
    A = co.sample_uniform(SEED, 1, 0, n * m, q).reshape(n, m)
    v = co.sample_uniform(SEED, 6, 0, B * m, q).reshape(B, m)
    key = ctx.key_upload(A, q)
    out = ctx.matmul(key, ctx.to_dev_u32(v), X_U32, B)
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, v, co.X_U32, B))
    sg = (co.sample_uniform(SEED, 7, 0, B * m, 2001).astype(np.int64) - 1000).astype(np.int32).reshape(B, m)
    out = ctx.matmul(key, ctx.to_dev_i32(sg), X_I32, B, x_digits=2)
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, sg, co.X_I32, B))


def test_matmul_transposed_key_dual_regev(ctx):
    """DualRegev::enc (dual_regev.rs:385-413): c^t = s^t*pk + e^t, key uploaded transposed."""
    n, m, q, B = 20, 270, 7993, 45
    pk = co.sample_uniform(SEED, 1, 0, n * (m + 1), q).reshape(n, m + 1)
    s = co.sample_uniform(SEED, 2, 0, B * n, q).reshape(B, n)
    e = co.sample_noise(SEED, 4, 0, B * (m + 1), q, 5).reshape(B, m + 1)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
    key = ctx.key_upload(pk, q, transpose=True)
    assert (key.rows, key.cols) == (m + 1, n)
    out = ctx.to_host_u32(ctx.matmul(key, ctx.to_dev_u32(s), X_U32, B, addend=ctx.to_dev_u32(e), msg=ctx.to_dev_u8(msg)))
    ref = co.matmul_batch(np.ascontiguousarray(pk.T), q, s, co.X_U32, B, addend=e, msg=msg)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("n,q,B", [(13, 1427, 100), (256, 16777213, 1000), (1024, 1073741789, 77)])
def test_dot_decode(ctx, n, q, B):
    s = co.sample_uniform(SEED, 2, 0, n, q)
    Cm = co.sample_uniform(SEED, 9, 0, B * (n + 1), q).reshape(B, n + 1)
    w = ((q - s.astype(np.int64)) % q).astype(np.uint32)
    bits = ctx.to_host_u8(ctx.dot_decode(ctx.to_dev_u32(w), q, ctx.to_dev_u32(Cm)))
    assert np.array_equal(bits, co.dot_decode_batch(s, q, Cm))


@pytest.mark.parametrize("n,q,B", [(1024, 1073479681, 37), (16, 2399, 50), (4, 16, 9), (64, 257, 70), (512, 92897729, 6),
                                   (2, 5, 300), (2048, 1073479681, 3), (256, 7681, 20),
                                   # wave-per-polynomial kernel: every degree it serves, lazy (q < 2^30) and
                                   # canonical (q >= 2^30) butterflies, ragged batch sizes, one > grid*waves
                                   (512, 12289, 9), (2048, 12289, 5), (256, 2013265921, 11), (512, 2013265921, 7),
                                   (1024, 2013265921, 13), (2048, 2013265921, 2), (1024, 12289, 1), (1024, 1073479681, 8200),
                                   (4096, 1073479681, 2), (128, 257, 10)])
def test_polymul(ctx, n, q, B):
    a = co.sample_uniform(SEED, 6, 0, B * n, q).reshape(B, n)
    b = co.sample_uniform(SEED, 6, B * n, B * n, q).reshape(B, n)
    ring = ctx.ring(n, q)
    c = ctx.to_host_u32(ring.polymul(ctx.to_dev_u32(a), ctx.to_dev_u32(b)))
    assert np.array_equal(c, co.polymul_batch(a, b, q))


@pytest.mark.parametrize("n,q,W,B", [(1024, 1073479681, 2900, 21), (16, 2399, 4, 64), (64, 257 * 512 + 1, 3, 33), (512, 92897729, 200, 5),
                                     (1024, 2013265921, 2900, 9), (256, 7681, 2, 70), (2048, 1073479681, 100, 3), (512, 12289, 2, 6)])
def test_ringlpr_enc_dec(ctx, n, q, W, B):
    a = co.sample_uniform(SEED, 1, 0, n, q)
    s = co.sample_noise(SEED, 2, 0, n, q, W)
    e = co.sample_noise(SEED, 4, 0, n, q, W)
    bpoly = (co.polymul_batch(a[None], s[None], q)[0].astype(np.int64) + e) % q
    r = co.sample_noise(SEED, 3, 0, B * n, q, W).reshape(B, n)
    e1 = co.sample_noise(SEED, 4, n, B * n, q, W).reshape(B, n)
    e2 = co.sample_noise(SEED, 4, n + B * n, B * n, q, W).reshape(B, n)
    mu = co.sample_bits(SEED, 5, 0, B, n, (n + 7) // 8)
    ring = ctx.ring(n, q)
    pk = ring.key(np.stack([a, bpoly.astype(np.uint32)]))
    sk = ring.key(s[None])
    u, v = ring.ringlpr_enc(pk, ctx.to_dev_u32(r), ctx.to_dev_u32(e1), ctx.to_dev_u32(e2), ctx.to_dev_u8(mu))
    ur, vr = co.ringlpr_enc_batch(a, bpoly, q, r, e1, e2, mu)
    assert np.array_equal(ctx.to_host_u32(u), ur) and np.array_equal(ctx.to_host_u32(v), vr)
    dec = ctx.to_host_u8(ring.ringlpr_dec(sk, u, v))
    assert np.array_equal(dec, co.ringlpr_dec_batch(s, q, ur, vr))


@pytest.mark.parametrize("n,q,B", [(1024, 1073479681, 5), (8, 512, 40), (64, 257, 17), (1024, 2013265921, 3),
                                   (256, 7681, 9), (512, 12289, 6), (2048, 1073479681, 2)])
def test_fdh_ring_verify(ctx, n, q, B):
    k = (q - 1).bit_length()
    mbar = k + 2
    A = co.sample_uniform(SEED, 1, 0, mbar * n, q).reshape(mbar, n)
    A[0] = 0; A[0, 0] = 1
    sigma = (co.sample_uniform(SEED, 7, 0, B * mbar * n, 201).astype(np.int64) - 100).astype(np.int32).reshape(B, mbar, n)
    bound = int((sigma[0].astype(np.int64) ** 2).sum()) + 50
    zero_u = np.zeros((B, n), np.uint32)
    t_ref, _ = co.fdh_ring_verify_batch(A, q, sigma, zero_u, bound)
    u = t_ref.copy()
    u[1, 3] = (u[1, 3] + 1) % q          # wrong target -> reject
    ring = ctx.ring(n, q)
    ak = ring.key(A)
    ok, t = ring.fdh_verify(ak, ctx.to_dev_i32(sigma), ctx.to_dev_u32(u), bound, want_t=True)
    t_ref, ok_ref = co.fdh_ring_verify_batch(A, q, sigma, u, bound)
    assert np.array_equal(ctx.to_host_u32(t), t_ref)
    assert np.array_equal(ctx.to_host_u8(ok), ok_ref)
    assert ok_ref[1] == 0 and ok_ref[0] == 1


@pytest.mark.parametrize("n,q", [(1024, 2147473409), (256, 2013265921), (1024, 1073479681), (64, 257)])
def test_fdh_ring_verify_full_range_sigma(ctx, n, q):
    """sigma is a MatPolyOverZ (gpv_ring.rs:237 reduces it into [0,q)): coefficients over the whole
    int32 range, INT32_MIN / INT32_MAX included, with q just below 2^31 and in (2^30, 2^31) -- the
    wave kernel used to wrap positive coefficients >= 2^32 - off32 (round-1 advisor finding)."""
    B = 3
    k = (q - 1).bit_length()
    mbar = k + 2
    A = co.sample_uniform(SEED, 1, 0, mbar * n, q).reshape(mbar, n)
    raw = co.sample_uniform(SEED, 7, 0, B * mbar * n, 2 ** 31 - 1).astype(np.int64) * 2 - (2 ** 31 - 1)
    sigma = np.clip(raw, -2 ** 31, 2 ** 31 - 1).astype(np.int32).reshape(B, mbar, n)
    sigma[0, 0, :4] = [2 ** 31 - 1, -2 ** 31, 2 ** 31 - 2, -1]
    sigma[1, 1, :3] = [2 ** 32 - 2 * q - 1 if q > 2 ** 30 else 5, q - 1, -(q - 1)]
    sigma[2] = 0
    sigma[2, 2, 5] = 2 ** 31 - 1                          # norm 2^62 - 2^32 + 1: exactly at the bound below
    bound = (2 ** 31 - 1) ** 2
    zero_u = np.zeros((B, n), np.uint32)
    t_ref, _ = co.fdh_ring_verify_batch(A, q, sigma, zero_u, bound)
    ring = ctx.ring(n, q)
    ok, t = ring.fdh_verify(ring.key(A), ctx.to_dev_i32(sigma), ctx.to_dev_u32(t_ref), bound, want_t=True)
    _, ok_ref = co.fdh_ring_verify_batch(A, q, sigma, t_ref, bound)
    assert np.array_equal(ctx.to_host_u32(t), t_ref)
    assert np.array_equal(ctx.to_host_u8(ok), ok_ref)
    assert ok_ref.tolist() == [0, 0, 1]                   # the first two exceed the norm bound


def test_hash_to_mat_zq_sha256_kats(ctx):
    """construction/hash/sha256.rs:104,300-332 known answers, through the device kernel."""
    got = ctx.to_host_u32(ctx.hash_to_mat_zq_sha256(["Hello World!"], 2, 2, 7)).reshape(2, 2)
    assert got.tolist() == [[6, 3], [5, 2]]
    got = ctx.to_host_u32(ctx.hash_to_mat_zq_sha256(["Hello World!"], 2, 2, 256)).reshape(2, 2)
    assert got.tolist() == [[159, 26], [249, 141]]
    got = ctx.to_host_u32(ctx.hash_to_mat_zq_sha256(["qfall"], 2, 2, 16)).reshape(2, 2)
    assert got.tolist() == [[3, 12], [9, 12]]


@pytest.mark.parametrize("rows,cols,q", [(1024, 1, 1073479681), (8, 1, 512), (5, 3, 2147483629), (64, 1, 257)])
def test_hash_to_mat_zq_sha256_vs_oracle(ctx, rows, cols, q):
    """Ragged messages: empty, one block, the 55/56/64-byte padding edges, multi-block, non-ASCII."""
    from oracle import pyoracle as po
    msgs = ["", "a", "x" * 20, "y" * 55, "z" * 56, "w" * 64, "Hello World! " * 40, "gr\u00fc\u00dfe \u2713", "m" * 119, "k" * 120]
    got = ctx.to_host_u32(ctx.hash_to_mat_zq_sha256(msgs, rows, cols, q))
    for b, m in enumerate(msgs):
        ref = np.array(po.hash_to_mat_zq(m, rows, cols, q), dtype=np.uint64).reshape(-1)
        assert np.array_equal(got[b].astype(np.uint64), ref), (b, m[:10])


def test_c_abi_error_codes_and_empty_batches(ctx):
    """Raw C ABI: the codes a binding turns back into the reference's panics (include/qfgpu.h),
    and empty batches, which launch nothing and succeed."""
    import ctypes as C
    from crypto_amd import _lib
    lib, h = ctx.lib, ctx.h
    q, n = 1073479681, 1024
    ring = ctx.ring(n, q)
    rk = ring.key(co.sample_uniform(SEED, 1, 0, n, q)[None])           # one polynomial
    u = ctx.to_dev_u32(np.zeros((2, n), np.uint32))
    mu = ctx.to_dev_u8(np.zeros((2, n // 8), np.uint8))
    # RingLPR::enc needs pk = (a, b): a key with one polynomial is a dimension error
    rc = lib.qf_ringlpr_enc_dev(h, ring.h, rk.h, u.data_ptr(), u.data_ptr(), u.data_ptr(), mu.data_ptr(), 2, u.data_ptr(), u.data_ptr())
    assert rc == _lib.QF_ERR_DIM
    # a ring key of another ring: modulus mismatch
    ring2 = ctx.ring(n, 12289)
    rc = lib.qf_ringlpr_dec_dev(h, ring2.h, rk.h, u.data_ptr(), u.data_ptr(), 2, mu.data_ptr())
    assert rc == _lib.QF_ERR_MODULUS
    # null operands
    assert lib.qf_polymul_dev(h, ring.h, None, u.data_ptr(), 2, u.data_ptr()) == _lib.QF_ERR_ARG
    assert lib.qf_dot_decode_dev(h, None, 4, 17, u.data_ptr(), 1, mu.data_ptr()) == _lib.QF_ERR_ARG
    # empty batches
    assert lib.qf_polymul_dev(h, ring.h, None, None, 0, None) == 0
    assert lib.qf_dot_decode_dev(h, None, 4, 17, None, 0, None) == 0
    assert lib.qf_fdh_ring_verify_dev(h, ring.h, rk.h, None, None, 0, 0, None, None) == 0
    assert lib.qf_hash_to_mat_zq_sha256_dev(h, None, None, 0, n, 1, q, None) == 0
    # bad ring degrees / moduli
    hr = C.c_void_p()
    assert lib.qf_ring_create(h, 1000, q, C.byref(hr)) == _lib.QF_ERR_NOTSUP       # not a power of two
    assert lib.qf_ring_create(h, 1024, 2 ** 31, C.byref(hr)) == _lib.QF_ERR_MODULUS
    assert lib.qf_hash_to_mat_zq_sha256_dev(h, u.data_ptr(), u.data_ptr(), 1, 0, 1, q, u.data_ptr()) == _lib.QF_ERR_DIM
    assert b"" != lib.qf_last_error(h)


@pytest.mark.parametrize("split", ["0", "1"])
def test_dot_decode_noncanonical_words_and_forced_split(ctx, monkeypatch, split):
    """The single-sum fast path must notice operand words at or above 2^bits(q) and redo that group with
    split sums (exact for ANY uint32 words); QF_DECODE_SPLIT=1 runs the split sums throughout."""
    monkeypatch.setenv("QF_DECODE_SPLIT", split)
    n, q, B = 256, 16777213, 333
    s = co.sample_uniform(SEED, 2, 0, n, q)
    w = ((q - s.astype(np.int64)) % q).astype(np.uint32)
    Cm = co.sample_uniform(SEED, 9, 0, B * (n + 1), q).reshape(B, n + 1)
    Cm[7, 3] += np.uint32(200 * q)                     # same residues, words far above q in three groups
    Cm[100, n] += np.uint32(255 * q)
    Cm[332, 200] = np.uint32(2 ** 32 - 1)
    want = co.dot_decode_batch(s, q, (Cm.astype(np.uint64) % q).astype(np.uint32))
    bits = ctx.to_host_u8(ctx.dot_decode(ctx.to_dev_u32(w), q, ctx.to_dev_u32(Cm)))
    assert np.array_equal(bits, want)


@pytest.mark.parametrize("n,q,B", [(1, 2, 9), (2, 3, 64), (4097, 2147483647, 19), (16384, 12289, 8), (255, 2147483629, 1025)])
def test_dot_decode_extreme_moduli_and_lengths(ctx, n, q, B):
    """Smallest and largest supported moduli, n = 1, n beyond one pass of the wave (16,384): the single-sum
    path where (n+1)(2^bits(q)-1)^2 < 2^64, split sums beyond it; plus an empty batch."""
    s = co.sample_uniform(SEED, 2, 0, n, q)
    Cm = co.sample_uniform(SEED, 9, 0, B * (n + 1), q).reshape(B, n + 1)
    w = ((q - s.astype(np.int64)) % q).astype(np.uint32)
    bits = ctx.to_host_u8(ctx.dot_decode(ctx.to_dev_u32(w), q, ctx.to_dev_u32(Cm)))
    assert np.array_equal(bits, co.dot_decode_batch(s, q, Cm))
    assert ctx.dot_decode(ctx.to_dev_u32(w), q, ctx.empty_u32(0, n + 1)).numel() == 0


def test_dot_decode_ragged_and_large_n(ctx):
    """Batch sizes around the 4-ciphertexts-per-wave grouping and n that is not a multiple of 64."""
    for n, q, B in [(50, 124991, 1), (50, 124991, 5), (257, 16777213, 4099), (1000, 1073741789, 7), (3, 7, 130)]:
        s = co.sample_uniform(SEED, 2, 0, n, q)
        Cm = co.sample_uniform(SEED, 9, 0, B * (n + 1), q).reshape(B, n + 1)
        w = ((q - s.astype(np.int64)) % q).astype(np.uint32)
        bits = ctx.to_host_u8(ctx.dot_decode(ctx.to_dev_u32(w), q, ctx.to_dev_u32(Cm)))
        assert np.array_equal(bits, co.dot_decode_batch(s, q, Cm)), (n, q, B)


@pytest.mark.parametrize("env", [{}, {"QF_GEMM_FOLD": "0"}, {"QF_GEMM_EPI2": "1"}])
def test_matmul_wide_operand_paths(ctx, monkeypatch, env):
    """Wide operands, the ways qf_launch_gemm can run them -- digits folded into the inner dimension
    (default), the fused <LA,LB> kernel, the fold with the two-step fp64 recombination forced: same
    bits as the oracle, with addend and message, for 2- and 3-digit operands."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, m, q, B = 40, 4500, 8380417, 50
    A = co.sample_uniform(SEED, 1, 0, n * m, q).reshape(n, m)
    key = ctx.key_upload(A, q)
    v = co.sample_uniform(SEED, 6, 0, B * m, q).reshape(B, m)
    e = co.sample_uniform(SEED, 4, 0, B * n, q).reshape(B, n)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
    out = ctx.matmul(key, ctx.to_dev_u32(v), X_U32, B, addend=ctx.to_dev_u32(e), msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, v, co.X_U32, B, addend=e, msg=msg))
    r = co.sample_noise(SEED, 3, 0, B * m, q, 2000).reshape(B, m)
    out = ctx.matmul(key, ctx.to_dev_u32(r), X_U32, B, x_digits=2)
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, r, co.X_U32, B))


def test_fixed_classical_trapdoor_kat_on_device(ctx):
    """The reference's fixed trapdoor (short_basis_classical.rs:281-303): A*[R;I] = G mod 8 through
    kernel (a) (signed operand, one column per batch item) and kernel (c)."""
    import json, os
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["classical_trapdoor_n2_q8"]
    q, A, G = k["q"], np.array(k["A"], np.uint32), np.array(k["G"], np.uint32)
    T = np.vstack([np.array(k["R"], np.int64), np.eye(6, dtype=np.int64)])
    key = ctx.key_upload(A, q)
    out = ctx.matmul(key, ctx.to_dev_i32(np.ascontiguousarray(T.T)), X_I32, 6)
    assert np.array_equal(ctx.to_host_u32(out), G.T)
    for j in range(6):
        assert ctx.matvec(key, (T[:, j] % q).astype(np.uint32)).tolist() == G[:, j].tolist()


def test_fixed_ring_trapdoor_kat_on_device(ctx):
    """The reference's fixed ring trapdoor (short_basis_ring.rs:357-378, gadget_ring.rs:71-77) in
    Z_16[X]/(X^4+1): a_{2+i} = 2^i - (a_bar*r_i + e_i) through the product kernel, and
    a*[e_i; r_i; I_i] = g_i (is_trapdoor, gadget_ring.rs:189-210) through the fused FDH kernel."""
    import json, os
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["ring_trapdoor_n4_q16"]
    n, q = k["n"], k["q"]
    pad = lambda p: list(p) + [0] * (n - len(p))
    a = np.array([pad(p) for p in k["a"]], np.uint32)
    r = np.array([pad(p) for p in k["r"]], np.int64)
    e = np.array([pad(p) for p in k["e"]], np.int64)
    ring = ctx.ring(n, q)
    abar = np.repeat(a[1:2], 4, axis=0)
    ar = ctx.to_host_u32(ring.polymul(ctx.to_dev_u32(abar), ctx.to_dev_u32((r % q).astype(np.uint32)))).astype(np.int64)
    g = np.zeros((4, n), np.int64); g[:, 0] = [1, 2, 4, 8]
    assert np.array_equal((g - ar - e) % q, a[2:].astype(np.int64))
    sigma = np.zeros((4, 6, n), np.int32)
    sigma[:, 0], sigma[:, 1] = e, r
    for i in range(4):
        sigma[i, 2 + i, 0] = 1
    ok, t = ring.fdh_verify(ring.key(a), ctx.to_dev_i32(sigma), ctx.to_dev_u32(g.astype(np.uint32)), 10 ** 6, want_t=True)
    assert np.array_equal(ctx.to_host_u32(t).astype(np.int64), g % q)
    assert ctx.to_host_u8(ok).tolist() == [1, 1, 1, 1]


def test_reference_gen_sa_r_outputs_on_device(ctx):
    """The two literal outputs the reference's tests hold for this path (working_sa_r, short_basis_ring.rs:
    404-439; working_sa_r_identity, short_basis_classical.rs:339-356): every ring product of the first through
    the product kernel, the two Z_q matrix products of the second through kernels (a) and (c)."""
    import json, os
    from test_oracle_golden import ring_sa_r_products
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
    n, q, cases = ring_sa_r_products(kats)
    a = np.array([c[0] for c in cases], np.uint32)
    b = np.array([c[1] for c in cases], np.uint32)
    want = np.array([c[2] for c in cases], np.uint32)
    got = ctx.to_host_u32(ctx.ring(n, q).polymul(ctx.to_dev_u32(a), ctx.to_dev_u32(b)))
    assert np.array_equal(got, want)
    k = kats["classical_trapdoor_n2_q8"]
    q, A, G = k["q"], np.array(k["A"], np.uint32), np.array(k["G"], np.uint32)
    M = np.array(k["sa_r_identity"], np.int64)
    W = M[7:, 6:]
    E = -np.vstack([np.eye(7, dtype=np.int64), np.zeros((6, 7), np.int64)])
    kG, kA = ctx.key_upload(G, q), ctx.key_upload(A, q)
    lhs = ctx.to_host_u32(ctx.matmul(kG, ctx.to_dev_i32(np.ascontiguousarray(W.T)), X_I32, 7))
    rhs = ctx.to_host_u32(ctx.matmul(kA, ctx.to_dev_i32(np.ascontiguousarray(E.T)), X_I32, 7))
    assert np.array_equal(lhs, rhs)
    assert np.array_equal(rhs.T, (-A[:, :7].astype(np.int64)) % q)
    for j in range(7):
        assert ctx.matvec(kG, (W[:, j] % q).astype(np.uint32)).tolist() == rhs[j].tolist()
        assert ctx.matvec(kA, (E[:, j] % q).astype(np.uint32)).tolist() == rhs[j].tolist()


@pytest.mark.parametrize("bits", [False, True])
def test_matmul_chunked_raster_order(ctx, monkeypatch, bits):
    """qf_gemm_raster: chunks of message tiles run through every column tile; 19 message tiles in
    chunks of 8 (a short last chunk), plain and packed-remainder kernels."""
    monkeypatch.setenv("QF_GEMM_MCHUNK_BITS" if bits else "QF_GEMM_MCHUNK", "8")
    B = 256 * 18 + 77
    if bits:
        n, m, q = 256, 200, 16777213
        pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m)
        xb = co.sample_bits(SEED, 3, 0, B, m, bit_stride(m))
        msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
        out = ctx.matmul(ctx.key_upload(pk, q), ctx.to_dev_u8(xb), X_BITS, B, msg=ctx.to_dev_u8(msg))
        assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(pk, q, xb, co.X_BITS, B, msg=msg))
    else:
        n, q = 128, 1073741789
        pk = co.sample_uniform(SEED, 1, 0, (n + 1) * n, q).reshape(n + 1, n)
        r = co.sample_noise(SEED, 3, 0, B * n, q, 3000).reshape(B, n)
        e = co.sample_noise(SEED, 4, 0, B * (n + 1), q, 3000).reshape(B, n + 1)
        out = ctx.matmul(ctx.key_upload(pk, q), ctx.to_dev_u32(r), X_U32, B, addend=ctx.to_dev_u32(e), x_digits=2)
        assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(pk, q, r, co.X_U32, B, addend=e))


@pytest.mark.parametrize("q,n,s", [(983, 10, 983 * 0.0072), (1073741789, 1024, 1073741789 * 6.25e-6), (16777213, 256, 3000.5), (7, 2, 40.0)])
def test_discrete_gauss_sampler(ctx, q, n, s):
    """qf_sample_discrete_gauss_dev: bit-identical to the oracle's inversion sampler, and distributed as
    D_{Z,s}: support |x| <= ceil(s log2 n), mean 0, variance s^2/(2 pi), per-value frequencies (chi-square
    against exp(-pi x^2/s^2)).  Parity with the reference is distributional only (its RNG is unseeded)."""
    N = 400000
    d = ctx.to_host_u32(ctx.sample_discrete_gauss(SEED, 4, 123, N, q, n, s))
    assert np.array_equal(d, co.sample_discrete_gauss(SEED, 4, 123, N, q, n, s))
    if q < 4 * s * np.log2(n):
        return                                   # wraps around mod q: only the parity check applies
    x = d.astype(np.int64)
    x[x > q // 2] -= q
    T = int(np.ceil(s * np.log2(n)))
    assert np.abs(x).max() <= T
    sigma = s / np.sqrt(2 * np.pi)
    assert abs(x.mean()) < 5 * sigma / np.sqrt(N)
    assert abs(x.var() / sigma ** 2 - 1) < 0.02
    # chi-square over 40 equiprobable-ish bins of width sigma/4 around 0
    edges = np.arange(-20, 21) * max(1, int(round(sigma / 4)))
    support = np.arange(-T, T + 1)
    pmf = np.exp(-np.pi * support.astype(np.float64) ** 2 / s ** 2)
    pmf /= pmf.sum()
    exp = np.array([pmf[(support >= a) & (support < b)].sum() for a, b in zip(edges[:-1], edges[1:])]) * N
    obs = np.histogram(x, bins=edges)[0]
    keep = exp > 50
    chi2 = (((obs - exp) ** 2) / exp)[keep].sum()
    assert chi2 < keep.sum() + 6 * np.sqrt(2 * keep.sum()), (chi2, keep.sum())


@pytest.mark.parametrize("rows,cols,q,B,P", [
    (70, 230, 7764299, 300, 0),         # 3 row blocks -> 2 column tiles, the second half empty; one unit per tile
    (129, 256, 7764299, 2100, 3),       # 5 row blocks, batch ends inside a block, 3 message splits
    (5314, 230, 7764299, 700, 2),       # DualRegev secure128 key (167 row blocks -> 84 tiles)
    (100, 128, 1073741789, 515, 0),     # 4 key digits x 4 operand digits: K' = 512
    (97, 500, 40961, 1030, 5),          # 2 x 2 digits, K' = 1024
    (64, 2000, 251, 257, 0),            # 1 digit, K' = 2048 (the LDS limit)
    (33, 40, 7764299, 1, 0),            # one message
])
@pytest.mark.parametrize("with_addend", [False, True])
def test_matmul_short_k_sweep(ctx, monkeypatch, rows, cols, q, B, P, with_addend):
    """qf_gemm_shortk_kernel (key tile resident in LDS, waves sweep message blocks on their own): forced on for
    small shapes, against the oracle -- padded column tiles, ragged batches, message splits, addend + message
    term on the last row, the two-step fp64 recombination, and the same product through the tiled kernel."""
    monkeypatch.setenv("QF_GEMM_SHORTK", "1")
    if P:
        monkeypatch.setenv("QF_GEMM_SHORTK_P", str(P))
    A = co.sample_uniform(SEED, 1, rows, rows * cols, q).reshape(rows, cols)
    X = co.sample_uniform(SEED, 3, 11, B * cols, q).reshape(B, cols)
    e = co.sample_uniform(SEED, 4, 5, B * rows, q).reshape(B, rows) if with_addend else None
    msg = co.sample_uniform(SEED, 5, 0, B, 2).astype(np.uint8) if with_addend else None
    key = ctx.key_upload(A, q)
    ref = co.matmul_batch(A, q, X, co.X_U32, B, addend=e, msg=msg, mt=True)
    dX = ctx.to_dev_u32(X)
    de = ctx.to_dev_u32(e) if with_addend else None
    dm = ctx.to_dev_u8(msg) if with_addend else None
    guard = ctx.empty_u32(B + 2, rows)
    guard.fill_(-1)
    out = ctx.matmul(key, dX, X_U32, B, addend=de, msg=dm, out=guard[1:B + 1])
    assert np.array_equal(ctx.to_host_u32(out), ref)
    g = ctx.to_host_u32(guard)
    assert (g[0] == 0xFFFFFFFF).all() and (g[B + 1] == 0xFFFFFFFF).all()      # nothing written around the batch
    monkeypatch.setenv("QF_GEMM_EPI2", "1")
    assert np.array_equal(ctx.to_host_u32(ctx.matmul(key, dX, X_U32, B, addend=de, msg=dm)), ref)
    monkeypatch.delenv("QF_GEMM_EPI2")
    monkeypatch.setenv("QF_GEMM_SHORTK", "0")
    assert np.array_equal(ctx.to_host_u32(ctx.matmul(key, dX, X_U32, B, addend=de, msg=dm)), ref)


def test_keyed_samplers_match_the_oracle(ctx):
    """qf_sample_*_keyed_dev (ChaCha20 under a 256-bit key: what an object constructed without a seed draws
    from) against the oracle's restatement, which RFC 7539's block vector pins (tests/test_oracle_golden.py):
    uniform, bit rows (16-byte and 4-byte stride paths), Irwin-Hall noise, D_{Z,s}; counters beyond 2^32."""
    key = bytes((11 * i + 5) & 255 for i in range(32))
    sd = (key, 0x0123456789ABCDEF)
    q = 1073741789
    for first in (0, 77, (1 << 33) + 5):
        assert np.array_equal(ctx.to_host_u32(ctx.sample_uniform(sd, 1, first, 5000, q)),
                              co.sample_uniform_keyed(key, sd[1], 1, first, 5000, q))
    assert np.array_equal(ctx.to_host_u32(ctx.sample_noise(sd, 4, 9, 3000, q, 3333)), co.sample_noise_keyed(key, sd[1], 4, 9, 3000, q, 3333))
    assert np.array_equal(ctx.to_host_u32(ctx.sample_discrete_gauss(sd, 2, 5, 4000, 983, 10, 7.08)),
                          co.sample_discrete_gauss_keyed(key, sd[1], 2, 5, 4000, 983, 10, 7.08))
    for cols, stride in ((867, None), (6168, None), (77, 12), (64, 8)):
        st = stride or bit_stride(cols)
        assert np.array_equal(ctx.to_host_u8(ctx.sample_bits(sd, 3, 3, 9, cols, stride=st)), co.sample_bits_keyed(key, sd[1], 3, 3, 9, cols, st))
    # another key, another nonce, another stream: different draws
    a = ctx.to_host_u32(ctx.sample_uniform(sd, 1, 0, 64, q))
    assert not np.array_equal(a, ctx.to_host_u32(ctx.sample_uniform((bytes(32), sd[1]), 1, 0, 64, q)))
    assert not np.array_equal(a, ctx.to_host_u32(ctx.sample_uniform((key, sd[1] + 1), 1, 0, 64, q)))
    assert not np.array_equal(a, ctx.to_host_u32(ctx.sample_uniform(sd, 2, 0, 64, q)))
    with pytest.raises(QfError):
        ctx.sample_uniform((b"short", 0), 1, 0, 4, q)


def test_replay_fixtures_on_device(ctx):
    """The fixtures meant for replay through the real reference (tests/golden/replay):
    the device reproduces every expected value."""
    import os
    from crypto_amd.math import MatZ, MatZq, parse_poly
    path = os.path.join(os.path.dirname(__file__), "golden", "replay", "fixtures.txt")
    f = dict(l.split("=", 1) for l in open(path).read().splitlines() if "=" in l)
    u32 = lambda m: np.ascontiguousarray(m.entries.astype(np.uint32))
    pk, x, c = MatZq.from_str(f["regev_pk"]), MatZ.from_str(f["regev_x"]), MatZq.from_str(f["regev_c"])
    xb = pack_bits(x.entries[:, 0][None, :].astype(np.uint8), bit_stride(pk.entries.shape[1]))
    out = ctx.matmul(ctx.key_upload(u32(pk), pk.q), ctx.to_dev_u8(xb), X_BITS, 1, msg=ctx.to_dev_u8(np.array([1], np.uint8)))
    assert ctx.to_host_u32(out)[0].tolist() == [int(v) for v in c.entries[:, 0]]
    A, v, d = MatZq.from_str(f["sis_key"]), MatZq.from_str(f["sis_value"]), MatZq.from_str(f["sis_digest"])
    assert ctx.matvec(ctx.key_upload(u32(A), A.q), u32(v)[:, 0].copy()).tolist() == [int(t) for t in d.entries[:, 0]]
    pk, r, e, c = (MatZq.from_str(f[k]) for k in ("lpr_pk", "lpr_r", "lpr_e", "lpr_c"))
    out = ctx.matmul(ctx.key_upload(u32(pk), pk.q), ctx.to_dev_u32(u32(r).T.copy()), X_U32, 1, addend=ctx.to_dev_u32(u32(e).T.copy()),
                     msg=ctx.to_dev_u8(np.array([1], np.uint8)))
    assert ctx.to_host_u32(out)[0].tolist() == [int(t) for t in c.entries[:, 0]]
    n, q = int(f["ring_n"]), int(f["ring_q"])
    pad = lambda p: np.array([p + [0] * (n - len(p))], np.uint32)
    a, b, cc = (pad(parse_poly(f[k])) for k in ("ring_a", "ring_b", "ring_c"))
    assert np.array_equal(ctx.to_host_u32(ctx.ring(n, q).polymul(ctx.to_dev_u32(a), ctx.to_dev_u32(b))), cc)


def test_replay_fixtures_round3_on_device(ctx):
    """The round-3 fixtures (Regev::dec incl. the tie of the decision rule, DualRegev enc/dec, RingLPR enc/dec incl. tie
    coefficients, PSFGPVRing::f_a at n = 8 / q = 512, hash_to_mat_zq_sha256): the device reproduces every expected value
    that tests/golden/replay/replay_test.rs asserts of the real reference."""
    import os
    from crypto_amd.math import MatPolyOverZ, MatZ, MatZq, parse_poly
    path = os.path.join(os.path.dirname(__file__), "golden", "replay", "fixtures.txt")
    f = dict(l.split("=", 1) for l in open(path).read().splitlines() if "=" in l)
    u32 = lambda m: np.ascontiguousarray(m.entries.astype(np.uint32))
    # Regev::dec: w = -s, the six ciphertexts as one batch
    sk = MatZq.from_str(f["regev_dec_sk"])
    q = sk.q
    Cm = np.stack([u32(MatZq.from_str(f[f"regev_dec_c{k}"]))[:, 0] for k in range(int(f["regev_dec_count"]))])
    w = ((q - u32(sk)[:, 0].astype(np.int64)) % q).astype(np.uint32)
    assert ctx.to_host_u8(ctx.dot_decode(ctx.to_dev_u32(w), q, ctx.to_dev_u32(Cm))).tolist() == [int(b) for b in f["regev_dec_bits"].split()]
    # DualRegev: transposed key, full-width operand s, then c^t [-x ; 1]
    pk, s, e = (MatZq.from_str(f[k]) for k in ("dualregev_pk", "dualregev_s", "dualregev_e"))
    x = MatZ.from_str(f["dualregev_sk"]).entries[:, 0].astype(np.int64)
    q = pk.q
    key = ctx.key_upload(u32(pk), q, transpose=True)
    wx = ctx.to_dev_u32(((q - x) % q).astype(np.uint32))
    for msg in (0, 1):
        c = ctx.matmul(key, ctx.to_dev_u32(u32(s).T.copy()), X_U32, 1, addend=ctx.to_dev_u32(u32(e).T.copy()), msg=ctx.to_dev_u8(np.array([msg], np.uint8)))
        assert ctx.to_host_u32(c)[0].tolist() == [int(t) for t in MatZq.from_str(f[f"dualregev_c{msg}"]).entries[:, 0]]
        assert ctx.to_host_u8(ctx.dot_decode(wx, q, c)).tolist() == [msg]
    # RingLPR (q = 2399: not NTT-friendly, the exact schoolbook kernel -- the reference's own default set)
    n, q = int(f["ringlpr_n"]), int(f["ringlpr_q"])
    P = lambda k: np.array(parse_poly(f[k]) + [0] * (n - len(parse_poly(f[k]))), np.uint32)
    ring = ctx.ring(n, q)
    pkr = ring.key(np.stack([P("ringlpr_a"), P("ringlpr_b")]))
    mu = np.packbits((P("ringlpr_mu_q_half") != 0).astype(np.uint8)[None, :], axis=1, bitorder="little")
    d = lambda k: ctx.to_dev_u32(P(k)[None])
    u, v = ring.ringlpr_enc(pkr, d("ringlpr_r"), d("ringlpr_e1"), d("ringlpr_e2"), ctx.to_dev_u8(mu))
    assert np.array_equal(ctx.to_host_u32(u)[0], P("ringlpr_u")) and np.array_equal(ctx.to_host_u32(v)[0], P("ringlpr_v"))
    skr = ring.key(P("ringlpr_sk")[None])
    as_int = lambda t: int.from_bytes(bytes(ctx.to_host_u8(t)[0]), "little")
    assert as_int(ring.ringlpr_dec(skr, u, v)) == int(f["ringlpr_message"])
    assert as_int(ring.ringlpr_dec(skr, ctx.to_dev_u32(np.zeros((1, n), np.uint32)), d("ringlpr_tie_v"))) == int(f["ringlpr_tie_message"])
    # f_a inside the fused verification kernel (t returned, compared with u = t: accepted)
    n, q = int(f["fa_n"]), int(f["fa_q"])
    A = MatPolyOverZ.from_str(f["fa_a"], n).entries.astype(np.uint32)
    sig = MatPolyOverZ.from_str(f["fa_sigma"], n).entries.astype(np.int32)
    t_want = MatPolyOverZ.from_str(f["fa_t"], n).entries.astype(np.uint32)
    ring = ctx.ring(n, q)
    ok, t = ring.fdh_verify(ring.key(A), ctx.to_dev_i32(sig[None]), ctx.to_dev_u32(t_want), 100 * 100 * n * 11, want_t=True)
    assert np.array_equal(ctx.to_host_u32(t), t_want) and ctx.to_host_u8(ok).tolist() == [1]
    # hash_to_mat_zq_sha256 on the device
    for key_, rows, cols, qq in (("hash_8x1_512", 8, 1, 512), ("hash_3x2_big", 3, 2, 1073479681)):
        h = ctx.to_host_u32(ctx.hash_to_mat_zq_sha256([f["hash_msg"]], rows, cols, qq)).reshape(rows, cols)
        assert h.tolist() == MatZq.from_str(f[key_]).entries.astype(np.int64).tolist()


@pytest.mark.parametrize("rows", [65, 74, 130, 194, 257, 266])
@pytest.mark.parametrize("q", [251, 65521, 16777213])
def test_matmul_bits_packed_remainder_as_extra_fragment(ctx, monkeypatch, rows, q):
    """Keys whose FULL row blocks split evenly into tiles of 2..4: the packed remainder block rides as an extra
    fragment of the last column tile (qf_gemm_packedx_kernel; Regev n = 256 is the 4 + 4 + 1 case).  Several
    k-tiles, a partial last message tile, addend and message term; QF_GEMM_NOPACKX=1 must give the same."""
    cols, B = 1300, 300
    A = co.sample_uniform(SEED, 1, rows, rows * cols, q).reshape(rows, cols)
    xb = co.sample_bits(SEED, 3, 0, B, cols, bit_stride(cols))
    E = co.sample_uniform(SEED, 4, 0, B * rows, q).reshape(B, rows)
    msg = co.sample_uniform(SEED, 5, 0, B, 2).astype(np.uint8)
    key = ctx.key_upload(A, q)
    want = co.matmul_batch(A, q, xb, co.X_BITS, B, addend=E, msg=msg)
    out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, addend=ctx.to_dev_u32(E), msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), want)
    monkeypatch.setenv("QF_GEMM_NOPACKX", "1")
    out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, addend=ctx.to_dev_u32(E), msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), want)


@pytest.mark.parametrize("rows,cols,q,B", [(33, 31, 12289, 257), (257, 17, 16777213, 300), (4, 1012, 65521, 40), (296, 253, 16777213, 65)])
def test_matmul_bits_with_addend_on_packed_remainder(ctx, rows, cols, q, B):
    """Bit operand + addend (+ message) on keys with a packed remainder block: the remainder rows take the
    addend too (found by tools/soak_gemm.py: the packed epilogue used to drop it)."""
    A = co.sample_uniform(SEED, 1, 0, rows * cols, q).reshape(rows, cols)
    xb = co.sample_bits(SEED, 3, 0, B, cols, bit_stride(cols))
    E = co.sample_uniform(SEED, 4, 0, B * rows, q).reshape(B, rows)
    msg = co.sample_uniform(SEED, 5, 0, B, 2).astype(np.uint8)
    key = ctx.key_upload(A, q)
    out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, addend=ctx.to_dev_u32(E), msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, xb, co.X_BITS, B, addend=E, msg=msg))
    out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, addend=ctx.to_dev_u32(E))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, xb, co.X_BITS, B, addend=E))


@pytest.mark.parametrize("rows,cols,q,B", [(40, 100, 2147483647, 45), (33, 260, 2147483629, 33), (70, 64, 3, 65), (5, 40, 2, 31),
                                           (96, 128, 2147483647, 64), (64, 96, 1073741827, 32)])
def test_matmul_epilogue_extreme_moduli(ctx, rows, cols, q, B):
    """The fp64 epilogue of the one-digit kernels (qf_gemm.h qf_epi_f64) at the ends of the modulus range: the largest
    moduli the library takes (2^31 - 1 and the prime below it), q just above 2^30, q = 2 and 3; bit operands with message,
    digit operands with addend and message."""
    A = co.sample_uniform(SEED, 1, 0, rows * cols, q).reshape(rows, cols)
    key = ctx.key_upload(A, q)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2)).astype(np.uint8)
    xb = co.sample_bits(SEED, 3, 0, B, cols, bit_stride(cols))
    out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, xb, co.X_BITS, B, msg=msg))
    v = co.sample_uniform(SEED, 6, 0, B * cols, q).reshape(B, cols)
    e = co.sample_uniform(SEED, 4, 0, B * rows, q).reshape(B, rows)
    out = ctx.matmul(key, ctx.to_dev_u32(v), X_U32, B, addend=ctx.to_dev_u32(e), msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, v, co.X_U32, B, addend=e, msg=msg))


def test_matmul_epilogue_two_step_recombination(ctx):
    """A 4-digit key times a 4-digit operand over 8,200 columns: 257 * 65536 * |acc| passes 2^52, so the launcher
    sets QfGemmArgs::epi2 and the upper digit pair is reduced before the recombination (no override here)."""
    rows, cols, q, B = 34, 8200, 2147483647, 20
    A = co.sample_uniform(SEED, 1, 0, rows * cols, q).reshape(rows, cols)
    key = ctx.key_upload(A, q)
    v = co.sample_uniform(SEED, 6, 0, B * cols, q).reshape(B, cols)
    e = co.sample_uniform(SEED, 4, 0, B * rows, q).reshape(B, rows)
    out = ctx.matmul(key, ctx.to_dev_u32(v), X_U32, B, addend=ctx.to_dev_u32(e))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, v, co.X_U32, B, addend=e))


@pytest.mark.parametrize("B", [1, 31, 33, 257])
def test_matmul_never_writes_beyond_the_batch(ctx, B):
    """The epilogue's stores go through a buffer descriptor whose record count is the batch bound: rows of the last,
    partial message block beyond B must stay untouched (guard rows behind the output), with bit and digit operands."""
    rows, cols, q = 45, 200, 8380417
    A = co.sample_uniform(SEED, 1, 0, rows * cols, q).reshape(rows, cols)
    key = ctx.key_upload(A, q)
    e = co.sample_uniform(SEED, 4, 0, B * rows, q).reshape(B, rows)
    xb = co.sample_bits(SEED, 3, 0, B, cols, bit_stride(cols))
    v = co.sample_uniform(SEED, 6, 0, B * cols, q).reshape(B, cols)
    for X, kind, ckind, xh in ((ctx.to_dev_u8(xb), X_BITS, co.X_BITS, xb), (ctx.to_dev_u32(v), X_U32, co.X_U32, v)):
        big = torch.full((B + 40, rows), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        ctx.matmul(key, X, kind, B, addend=ctx.to_dev_u32(e), out=big[:B])
        host = ctx.to_host_u32(big)
        assert np.array_equal(host[:B], co.matmul_batch(A, q, xh, ckind, B, addend=e))
        assert (host[B:] == 0x5A5A5A5A).all()


@pytest.mark.parametrize("W,digits", [(100, 1), (30000, 1), (30000, 2)])
def test_matmul_in_place_accumulation_with_digit_promise(ctx, W, digits):
    """addend == out (include/qfgpu.h): the product over a promised digit count is launched before the promise is
    known to hold, except in place, where a broken promise (W = 30000 does not fit one digit) would have destroyed
    the addend before the full-width redo."""
    n, q, B = 40, 65521, 70
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * n, q).reshape(n + 1, n)
    r = co.sample_noise(SEED, 3, 0, B * n, q, W).reshape(B, n)
    e = co.sample_uniform(SEED, 4, 0, B * (n + 1), q).reshape(B, n + 1)
    key = ctx.key_upload(pk, q)
    buf = ctx.to_dev_u32(e)
    ctx.matmul(key, ctx.to_dev_u32(r), X_U32, B, addend=buf, x_digits=digits, out=buf)
    assert np.array_equal(ctx.to_host_u32(buf), co.matmul_batch(pk, q, r, co.X_U32, B, addend=e))
