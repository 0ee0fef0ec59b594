"""Seeded random-shape differential tests: the C ABI kernels against the CPU oracle on shapes the
hand-picked cases do not hit (every residue of rows mod 32 against the packed-remainder rule,
K around multiples of 128, moduli of every digit count, ragged batches)."""
import numpy as np
import pytest

from oracle import coracle as co
from crypto_amd.device import X_BITS, X_U32, bit_stride

pytestmark = pytest.mark.gpu
SEED = 0x5146414C4C
PRIMES = [7, 251, 257, 12289, 65521, 65537, 8380417, 16777213, 16777259, 1073741789, 2147483629]


def _cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n_cases):
        q = int(rng.choice(PRIMES))
        rows = int(rng.integers(1, 140))
        cols = int(rng.choice([rng.integers(1, 40), rng.integers(100, 140), rng.integers(250, 262), rng.integers(500, 530)]))
        B = int(rng.choice([1, 2, 31, 32, 33, 255, 256, 257, 300]))
        yield rows, cols, q, B


@pytest.mark.parametrize("rows,cols,q,B", list(_cases(36, 1)))
def test_random_bits_matmul(ctx, rows, cols, q, B):
    A = co.sample_uniform(SEED, 1, 0, rows * cols, q).reshape(rows, cols)
    st = bit_stride(cols)
    xb = co.sample_bits(SEED, 3, 0, B, cols, st)
    msg = co.sample_uniform(SEED, 5, 0, B, 2).astype(np.uint8)
    key = ctx.key_upload(A, q)
    out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, msg=ctx.to_dev_u8(msg))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, xb, co.X_BITS, B, msg=msg))


@pytest.mark.parametrize("rows,cols,q,B", list(_cases(24, 2)))
def test_random_u32_matmul(ctx, rows, cols, q, B):
    A = co.sample_uniform(SEED, 1, 0, rows * cols, q).reshape(rows, cols)
    X = co.sample_uniform(SEED, 6, 0, B * cols, q).reshape(B, cols)
    E = co.sample_uniform(SEED, 4, 0, B * rows, q).reshape(B, rows)
    key = ctx.key_upload(A, q)
    out = ctx.matmul(key, ctx.to_dev_u32(X), X_U32, B, addend=ctx.to_dev_u32(E))
    assert np.array_equal(ctx.to_host_u32(out), co.matmul_batch(A, q, X, co.X_U32, B, addend=E))
    x1 = co.sample_uniform(SEED, 7, 0, cols, q)
    assert np.array_equal(ctx.matvec(key, x1), co.matvec(A, x1, q))


@pytest.mark.parametrize("n,q,B", [(256, 12289, 3), (256, 2013265921, 5), (512, 12289, 4), (1024, 12289, 2),
                                   (2048, 2013265921, 1), (128, 257, 9), (32, 193, 33), (1024, 1073479681, 4099)])
def test_random_ring_enc_dec_round_trip(ctx, n, q, B):
    """encode -> decode returns every message bit (noise far inside q/4), and u, v match the oracle."""
    a = co.sample_uniform(SEED, 1, 0, n, q)
    s = co.sample_noise(SEED, 2, 0, n, q, 1)
    e = co.sample_noise(SEED, 4, 0, n, q, 1)
    bpoly = ((co.polymul_batch(a[None], s[None], q)[0].astype(np.int64) + e) % q).astype(np.uint32)
    r = co.sample_noise(SEED, 3, 0, B * n, q, 1).reshape(B, n)
    e1 = co.sample_noise(SEED, 4, n, B * n, q, 1).reshape(B, n)
    e2 = co.sample_noise(SEED, 4, n + B * n, B * n, q, 1).reshape(B, n)
    mu = co.sample_bits(SEED, 5, 0, B, n, (n + 7) // 8)
    ring = ctx.ring(n, q)
    pk, sk = ring.key(np.stack([a, bpoly])), ring.key(s[None])
    u, v = ring.ringlpr_enc(pk, ctx.to_dev_u32(r), ctx.to_dev_u32(e1), ctx.to_dev_u32(e2), ctx.to_dev_u8(mu))
    if B <= 64:
        ur, vr = co.ringlpr_enc_batch(a, bpoly, q, r, e1, e2, mu)
        assert np.array_equal(ctx.to_host_u32(u), ur) and np.array_equal(ctx.to_host_u32(v), vr)
    dec = ctx.to_host_u8(ring.ringlpr_dec(sk, u, v))
    if q > 64 * n:   # 4-term noise of width 1: |noise| <= 4 + 8n*... stays below q/4 only for large q
        assert np.array_equal(dec, mu)
