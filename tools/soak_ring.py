"""Diagnostic soak: random degrees / moduli / batch sizes for the ring kernels (product, RingLPR enc/dec, fused FDH
verify), the batched decode, mat-vec / vec-mat and the SHA-256 hash-to-Z_q against the C oracle.
usage: python tools/soak_ring.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from crypto_amd.device import Context

QS = [5, 16, 257, 512, 2399, 7681, 12289, 65537, 92897729, 1073479681, 2013265921, 2147483629, 8380417]
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
ctx = Context(0)
SEED = 77
bad = 0
t0 = time.time()
for c in range(ncases):
    kind = int(rng.integers(0, 7))
    q = int(rng.choice(QS))
    ok = True
    desc = ""
    if kind <= 3:
        n = 1 << int(rng.integers(1, 12))
        B = int(rng.choice([1, 2, 3, 5, 17, 64, 129, 1000])) if n <= 1024 else int(rng.choice([1, 2, 3, 9]))
        ring = ctx.ring(n, q)
        desc = "n %4d q %10d B %4d" % (n, q, B)
        if kind == 0:
            a = co.sample_uniform(SEED, 6, c, B * n, q).reshape(B, n)
            b = co.sample_uniform(SEED, 6, c + B * n, B * n, q).reshape(B, n)
            ok = np.array_equal(ctx.to_host_u32(ring.polymul(ctx.to_dev_u32(a), ctx.to_dev_u32(b))), co.polymul_batch(a, b, q))
        elif kind in (1, 2):
            B = max(1, min(B, int(1e9 // (3 * n * n))))
            W = int(rng.choice([1, 3, 50])) if q > 1000 else 1
            a = co.sample_uniform(SEED, 1, c, n, q)
            s = co.sample_noise(SEED, 2, c, n, q, W)
            bp = co.sample_uniform(SEED, 1, c + n, n, q)
            r = co.sample_noise(SEED, 3, c, B * n, q, W).reshape(B, n)
            e1 = co.sample_uniform(SEED, 4, c, B * n, q).reshape(B, n)
            e2 = co.sample_noise(SEED, 4, c + B * n, B * n, q, W).reshape(B, n)
            mu = co.sample_bits(SEED, 5, c, B, n, max(4, (n + 7) // 8))
            if n < 32:
                continue          # message rows shorter than one 32-bit word: covered by the unit tests' layouts only
            pk = ring.key(np.stack([a, bp]))
            u, v = ring.ringlpr_enc(pk, ctx.to_dev_u32(r), ctx.to_dev_u32(e1), ctx.to_dev_u32(e2), ctx.to_dev_u8(mu))
            ur, vr = co.ringlpr_enc_batch(a, bp, q, r, e1, e2, mu)
            ok = np.array_equal(ctx.to_host_u32(u), ur) and np.array_equal(ctx.to_host_u32(v), vr)
            dec = ctx.to_host_u8(ring.ringlpr_dec(ring.key(s[None]), u, v))
            ok = ok and np.array_equal(dec, co.ringlpr_dec_batch(s, q, ur, vr))
        else:
            mbar = int(rng.integers(1, 34))
            B = max(1, min(B, 64, int(3e8 // (mbar * n * n))))
            A = co.sample_uniform(SEED, 1, c, mbar * n, q).reshape(mbar, n)
            amp = int(rng.choice([1, 100, 2 ** 20, 2 ** 31 - 1]))
            sigma = rng.integers(-amp, amp + 1, (B, mbar, n)).astype(np.int32)
            zero = np.zeros((B, n), np.uint32)
            bound = int(rng.choice([0, 10 ** 6, 2 ** 40, 2 ** 63, 2 ** 64 - 1]))
            t_ref, _ = co.fdh_ring_verify_batch(A, q, sigma, zero, bound)
            u = t_ref.copy()
            flip = rng.integers(0, 2, B).astype(bool)
            u[flip, int(rng.integers(0, n))] ^= 1
            u %= q
            okd, t = ring.fdh_verify(ring.key(A), ctx.to_dev_i32(sigma), ctx.to_dev_u32(u), bound, want_t=True)
            t_ref, ok_ref = co.fdh_ring_verify_batch(A, q, sigma, u, bound)
            ok = np.array_equal(ctx.to_host_u32(t), t_ref) and np.array_equal(ctx.to_host_u8(okd), ok_ref)
            desc += " mbar %2d amp %d bound %d" % (mbar, amp, bound)
    elif kind == 4:
        n = int(rng.choice([1, 2, 13, 63, 64, 65, 256, 257, 1024, 3000]))
        B = int(rng.choice([1, 7, 32, 33, 255, 4097]))
        s = co.sample_uniform(SEED, 2, c, n, q)
        Cm = co.sample_uniform(SEED, 9, c, B * (n + 1), q).reshape(B, n + 1)
        w = ((q - s.astype(np.int64)) % q).astype(np.uint32)
        ok = np.array_equal(ctx.to_host_u8(ctx.dot_decode(ctx.to_dev_u32(w), q, ctx.to_dev_u32(Cm))), co.dot_decode_batch(s, q, Cm))
        desc = "decode n %4d q %10d B %4d" % (n, q, B)
    elif kind == 5:
        rows, cols = int(rng.integers(1, 700)), int(rng.choice([rng.integers(1, 50), rng.integers(500, 3000), 4 * int(rng.integers(100, 3000))]))
        A = co.sample_uniform(SEED, 1, c, rows * cols, q).reshape(rows, cols)
        key = ctx.key_upload(A, q)
        x = co.sample_uniform(SEED, 6, c, cols, q)
        s = co.sample_uniform(SEED, 2, c, rows, q)
        ok = np.array_equal(ctx.matvec(key, x), co.matvec(A, x, q)) and np.array_equal(ctx.vecmat(key, s), co.vecmat(s, A, q))
        key.close()
        desc = "matvec %3d x %5d q %10d" % (rows, cols, q)
    else:
        B = int(rng.integers(1, 20))
        rows, cols = int(rng.integers(1, 40)), int(rng.integers(1, 4))
        msgs = ["".join(chr(int(v)) for v in rng.integers(32, 127, int(rng.integers(0, 200)))) for _ in range(B)]
        got = ctx.to_host_u32(ctx.hash_to_mat_zq_sha256(msgs, rows, cols, q)).reshape(B, rows, cols)
        ok = all(np.array_equal(got[i], co.hash_to_mat_zq(msgs[i], rows, cols, q)) for i in range(B))
        desc = "hash %2d x %d q %10d B %2d" % (rows, cols, q, B)
    bad += not ok
    print("case %3d kind %d %s %s" % (c, kind, desc, "ok" if ok else "MISMATCH"), flush=True)
print("%d cases, %d mismatches, %.1f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
