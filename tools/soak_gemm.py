"""Diagnostic soak: random shapes / moduli / digit promises / batch sizes for the mod-q GEMM paths against the C
oracle (beyond the seeded cases of tests/test_gpu_random_shapes.py).  usage: python tools/soak_gemm.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from crypto_amd.device import Context, X_BITS, X_BITS_KT, X_PLANES, X_U32, X_I32, bit_stride

PRIMES = [7, 16, 251, 257, 12289, 65521, 65537, 8380417, 16777213, 16777259, 1073741789, 2147483629]
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ctx = Context(0)
SEED = 99
bad = 0
t0 = time.time()
for c in range(ncases):
    q = int(rng.choice(PRIMES))
    rows = int(rng.choice([rng.integers(1, 40), rng.integers(60, 140), rng.integers(250, 300)]))
    cols = int(rng.choice([rng.integers(1, 40), rng.integers(100, 140), rng.integers(250, 262), rng.integers(1000, 1100)]))
    B = int(rng.choice([1, 33, 257, 300, 2049, 8193 + int(rng.integers(0, 700))]))
    kind = int(rng.integers(0, 6))
    A = co.sample_uniform(SEED, 1, c, rows * cols, q).reshape(rows, cols)
    key = ctx.key_upload(A, q)
    msg = co.sample_uniform(SEED, 5, c, B, 2).astype(np.uint8) if rng.integers(0, 2) else None
    E = co.sample_uniform(SEED, 4, c, B * rows, q).reshape(B, rows) if rng.integers(0, 2) else None
    dm = None if msg is None else ctx.to_dev_u8(msg)
    de = None if E is None else ctx.to_dev_u32(E)
    if kind == 0:
        xb = co.sample_bits(SEED, 3, c, B, cols, bit_stride(cols))
        out = ctx.matmul(key, ctx.to_dev_u8(xb), X_BITS, B, addend=de, msg=dm)
        ref = co.matmul_batch(A, q, xb, co.X_BITS, B, addend=E, msg=msg, mt=True)
    elif kind == 4:   # the same bit rows k-tile-major (QF_X_BITS_KT): converted from rows or drawn in that form, planes wider than B
        xb = co.sample_bits(SEED, 3, c, B, cols, bit_stride(cols))
        if rng.integers(0, 2):
            xk = ctx.bits_to_kt(ctx.to_dev_u8(xb), B, cols)
        else:
            import torch
            xk = torch.full(((cols + 127) // 128, B + int(rng.integers(0, 40)), 16), 0xC3, dtype=torch.uint8, device=ctx.device)
            ctx.sample_bits_kt(SEED, 3, c, B, cols, out=xk)
        out = ctx.matmul(key, xk, X_BITS_KT, B, addend=de, msg=dm)
        ref = co.matmul_batch(A, q, xb, co.X_BITS, B, addend=E, msg=msg, mt=True)
    elif kind == 5:   # a wide operand recoded once into digit planes (QF_X_PLANES), product on the planes
        la = 1
        while 256 ** la < q: la += 1
        dg = int(rng.integers(1, la + 1))
        W = [30, 8000, 2000000, 500000000][dg - 1]
        if dg == la or 4 * W >= q // 2:
            X = co.sample_uniform(SEED, 6, c, B * cols, q).reshape(B, cols); dg = la
        else:
            X = co.sample_noise(SEED, 3, c, B * cols, q, W).reshape(B, cols)
        planes = ctx.recode_planes(key, ctx.to_dev_u32(X), X_U32, B, dg)
        out = ctx.matmul(key, planes, X_PLANES, B, addend=de, msg=dm, x_digits=dg)
        ref = co.matmul_batch(A, q, X, co.X_U32, B, addend=E, msg=msg, mt=True)
    elif kind == 1:
        X = co.sample_uniform(SEED, 6, c, B * cols, q).reshape(B, cols)
        out = ctx.matmul(key, ctx.to_dev_u32(X), X_U32, B, addend=de, msg=dm)
        ref = co.matmul_batch(A, q, X, co.X_U32, B, addend=E, msg=msg, mt=True)
    elif kind == 2:
        W = int(rng.choice([1, 30, 3000, 100000]))
        X = co.sample_noise(SEED, 3, c, B * cols, q, W).reshape(B, cols)
        out = ctx.matmul(key, ctx.to_dev_u32(X), X_U32, B, addend=de, msg=dm, x_digits=int(rng.integers(1, 4)))
        ref = co.matmul_batch(A, q, X, co.X_U32, B, addend=E, msg=msg, mt=True)
    else:
        X = rng.integers(-5000, 5001, (B, cols)).astype(np.int32)
        out = ctx.matmul(key, ctx.to_dev_i32(X), X_I32, B, addend=de, msg=dm, x_digits=2)
        ref = co.matmul_batch(A, q, (X.astype(np.int64) % q).astype(np.uint32), co.X_U32, B, addend=E, msg=msg, mt=True)
    ok = np.array_equal(ctx.to_host_u32(out), ref)
    bad += not ok
    print("case %3d kind %d rows %3d cols %4d q %10d B %5d addend %d msg %d %s" % (c, kind, rows, cols, q, B, E is not None, msg is not None, "ok" if ok else "MISMATCH"), flush=True)
    key.close()
print("%d cases, %d mismatches, %.1f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
