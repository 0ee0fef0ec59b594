"""Diagnostic: would Regev n=256 (257 rows) run faster as a 256-row product with 4-block tiles + a separate
launch for the last row?  Times the 256-row and the 1-row (packed remainder only) products on the same operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_BITS
ctx = Context(0)
ctx.set_timing(True)
n, m, q, B = 256, 6168, 16777213, 65536
pk = ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m)
xb = ctx.sample_bits(1, 3, 0, B, m)
def t(key, rows, label):
    out = ctx.empty_u32(B, rows)
    for _ in range(30): ctx.matmul(key, xb, X_BITS, B, out=out)
    tot = 0.0
    for _ in range(30):
        ctx.matmul(key, xb, X_BITS, B, out=out); tot += ctx.last_kernel_ms("gemm")
    print("%s rows=%d gemm %.4f ms" % (label, rows, tot / 30), flush=True)
for rep in range(2):
    t(ctx.key_upload(pk, q), n + 1, "full 257")
    t(ctx.key_upload(pk[:n].contiguous(), q), n, "first 256")
    t(ctx.key_upload(pk[n:].contiguous(), q), 1, "last row")
    t(ctx.key_upload(pk[:128].contiguous(), q), 128, "128 rows")
