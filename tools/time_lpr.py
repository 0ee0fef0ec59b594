"""Diagnostic: LPR-shaped GEMM (n=1024, 2-digit operand) with and without the addend; kernel time from the context's timers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_U32
ctx = Context(0)
n, q, W = 1024, 1073741789, 2319
B = 1 << 18
pk = ctx.sample_uniform(1, 1, 0, (n + 1) * n, q).view(n + 1, n)
key = ctx.key_upload(pk, q)
r = ctx.sample_noise(1, 3, 0, B * n, q, W).view(B, n)
e = ctx.sample_noise(1, 4, 0, B * (n + 1), q, W).view(B, n + 1)
out = ctx.empty_u32(B, n + 1)
ctx.set_timing(True)
def t(fn, reps=10):
    for _ in range(20):
        fn()
    tot = 0.0
    for _ in range(reps):
        fn(); tot += ctx.last_kernel_ms("gemm")
    return tot / reps
print("with addend    %.4f ms" % t(lambda: ctx.matmul(key, r, X_U32, B, addend=e, x_digits=2, out=out)), flush=True)
print("without addend %.4f ms" % t(lambda: ctx.matmul(key, r, X_U32, B, x_digits=2, out=out)), flush=True)
