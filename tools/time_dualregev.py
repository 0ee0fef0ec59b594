"""Diagnostic: DualRegev-shaped GEMM (secure128: key 5314 x 230, 3-digit operand, B = 65536) -- kernel time from the context's timers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_U32
ctx = Context(0)
n, m, q, W = 230, 5313, 7764299, 2951
B = 1 << 16
pkT = ctx.sample_uniform(1, 1, 0, (m + 1) * n, q).view(m + 1, n)
key = ctx.key_upload(pkT, q)
s = ctx.sample_uniform(1, 3, 0, B * n, q).view(B, n)
e = ctx.sample_noise(1, 4, 0, B * (m + 1), q, W).view(B, m + 1)
out = ctx.empty_u32(B, m + 1)
ctx.set_timing(True)
def t(fn, reps=10):
    for _ in range(20):
        fn()
    tot = 0.0
    for _ in range(reps):
        fn(); tot += ctx.last_kernel_ms("gemm")
    return tot / reps
print("dualregev with addend    %.4f ms" % t(lambda: ctx.matmul(key, s, X_U32, B, addend=e, out=out)), flush=True)
print("dualregev without addend %.4f ms" % t(lambda: ctx.matmul(key, s, X_U32, B, out=out)), flush=True)
