"""Diagnostic: mean GEMM launch time of the BASELINE shapes with the default tiling (for A/B runs of library variants under
QF_LIB_PATH): python tools/time_default.py [regev] [sis] [lpr]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_BITS, X_U32
ctx = Context(0)
ctx.set_timing(True)

def timeit(fn, reps=40):
    for _ in range(40):
        fn()
    tot = 0.0
    for _ in range(reps):
        fn(); tot += ctx.last_kernel_ms("gemm")
    return tot / reps

which = sys.argv[1:] or ["regev", "sis", "lpr"]
res = []
if "regev" in which:
    n, m, q, B = 256, 6168, 16777213, 65536
    key = ctx.key_upload(ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m), q)
    xb = ctx.sample_bits(1, 3, 0, B, m)
    msg = ctx.sample_bits(1, 5, 0, 1, B).view(-1)[: B // 8]
    out = ctx.empty_u32(B, n + 1)
    res.append("regev %.4f ms" % timeit(lambda: ctx.matmul(key, xb, X_BITS, B, out=out)))
if "regevcold" in which:   # six rotating operand / output sets (6 x 118 MB > the Infinity Cache), as bench.py's headline
    n, m, q, B = 256, 6168, 16777213, 65536
    key = ctx.key_upload(ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m), q)
    xs = [ctx.sample_bits(1, 3, i, B, m) for i in range(6)]
    outs = [ctx.empty_u32(B, n + 1) for i in range(6)]
    it = {"i": 0}
    def step():
        i = it["i"] = (it["i"] + 1) % 6
        ctx.matmul(key, xs[i], X_BITS, B, out=outs[i])
    res.append("regev(cold) %.4f ms" % timeit(step, 60))
if "sis" in which:
    n, q = 512, 8380417
    m, B = n * 23, 1 << 18
    key = ctx.key_upload(ctx.sample_uniform(1, 1, 0, n * m, q).view(n, m), q)
    xb = ctx.sample_bits(1, 3, 0, B, m)
    out = ctx.empty_u32(B, n)
    res.append("sis(2^18) %.4f ms" % timeit(lambda: ctx.matmul(key, xb, X_BITS, B, out=out), 10))
if "lpr" in which:
    n, q, W, B = 1024, 1073741789, 2319, 1 << 18
    key = ctx.key_upload(ctx.sample_uniform(1, 1, 0, (n + 1) * n, q).view(n + 1, n), q)
    r = ctx.sample_noise(1, 3, 0, B * n, q, W).view(B, n)
    e = ctx.sample_noise(1, 4, 0, B * (n + 1), q, W).view(B, n + 1)
    out = ctx.empty_u32(B, n + 1)
    res.append("lpr %.4f ms" % timeit(lambda: ctx.matmul(key, r, X_U32, B, addend=e, x_digits=2, out=out), 10))
print(os.environ.get("QF_LIB_PATH", "default lib"), " ".join(res), flush=True)
