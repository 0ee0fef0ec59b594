#!/usr/bin/env python3
"""Ring-kernel timing at n=1024, q=2^30-2^18+1 (BASELINE config 5): polymul (B=65536), RingLPR enc/dec
(B=65536), FDH verify (B=8192), HIP events around back-to-back launches.  QF_LIB_PATH selects a
diagnostic build (tools/build_variant.sh)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crypto_amd.device import Context
SEED = 0x5146414C4C
ctx = Context(0)
n, q, W = 1024, 1073479681, 36
ring = ctx.ring(n, q)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(3):
        torch.cuda.synchronize(); t0.record()
        for _ in range(reps): fn()
        t1.record(); torch.cuda.synchronize()
        best.append(t0.elapsed_time(t1) / reps)
    return min(best), float(np.median(best))
B = 1 << 16
a = ctx.sample_uniform(SEED, 6, 0, B * n, q).view(B, n)
b = ctx.sample_uniform(SEED, 6, B * n, B * n, q).view(B, n)
c = torch.empty_like(a)
for _ in range(30): ring.polymul(a, b, out=c)
m, med = timeit(lambda: ring.polymul(a, b, out=c))
res = {"lib": os.environ.get("QF_LIB_PATH", "default").split("/")[-1], "polymul_ms": round(med, 4), "polymul_per_s": round(B / med * 1e3 / 1e8, 3)}
keys = ctx.sample_uniform(SEED, 1, 0, 2 * n, q).view(2, n)
pk = ring.key(keys)
r = ctx.sample_noise(SEED, 3, 0, B * n, q, W).view(B, n)
e1 = ctx.sample_noise(SEED, 4, 0, B * n, q, W).view(B, n)
e2 = ctx.sample_noise(SEED, 8, 0, B * n, q, W).view(B, n)
mu = ctx.sample_bits(SEED, 5, 0, B, n, n // 8)
u, v = torch.empty_like(r), torch.empty_like(r)
m, med = timeit(lambda: ring.ringlpr_enc(pk, r, e1, e2, mu, u=u, v=v))
res.update({"enc_ms": round(med, 4), "enc_per_s_e8": round(B / med * 1e3 / 1e8, 3)})
sk = ring.key(keys[:1].contiguous())
out = ctx.empty_u8(B, n // 8)
m, med = timeit(lambda: ring.ringlpr_dec(sk, u, v, out=out))
res.update({"dec_ms": round(med, 4), "dec_per_s_e8": round(B / med * 1e3 / 1e8, 3)})
mbar, Bf = 32, 32768
A = ring.key(ctx.sample_uniform(SEED, 1, 0, mbar * n, q).view(mbar, n))
sig = (ctx.sample_uniform(SEED, 7, 0, Bf * mbar * n, 201) - 100).view(Bf, mbar, n)
uu = ctx.sample_uniform(SEED, 6, 0, Bf * n, q).view(Bf, n)
m, med = timeit(lambda: ring.fdh_verify(A, sig, uu, 10 ** 12), reps=10)
res.update({"fdh_ms": round(med, 4), "fdh_per_s_e7": round(Bf / med * 1e3 / 1e7, 3)})
print(json.dumps(res))
