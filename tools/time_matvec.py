#!/usr/bin/env python3
"""Kernel (c) timing: single-message SIS hash (n=512, m=11776, q=8380417) over 14 rotated keys
(24 keys x 18.1 MB of digit planes = 434 MB: beyond the 256 MiB Infinity Cache), back-to-back launches bracketed by HIP events, for the launch variants the
environment selects (QF_MATVEC_U = k-steps per wave).
Variants are interleaved in rounds inside one process."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crypto_amd.device import Context
SEED = 0x5146414C4C
ctx = Context(0)
n, q = 512, 8380417
m = n * 23
nkeys = int(os.environ.get("NKEYS", "24"))   # 24 x 18.1 MB of planes = 434 MB
keys = [ctx.key_upload(ctx.sample_uniform(SEED, 1, i * n * m, n * m, q).view(n, m), q) for i in range(nkeys)]
x = ctx.sample_uniform(SEED, 6, 0, m, 2)
y = ctx.empty_u32(n)
base = {"QF_MATVEC_U": "0"}
variants = [("default", {}), ("U1", {"QF_MATVEC_U": "1"}), ("U2", {"QF_MATVEC_U": "2"}), ("U3", {"QF_MATVEC_U": "3"}), ("U4", {"QF_MATVEC_U": "4"})]
static = os.environ.get("STATIC") == "1"
fn = ctx.lib.qf_matvec_modq_static_dev if static else ctx.lib.qf_matvec_modq_dev
res = {k: [] for k, _ in variants}
reps = 4 * nkeys
for rnd in range(6):
    for name, env in variants:
        os.environ.update(base); os.environ.update(env)
        for i in range(nkeys):
            fn(ctx.h, keys[i].h, x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(ctx.h, keys[i % nkeys].h, x.data_ptr(), y.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / reps * 1e3)
bytes_alg = n * m * 3
for name, v in res.items():
    med = float(np.median(v))
    print(json.dumps({"variant": name, "static": static, "us_per_hash_median": round(med, 2), "min": round(min(v), 2),
                      "GBps_algorithmic_18.1MB": round(bytes_alg / med / 1e3, 1), "frac_of_8TBps": round(bytes_alg / med / 1e3 / 8000, 3)}))
