"""Diagnostic: Regev decode (n = 256, 65,536 ciphertexts per launch, 6 rotating sets so the rows come from HBM) -- kernel time from the context's timers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context
ctx = Context(0)
n, q, NS = 256, 16777213, 6
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = ctx.sample_uniform(1, 2, 0, n, q)
sets = [ctx.sample_uniform(1, 9, i * B * (n + 1), B * (n + 1), q).view(B, n + 1) for i in range(NS)]
out = ctx.empty_u8(B)
ctx.set_timing(True)
for _ in range(30):
    for c in sets: ctx.dot_decode(w, q, c, out=out)
tot, k = 0.0, 0
for _ in range(20):
    for c in sets:
        ctx.dot_decode(w, q, c, out=out); tot += ctx.last_kernel_ms("decode"); k += 1
ms = tot / k
print("decode %.2f us  %.0f GB/s  %.3f of 8 TB/s" % (ms * 1e3, B * (n + 1) * 4 / ms / 1e6, B * (n + 1) * 4 / ms / 1e6 / 8000), flush=True)
