"""Diagnostic: packed-remainder GEMM vs the padded tiling on the Regev shape, alternating in one
process (the dispatcher reads QF_GEMM_NOPACK at every call), 50 launches per sample."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_BITS
ctx = Context(0)
n, m, q, B = 256, 6168, 16777213, 65536
pk = ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m)
key = ctx.key_upload(pk, q)
xb = ctx.sample_bits(1, 3, 0, B, m)
out = ctx.empty_u32(B, n + 1)
def timeit(reps=50):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.matmul(key, xb, X_BITS, B, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(20):
    ctx.matmul(key, xb, X_BITS, B, out=out)
for rnd in range(5):
    os.environ.pop("QF_GEMM_NOPACK", None)
    a = timeit()
    os.environ["QF_GEMM_NOPACK"] = "1"
    b = timeit()
    print("round %d packed %.4f ms  padded %.4f ms" % (rnd, a, b), flush=True)
