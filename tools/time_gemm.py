"""Diagnostic: mean launch time of the Regev-shaped GEMM for whichever library QF_LIB_PATH names."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_BITS
ctx = Context(0)
n, m, q, B = 256, 6168, 16777213, 65536
pk = ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m)
key = ctx.key_upload(pk, q)
xb = ctx.sample_bits(1, 3, 0, B, m)
out = ctx.empty_u32(B, n + 1)
for _ in range(3):
    ctx.matmul(key, xb, X_BITS, B, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ctx.matmul(key, xb, X_BITS, B, out=out)
e1.record()
torch.cuda.synchronize()
print(os.environ.get("QF_LIB_PATH", "default").split("libqfgpu")[-1], "%.4f ms" % (e0.elapsed_time(e1) / 10))
