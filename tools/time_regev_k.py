"""Diagnostic: fixed part of the Regev cfg2 GEMM launch (bit operands, key 257 x m, B = 65,536): the same launch at m = 6,168 and
m = 12,336; T(2m) - T(m) is pure main loop (48.2 k-tiles), the rest of T(m) is prologue + epilogue + dispatch of the two
workgroup tiles each CU runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_BITS
ctx = Context(0)
ctx.set_timing(True)
n, q, B = 256, 16777213, 65536
res = {}
for m in (6168, 12336):
    key = ctx.key_upload(ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m), q)
    xb = ctx.sample_bits(1, 3, 0, B, m)
    msg = (ctx.sample_uniform(1, 5, 0, B, 2)).to(torch.uint8)
    out = ctx.empty_u32(B, n + 1)
    for name, mm in (("plain", None), ("msg", msg), ("plain2", None), ("msg2", msg)):
        for _ in range(40): ctx.matmul(key, xb, X_BITS, B, msg=mm, out=out)
        tot = 0.0
        for _ in range(40):
            ctx.matmul(key, xb, X_BITS, B, msg=mm, out=out); tot += ctx.last_kernel_ms("gemm")
        res[(m, name)] = tot / 40
    key.close()
for name in ("msg", "plain", "msg2", "plain2"):
    t1, t2 = res[(6168, name)], res[(12336, name)]
    print("%-6s m=6168 %.4f ms  m=12336 %.4f ms  -> loop of one m %.4f ms, fixed part %.4f ms (%.1f %%)" % (name, t1, t2, t2 - t1, 2 * t1 - t2, 100 * (2 * t1 - t2) / t1), flush=True)
