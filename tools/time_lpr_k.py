"""Diagnostic: per-workgroup fixed cost of the tiled GEMM on the LPR shape -- the same 1025-row key and batch at K = 1024 and K = 2048
columns (2-digit operand: 16 resp. 32 k-tiles per workgroup); T(2K) - T(K) is 16 k-tiles of pure main loop, the rest of T(K) is
prologue + epilogue + dispatch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_U32
ctx = Context(0)
q, W, B, rows = 1073741789, 2319, 1 << 18, 1025
ctx.set_timing(True)
res = {}
for cols in (1024, 2048):
    key = ctx.key_upload(ctx.sample_uniform(1, 1, 0, rows * cols, q).view(rows, cols), q)
    r = ctx.sample_noise(1, 3, 0, B * cols, q, W).view(B, cols)
    e = ctx.sample_noise(1, 4, 0, B * rows, q, W).view(B, rows)
    out = ctx.empty_u32(B, rows)
    for name, add in (("addend", e), ("plain", None)):
        for _ in range(10): ctx.matmul(key, r, X_U32, B, addend=add, x_digits=2, out=out)
        tot = 0.0
        for _ in range(10):
            ctx.matmul(key, r, X_U32, B, addend=add, x_digits=2, out=out); tot += ctx.last_kernel_ms("gemm")
        res[(cols, name)] = tot / 10
    del r, e, out
    key.close()
for name in ("addend", "plain"):
    t1, t2 = res[(1024, name)], res[(2048, name)]
    print("%-6s K=1024 %.4f ms  K=2048 %.4f ms  -> 16 k-tiles of loop %.4f ms, fixed part of the K=1024 launch %.4f ms (%.1f %%)" % (name, t1, t2, t2 - t1, 2 * t1 - t2, 100 * (2 * t1 - t2) / t1), flush=True)
