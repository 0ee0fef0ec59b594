"""Diagnostic: mean launch time of the GEMM for the BASELINE shapes under the tile overrides
QF_GEMM_WN / QF_GEMM_MW (read by the dispatcher at every call).  Verifies each variant against the
default tiling bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_BITS, X_U32
ctx = Context(0)

def timeit(fn, reps=60):
    for _ in range(40):   # settle the clock the chip holds under load
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def sweep(name, fn, out, ops, combos):
    for k in ("QF_GEMM_WN", "QF_GEMM_MW"):
        os.environ.pop(k, None)
    fn(); torch.cuda.synchronize(); ref = out.clone()
    print("%s default %.4f ms" % (name, timeit(fn)), flush=True)
    for wn, mw in combos:
        os.environ["QF_GEMM_WN"] = str(wn); os.environ["QF_GEMM_MW"] = str(mw)
        out.zero_(); fn(); torch.cuda.synchronize()
        ok = torch.equal(out, ref)
        ms = timeit(fn)
        print("%s WN=%d MW=%d %.4f ms  %.0f TOP/s  %s" % (name, wn, mw, ms, ops / ms / 1e9, "ok" if ok else "MISMATCH"), flush=True)
    for k in ("QF_GEMM_WN", "QF_GEMM_MW"):
        os.environ.pop(k, None)

which = sys.argv[1:] or ["regev", "sis", "lpr"]
if "regev" in which:
    n, m, q, B = 256, 6168, 16777213, 65536
    pk = ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m)
    key = ctx.key_upload(pk, q)
    xb = ctx.sample_bits(1, 3, 0, B, m)
    out = ctx.empty_u32(B, n + 1)
    sweep("regev", lambda: ctx.matmul(key, xb, X_BITS, B, out=out), out, 2 * 3 * (n + 1) * m * B,
          [(3, 1), (2, 1), (1, 1), (2, 2), (1, 2)])
if "sis" in which:
    n, q = 512, 8380417
    m = n * 23
    B = 1 << 18
    A = ctx.sample_uniform(1, 1, 0, n * m, q).view(n, m)
    key = ctx.key_upload(A, q)
    xb = ctx.sample_bits(1, 3, 0, B, m)
    out = ctx.empty_u32(B, n)
    sweep("sis", lambda: ctx.matmul(key, xb, X_BITS, B, out=out), out, 2 * 3 * n * m * B,
          [(4, 1), (2, 1), (2, 2), (1, 2)])
if "lpr" in which:
    n, q, W = 1024, 1073741789, 2319
    B = 1 << 16
    pk = ctx.sample_uniform(1, 1, 0, (n + 1) * n, q).view(n + 1, n)
    key = ctx.key_upload(pk, q)
    r = ctx.sample_noise(1, 3, 0, B * n, q, W).view(B, n)
    e = ctx.sample_noise(1, 4, 0, B * (n + 1), q, W).view(B, n + 1)
    out = ctx.empty_u32(B, n + 1)
    ctx.set_timing(True)
    def run():
        ctx.matmul(key, r, X_U32, B, addend=e, x_digits=2, out=out)
    for wn, mw in [(0, 0), (2, 1), (3, 1), (4, 1)]:
        if wn:
            os.environ["QF_GEMM_WN"] = str(wn); os.environ["QF_GEMM_MW"] = str(mw)
        run(); run()
        tot = 0.0
        for _ in range(5):
            run(); tot += ctx.last_kernel_ms("gemm")
        print("lpr WN=%d MW=%d gemm %.4f ms %.0f TOP/s" % (wn, mw, tot / 5, 2 * 4 * 2 * (n + 1) * n * B / (tot / 5) / 1e9), flush=True)
