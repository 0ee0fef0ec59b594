"""Diagnostic: where does a GEMM workgroup spend its cycles?  Needs the QF_STAMP build
(make -C crypto_amd/csrc STAMP=1).  Prints mean per-wave cycle shares."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crypto_amd.device import Context, X_BITS
ctx = Context(0)
if "dualregev" in sys.argv[1:]:
    from crypto_amd.device import X_U32
    n, m, q, B = 230, 5313, 7764299, 1 << 16
    key = ctx.key_upload(ctx.sample_uniform(1, 1, 0, (m + 1) * n, q).view(m + 1, n), q)
    sv = ctx.sample_uniform(1, 3, 0, B * n, q).view(B, n)
    e = ctx.sample_noise(1, 4, 0, B * (m + 1), q, 2951).view(B, m + 1)
    out = ctx.empty_u32(B, m + 1)
    nwaves = (B // 256) * 42 * 8          # <3,1,4> tiles: 167 row blocks / 4
    run = lambda: ctx.matmul(key, sv, X_U32, B, addend=e, out=out)
elif "lpr" in sys.argv[1:]:
    from crypto_amd.device import X_U32
    n, q, B = 1024, 1073741789, 1 << 17
    pk = ctx.sample_uniform(1, 1, 0, (n + 1) * n, q).view(n + 1, n)
    key = ctx.key_upload(pk, q)
    r = ctx.sample_noise(1, 3, 0, B * n, q, 2319).view(B, n)
    e = ctx.sample_noise(1, 4, 0, B * (n + 1), q, 2319).view(B, n + 1)
    out = ctx.empty_u32(B, n + 1)
    nwaves = (B // 256) * 11 * 8          # <4,1,3> tiles: 33 row blocks / 3
    run = lambda: ctx.matmul(key, r, X_U32, B, addend=e, x_digits=2, out=out)
else:
    n, m, q, B = 256, 6168, 16777213, 65536
    pk = ctx.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m)
    key = ctx.key_upload(pk, q)
    xb = ctx.sample_bits(1, 3, 0, B, m)
    nwaves = (B // 256) * 3 * 8
    run = lambda: ctx.matmul(key, xb, X_BITS, B)
dbg = torch.zeros(nwaves * 12, dtype=torch.int64, device="cuda")
ctx.lib.qf_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(3):
    run()
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
d = raw[:nwaves * 8].reshape(nwaves, 8)
d2 = raw[nwaves * 8:].reshape(nwaves, 4)
tot = (d[:, 6] - d[:, 5]).mean()
print("per-wave cycles: total %.0f loop %.0f vmcnt-wait %.0f barrier %.0f at-wait %.0f epilogue %.0f" %
      (tot, d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), d[:, 4].mean()))
print("shares of total: loop %.3f vm %.3f bar %.3f at %.3f epi %.3f" % tuple(d[:, i].mean() / tot for i in range(5)))
print("kernel span cycles (max end - min start): %d" % (d[:, 6].max() - d[:, 5].min()))
print("epilogue split: addend-load wait %.0f  compute+store issue %.0f  final store drain %.0f (cycles per wave, 100 MHz s_memtime ticks x clock ratio)" %
      (d2[:, 0].mean(), d2[:, 1].mean(), d2[:, 2].mean()))
print("barrier wait by wave slot 0..7:", " ".join("%.0f" % d[w::8, 2].mean() for w in range(8)))
print("loop cycles by wave slot 0..7: ", " ".join("%.0f" % d[w::8, 0].mean() for w in range(8)))
print("epilogue by wave slot 0..7:    ", " ".join("%.0f" % d[w::8, 4].mean() for w in range(8)))
