"""Diagnostic: where does a wave of qf_ring_wave_kernel<10,polymul,lazy> spend its time?  Needs a QF_STAMP build
(tools/build_variant.sh STAMP -DQF_STAMP).  Every stamp drains the wave's outstanding memory operations first, so a
section's figure includes the latency of what it issued; s_memtime ticks at a constant 100 MHz."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crypto_amd.device import Context
ctx = Context(0)
n, q, B = 1024, 1073479681, 1 << 16
ring = ctx.ring(n, q)
a = ctx.sample_uniform(1, 6, 0, B * n, q).view(B, n)
b = ctx.sample_uniform(1, 6, B * n, B * n, q).view(B, n)
c = torch.empty_like(a)
nwaves = 256 * 8 * 4
dbg = torch.zeros(nwaves * 8, dtype=torch.int64, device="cuda")
for _ in range(5):
    ring.polymul(a, b, out=c)
torch.cuda.synchronize()
ctx.lib.qf_debug_set_buffer_ring(ctypes.c_void_p(dbg.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ring.polymul(a, b, out=c); e1.record(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwaves, 8).astype(np.float64)
d = d[d[:, 6] > 0]
per = d[:, :6] / d[:, 6:7]
tot = (d[:, 7] / d[:, 6]).mean()
names = ["global loads + range reduction", "2 forward transforms", "pointwise product", "inverse transform", "scaling + stores", "  of the transforms: 6 LDS transposes"]
print("launch %.3f ms (stamped build), %d waves, %.1f products per wave; ticks of 10 ns per product and wave:" % (e0.elapsed_time(e1), len(d), d[:, 6].mean()))
for i, nm in enumerate(names):
    print("  %-42s %8.1f  %.3f" % (nm, per[:, i].mean(), per[:, i].mean() / tot))
print("  %-42s %8.1f" % ("total", tot))
