"""Where the single-bit Regev cycle (BASELINE config 1) spends its time in the Python host layer."""
import cProfile, pstats, time, sys
from crypto_amd.device import Context
from crypto_amd.construction.pk_encryption import Regev

ctx = Context(0)
r = Regev.new_from_n(50, ctx=ctx, seed=7)
pk, sk = r.gen(); r.dec(sk, r.enc(pk, 1))
reps = 200
def cyc():
    for _ in range(reps):
        pk, sk = r.gen()
        assert r.dec(sk, r.enc(pk, 1)) == 1
t = time.perf_counter(); cyc(); print("cycle ms", (time.perf_counter() - t) / reps * 1e3)
t = time.perf_counter()
for _ in range(reps): r.gen()
print("gen ms", (time.perf_counter() - t) / reps * 1e3)
pk, sk = r.gen()
t = time.perf_counter()
for _ in range(reps): c = r.enc(pk, 1)
print("enc ms", (time.perf_counter() - t) / reps * 1e3)
t = time.perf_counter()
for _ in range(reps): r.dec(sk, c)
print("dec ms", (time.perf_counter() - t) / reps * 1e3)
pr = cProfile.Profile(); pr.enable(); cyc(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
# per-phase times inside the combined loop
tg = te = td = 0.0
for _ in range(reps):
    t0 = time.perf_counter(); pk, sk = r.gen()
    t1 = time.perf_counter(); c = r.enc(pk, 1)
    t2 = time.perf_counter(); b = r.dec(sk, c)
    t3 = time.perf_counter()
    tg += t1 - t0; te += t2 - t1; td += t3 - t2
print("in-cycle gen/enc/dec ms", tg / reps * 1e3, te / reps * 1e3, td / reps * 1e3)
