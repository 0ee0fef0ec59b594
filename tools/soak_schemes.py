"""Diagnostic soak: gen -> enc -> dec round trips of every scheme class over a range of security parameters
(single messages, multi-bit, batches).  usage: python tools/soak_schemes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crypto_amd.construction.pk_encryption import (LPR, DualRegev, DualRegevWithDiscreteGaussianRegularity, Regev,
                                                    RegevWithDiscreteGaussianRegularity, RingLPR)
from crypto_amd.runtime import default_context
ctx = default_context()
bad = 0
for cls, ns in [(Regev, [10, 17, 33, 64, 100]), (DualRegev, [10, 21, 40, 70]), (LPR, [10, 25, 48, 90, 130]),
                (RegevWithDiscreteGaussianRegularity, [2, 4, 7, 15, 40]), (DualRegevWithDiscreteGaussianRegularity, [2, 5, 9, 40]),
                (RingLPR, [16, 32, 64, 128, 256])]:
    for n in ns:
        for model in ([None, "discrete_gauss"] if cls in (Regev, DualRegev, LPR, RingLPR) else [None]):
            kw = {} if model is None else {"noise_model": model}
            s = cls.new_from_n(n, ctx=ctx, seed=1000 + n, **kw)
            pk, sk = s.gen()
            if cls is RingLPR:
                msgs = [0, 1, (1 << n) - 1, 12345 % (1 << n)]
                ok = all(s.dec(sk, s.enc(pk, m)) == m for m in msgs)
                u, v = s.enc_batch(pk, msgs)
                ok = ok and [int(x) for x in s.dec_batch(sk, u, v)] == msgs if hasattr(s, "dec_batch") else ok
            else:
                ok = all(s.dec(sk, s.enc(pk, m)) == m for m in (0, 1, 1, 0))
                ok = ok and s.dec_multiple_bits(sk, s.enc_multiple_bits(pk, 37)) == 37
                msgs = (np.arange(500) % 2)
                ok = ok and s.dec_batch(sk, s.enc_batch(pk, msgs)).tolist() == msgs.tolist()
            bad += not ok
            print("%-42s n %4d q %10d model %-14s %s" % (cls.__name__, n, getattr(s, "q_int", s.q), model, "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
