#!/usr/bin/env python3
"""Search a linear XOR swizzle of the low 5 LDS address bits that makes every phase layout of the
wave-per-polynomial NTT (crypto_amd/csrc/qf_ntt_wave.hip) bank-conflict free for 32-lane groups of
ds_read_b32 / ds_write_b32.  addr = i ^ XOR_{b>=5, bit b of i set} mask[b].  Prints mask tables."""
import itertools, random, sys

def phases(LG):
    RB = LG - 6
    NP = (LG + RB - 1) // RB
    out = []
    for P in range(NP):
        raw = LG - (P + 1) * RB
        split = raw < 0
        lo = 0 if split else raw
        nst = LG - P * RB if split else RB
        regbits = [lo + k if k < nst else LG - (RB - nst) + (k - nst) for k in range(RB)]
        lanebits = [b for b in range(LG) if b not in regbits]   # ascending: lane bit j -> index bit lanebits[j]
        out.append((regbits, lanebits))
    return out

def rank5(vecs):
    basis = []
    for v in vecs:
        for b in basis:
            v = min(v, v ^ b)
        if v:
            basis.append(v)
    return len(basis)

def ok(LG, masks):
    col = lambda b: (1 << b) if b < 5 else masks[b]
    for regbits, lanebits in phases(LG):
        # lanes 0..31 vary lane bits 0..4 ; lanes 32..63 same with lane bit 5 fixed
        if rank5([col(b) for b in lanebits[:5]]) < 5:
            return False
    return True

for LG in range(7, 13):
    found = None
    nb = LG - 5
    rng = random.Random(1)
    for _ in range(200000):
        masks = {5 + j: rng.randrange(32) for j in range(nb)}
        if ok(LG, masks):
            found = masks
            break
    print(LG, phases(LG), found)
