"""crypto_amd -- MI355X-native (gfx950) implementation of qFALL-crypto's data-parallel lattice
arithmetic: Z_q matrix products (Regev / DualRegev / LPR / SIS hash) on int8 MFMA, negacyclic
R_q products (RingLPR, GPV-ring FDH verify) by LDS-resident NTT, HBM-streaming mat-vec.

The product path is `crypto_amd.device` (C ABI in include/qfgpu.h) and the scheme layer in
`crypto_amd.construction` / `crypto_amd.primitive`, which mirrors the reference's trait API.
Nothing here imports `oracle/`.
"""
__version__ = "0.1.0"
