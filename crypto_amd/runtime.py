"""Process-wide default device context and the randomness the scheme layer draws from.

The reference samples from the OS's CSPRNG at every call (e.g. regev.rs:352-364,401).  Here a scheme
object owns a `Randomness`: per-stream counters over one of the two counter-mode generators of
include/qfgpu.h.

* With an explicit `seed=` (tests, bench, oracle parity): SplitMix64 under that 64-bit seed, so that
  every run -- GPU path and CPU oracle alike -- consumes the same inputs.  SplitMix64 is NOT a CSPRNG:
  its finaliser is a bijection of a 64-bit state, a published matrix A gives away the seed and with
  it the secret streams.  An explicit seed is a statement that confidentiality is not wanted.
* WITHOUT one (the default construction, the reference's behaviour): ChaCha20 under a fresh 256-bit
  key from os.urandom (`qf_sample_*_keyed_dev`).  Public and secret streams of one object share the
  key and differ in the nonce word; the public stream reveals nothing about the others.  `rng.seed`
  is then the pair (key32, nonce).
Streams: 1 uniform A / a, 2 secret, 3 x / r, 4 noise e, 5 messages, 6 free operands.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

MASTER_SEED = 0x5146414C4C  # "QFALL": the seed tests and benches pass EXPLICITLY; never a default


def fresh_key():
    """(key32, nonce) for objects constructed without a seed: 256 bits from the OS's CSPRNG."""
    return (os.urandom(32), 0)

_default_ctx = None


def default_context():
    """Lazily created Context on the current HIP device (no CPU fallback)."""
    global _default_ctx
    if _default_ctx is None:
        import torch
        from .device import Context
        dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        _default_ctx = Context(dev)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


class Randomness:
    """Counter-mode draws; each stream advances its own counter so draws never overlap."""

    def __init__(self, seed: Optional[int] = None, ctx=None):
        if seed is None or isinstance(seed, tuple):
            self.seed = fresh_key() if seed is None else (bytes(seed[0]), int(seed[1]))   # keyed: ChaCha20
        else:
            self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF                                   # seeded: SplitMix64
        self._ctx = ctx
        self.pos = {}

    @property
    def ctx(self):
        return self._ctx or default_context()

    def _take(self, stream: int, count: int) -> int:
        first = self.pos.get(stream, 0)
        self.pos[stream] = first + count
        return first

    def uniform(self, stream: int, count: int, q: int) -> np.ndarray:
        first = self._take(stream, count)
        return self.ctx.to_host_u32(self.ctx.sample_uniform(self.seed, stream, first, count, q))

    # Noise model of chi: "discrete_gauss" (the schemes' default) = D_{Z,s} with the reference's
    # parameters (n, s = alpha*q) -- what MatZq::sample_discrete_gauss(.., q, n, 0, s) draws
    # (regev.rs:356-363); "irwin_hall" = the benchmark generator's Gaussian stand-in of SURVEY 8d,
    # sum of 4 uniforms in [-W, W] (explicit opt-in).  Set through the schemes' noise_model argument.
    gauss = None    # (n, s) when the model is "discrete_gauss"

    def noise_dev(self, stream: int, count: int, q: int, W: int):
        """count draws from chi as a device tensor (canonical residues mod q)."""
        first = self._take(stream, count)
        if self.gauss is not None:
            return self.ctx.sample_discrete_gauss(self.seed, stream, first, count, q, self.gauss[0], self.gauss[1])
        return self.ctx.sample_noise(self.seed, stream, first, count, q, W)

    def noise(self, stream: int, count: int, q: int, W: int) -> np.ndarray:
        return self.ctx.to_host_u32(self.noise_dev(stream, count, q, W))

    def noise_bound(self, W: int) -> int:
        """largest |value| chi can produce under the current model"""
        if self.gauss is not None:
            import math
            return int(math.ceil(self.gauss[1] * math.log2(self.gauss[0])))
        return 4 * W

    def bits(self, stream: int, count: int) -> np.ndarray:
        """count independent bits as a 0/1 uint32 vector (MatZ::sample_uniform(m,1,0,2))."""
        from .device import unpack_bits
        first = self._take(stream, 1)  # one generator row per call
        packed = self.ctx.to_host_u8(self.ctx.sample_bits(self.seed, stream, first, 1, count))
        return unpack_bits(packed, count)[0].astype(np.uint32)
