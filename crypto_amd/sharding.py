"""Multi-GPU layout of the hot path: one process per GPU, the batch index range
[0, B) is cut into contiguous equal slices, read-only keys are replicated ONCE with a broadcast
(RCCL over xGMI when the backend is "nccl"), and there is no collective on the data path --
ciphertexts, digests and products stay sharded on the GPU that made them.

Everything here works on whatever backend `torch.distributed` was initialised with, so the
logic is exercised by world_size-2 gloo tests on CPU tensors (tests/test_sharding_gloo.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`; slices differ by at most one item and tile [0,total)."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def replicate(t: Optional[torch.Tensor], shape, dtype, device, src: int = 0) -> torch.Tensor:
    """Broadcast a tensor that only `src` holds; other ranks pass None.  One collective."""
    rank, ws = world()
    if ws == 1:
        assert t is not None
        return t
    if rank != src:
        t = torch.empty(shape, dtype=dtype, device=device)
    else:
        assert tuple(t.shape) == tuple(shape) and t.dtype == dtype
        t = t.contiguous()
    dist.broadcast(t, src=src)
    return t


def replicate_key(ctx, A: Optional[np.ndarray], rows: int, cols: int, q: int, transpose: bool = False, src: int = 0):
    """Public key / SIS matrix: uint32 matrix broadcast from `src`, recoded locally on every GPU
    (the digit planes are derived data, so only rows*cols*4 bytes cross xGMI)."""
    rank, _ = world()
    t = ctx.to_dev_u32(A) if rank == src else None
    t = replicate(t, (rows, cols), torch.int32, ctx.device, src)
    return ctx.key_upload(t, q, transpose=transpose)


def replicate_ring_key(ring, polys: Optional[np.ndarray], npoly: int, src: int = 0):
    """Ring keys (a, b, s, FDH parity-check row): npoly*n*4 bytes broadcast, NTT done locally."""
    rank, _ = world()
    t = ring.ctx.to_dev_u32(np.asarray(polys, np.uint32).reshape(npoly, ring.n) % ring.q) if rank == src else None
    t = replicate(t, (npoly, ring.n), torch.int32, ring.ctx.device, src)
    return ring.key(t)


def regev_enc_shard(ctx, pk: Optional[np.ndarray], n: int, m: int, q: int, total: int, seed: int,
                    x_stream: int = 3, msg_stream: int = 5):
    """What ONE rank does for a batch of `total` Regev encryptions sharded over the process group --
    the call sequence bench.py times: the key from rank 0 by one broadcast (recoded locally), this
    rank's contiguous slice [lo, hi) of the global randomness and message streams drawn on its own
    device, one batched product on kernel (a).  Returns (lo, hi, ciphertexts (hi-lo, n+1), message bits);
    the outputs stay on the GPU that made them."""
    from .device import X_BITS
    rank, ws = world()
    key = replicate_key(ctx, pk, n + 1, m, q)
    lo, hi = shard_range(total, rank, ws)
    B = hi - lo
    xb = ctx.sample_bits(seed, x_stream, lo, B, m)
    msg = (ctx.sample_uniform(seed, msg_stream, lo, B, 2) & 1).to(torch.uint8)
    c = ctx.matmul(key, xb, X_BITS, B, msg=msg)
    return lo, hi, c, msg


def gather_counts(local_count: int) -> int:
    """Sum of per-rank item counts (bench bookkeeping only; not on the data path)."""
    rank, ws = world()
    if ws == 1:
        return local_count
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([local_count], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    return int(t.item())
