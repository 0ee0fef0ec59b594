"""CPU, world_size 2 over gloo: the multi-GPU layout (contiguous batch shards, keys replicated
by one broadcast, no collective on the data path) reproduces the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5146414C4C


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from crypto_amd import sharding
    from oracle import coracle as co
    n, m, q, B = 12, 100, 7993, 37
    # only rank 0 "generates" the key; everyone gets it through ONE broadcast
    pk0 = torch.from_numpy(co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m).view(np.int32).copy()) if rank == 0 else None
    pk = sharding.replicate(pk0, (n + 1, m), torch.int32, torch.device("cpu"))
    lo, hi = sharding.shard_range(B, rank, ws)
    # each rank draws exactly its slice of the global randomness streams and processes it alone
    xb = co.sample_bits(SEED, 3, lo, hi - lo, m, 16)
    msg = co.sample_uniform(SEED, 5, lo, hi - lo, 2).astype(np.uint8)
    c = co.matmul_batch(pk.numpy().view(np.uint32), q, xb, co.X_BITS, hi - lo, msg=msg)
    total = sharding.gather_counts(hi - lo)
    q_out.put((rank, lo, hi, c, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process():
    from oracle import coracle as co
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, q_out)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted([q_out.get(timeout=120) for _ in range(ws)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n, m, q, B = 12, 100, 7993, 37
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m)
    xb = co.sample_bits(SEED, 3, 0, B, m, 16)
    msg = co.sample_uniform(SEED, 5, 0, B, 2).astype(np.uint8)
    whole = co.matmul_batch(pk, q, xb, co.X_BITS, B, msg=msg)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == B
    assert np.array_equal(np.vstack([res[0][3], res[1][3]]), whole)
    assert res[0][4] == B and res[1][4] == B


@pytest.mark.parametrize("total,ws", [(0, 2), (1, 2), (37, 2), (65536, 8), (10, 4), (7, 8)])
def test_shard_range_tiles_the_batch(total, ws):
    from crypto_amd.sharding import shard_range
    edges = [shard_range(total, r, ws) for r in range(ws)]
    assert edges[0][0] == 0 and edges[-1][1] == total
    for a, b in zip(edges, edges[1:]):
        assert a[1] == b[0]
    sizes = [hi - lo for lo, hi in edges]
    assert max(sizes) - min(sizes) <= 1
