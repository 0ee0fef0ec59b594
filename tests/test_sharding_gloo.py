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


class _MockContext:
    """Stands in for crypto_amd.device.Context on a box without a GPU: the same methods the sharded
    call sequence uses, CPU tensors, arithmetic by the oracle -- and a log of the calls made."""

    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = []

    def to_dev_u32(self, a):
        self.calls.append("to_dev_u32")
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.uint32)).view(np.int32).copy())

    def key_upload(self, A, q, transpose=False):
        self.calls.append("key_upload")
        assert isinstance(A, torch.Tensor) and A.dtype == torch.int32      # the broadcast buffer, as on the device
        m = A.numpy().view(np.uint32)
        return {"A": np.ascontiguousarray(m.T if transpose else m), "q": q}

    def sample_bits(self, seed, stream, first_row, rows, cols, stride=None):
        from oracle import coracle as co
        self.calls.append(("sample_bits", first_row, rows))
        return torch.from_numpy(co.sample_bits(seed, stream, first_row, rows, cols, stride or ((cols + 127) // 128) * 16))

    def sample_uniform(self, seed, stream, first, count, q):
        from oracle import coracle as co
        self.calls.append(("sample_uniform", first, count))
        return torch.from_numpy(co.sample_uniform(seed, stream, first, count, q).view(np.int32).copy())

    def matmul(self, key, X, kind, B, addend=None, msg=None, x_digits=0, out=None):
        from oracle import coracle as co
        self.calls.append(("matmul", B))
        return torch.from_numpy(co.matmul_batch(key["A"], key["q"], X.numpy(), kind, B, msg=msg.numpy() if msg is not None else None).view(np.int32).copy())


def _worker_calls(rank, ws, port, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from crypto_amd import sharding
    from oracle import coracle as co
    n, m, q, B = 12, 100, 7993, 37
    ctx = _MockContext()
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m) if rank == 0 else None   # only rank 0 holds the key
    lo, hi, c, msg = sharding.regev_enc_shard(ctx, pk, n, m, q, B, SEED)
    q_out.put((rank, lo, hi, c.numpy().view(np.uint32), msg.numpy(), ctx.calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_crypto_amd_call_sequence_with_mock_context():
    """The product's own sharded call sequence (crypto_amd.sharding.regev_enc_shard: broadcast key ->
    key_upload -> this rank's slice of the generator streams -> one batched product) on two gloo ranks
    with a mock Context: the shards tile the batch, reproduce the single-process result, and only rank 0
    ever touches the key material before the broadcast."""
    from oracle import coracle as co
    ws, port = 2, _free_port()
    mpc = mp.get_context("spawn")
    q_out = mpc.Queue()
    procs = [mpc.Process(target=_worker_calls, args=(r, ws, port, q_out)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted([q_out.get(timeout=120) for _ in range(ws)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n, m, q, B = 12, 100, 7993, 37
    pk = co.sample_uniform(SEED, 1, 0, (n + 1) * m, q).reshape(n + 1, m)
    xb = co.sample_bits(SEED, 3, 0, B, m, 16)
    msg = (co.sample_uniform(SEED, 5, 0, B, 2) & 1).astype(np.uint8)
    whole = co.matmul_batch(pk, q, xb, co.X_BITS, B, msg=msg)
    (_, lo0, hi0, c0, m0, calls0), (_, lo1, hi1, c1, m1, calls1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37)
    assert np.array_equal(np.vstack([c0, c1]), whole)
    assert np.array_equal(np.concatenate([m0, m1]), msg)
    assert calls0 == ["to_dev_u32", "key_upload", ("sample_bits", 0, 19), ("sample_uniform", 0, 19), ("matmul", 19)]
    assert calls1 == ["key_upload", ("sample_bits", 19, 18), ("sample_uniform", 19, 18), ("matmul", 18)]


def test_bench_launcher_refuses_more_ranks_than_gpus():
    """python bench.py --gpus 8 on a node with fewer GPUs must fail loudly, never print an n_gpus=1 line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QF_BENCH_REHEARSAL")}
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node really has 8 GPUs")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "sis", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "--gpus 8" in p.stderr and "refusing" in p.stderr


@pytest.mark.parametrize("total,ws", [(0, 2), (1, 2), (37, 2), (65536, 8), (10, 4), (7, 8)])
def test_shard_range_tiles_the_batch(total, ws):
    from crypto_amd.sharding import shard_range
    edges = [shard_range(total, r, ws) for r in range(ws)]
    assert edges[0][0] == 0 and edges[-1][1] == total
    for a, b in zip(edges, edges[1:]):
        assert a[1] == b[0]
    sizes = [hi - lo for lo, hi in edges]
    assert max(sizes) - min(sizes) <= 1
