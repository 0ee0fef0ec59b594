#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): batched Regev enc+dec, n=256, m=6168, q=16,777,213,
65,536 one-bit messages per GPU, through the C ABI on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch: encrypt B messages (kernel a, int8-MFMA mod-q
GEMM) and decrypt them (batched dot-decode).  Inputs (public key, bit-packed randomness x,
message bits) are generated on the device by the seeded generator and are resident in HBM when
the timed region starts.  Multi-GPU: one process per GPU; rank 0 generates the key, the uint32
key matrix is replicated with ONE RCCL broadcast, every rank recodes it locally and encrypts
its own shard of B messages (weak scaling, no collective on the data path).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the GEMM): algorithmic
int8 ops per launch / its mean launch duration measured with HIP events on the launch stream
inside the timed region.  `cpu_baseline` is the C restatement of the reference (oracle/,
one message per call) timed on the host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SEED = 0x5146414C4C
PROFILE_DIR = os.path.join(ROOT, "profiles", "r1")


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC passes (tools/profile_round.sh:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench, FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when no profile of this tree is committed."""
    try:
        t = json.load(open(os.path.join(PROFILE_DIR, "traffic.json")))
    except (OSError, ValueError):
        return None
    for k, v in t.items():
        if k.startswith(kernel_prefix) and v.get("fetch_bytes_per_launch_x2") is not None and v.get("write_bytes_per_launch") is not None:
            return v["fetch_bytes_per_launch_x2"] + v["write_bytes_per_launch"]
    return None


INT8_PEAK_TOPS = 5033.0   # 2048 int8 op/clk/SIMD x 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md "Matrix cores": I8 = 2x BF16)
HBM_PEAK_GBS = 8000.0


def spin(fn, ms):
    """Untimed: run fn back to back for about `ms` milliseconds of GPU time (DVFS settle)."""
    if ms <= 0:
        return
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spin-ms", type=float, default=60.0,
                    help="untimed: keep the GPU busy this long before the W warmup steps (the clock the chip "
                         "holds under load settles ~20 ms after idle; measured steady state is 15 %% faster than "
                         "the first launches)")
    ap.add_argument("--batch", type=int, default=65536, help="messages per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-ring", action="store_true", help="skip the secondary ring-polymul measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--workload", default="regev", choices=["regev", "sis", "lpr", "ringlpr", "fdh", "matvec", "regev_cycle"],
                    help="regev = the headline (BASELINE config 2); the others are the remaining configs, "
                         "each printing its own JSON line")
    return ap.parse_args()


def _init_dist(world):
    """One process per GPU, RCCL (backend "nccl").  QF_BENCH_REHEARSAL=1 is for rehearsing the
    multi-process code path on a ONE-GPU box only: every rank uses cuda:0 and the collectives go
    over gloo -- its numbers mean nothing."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("QF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return local, dist


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local, dist = _init_dist(world)

    from crypto_amd.device import Context, X_BITS, bit_stride
    ctx = Context(local)
    dev = ctx.device

    # ---- workload: BASELINE config 2 -------------------------------------------------------
    n, m, q = 256, 6168, 16777213
    B = args.batch
    W = 4  # noise half-width of the Gaussian stand-in: |e| <= 16, far inside the q/4 decryption margin
    la, lb = 3, 1
    ops_per_enc = 2 * la * lb * (n + 1) * m            # algorithmic int8 ops (9,511,056)

    # key generation on rank 0 (regev.rs:350-374), then ONE broadcast of the uint32 key
    pk = ctx.empty_u32(n + 1, m)
    s = ctx.empty_u32(n)
    if rank == 0:
        ctx.sample_uniform(SEED, 1, 0, n * m, q, out=pk[:n].view(-1))
        ctx.sample_uniform(SEED, 2, 0, n, q, out=s)
        e = ctx.sample_noise(SEED, 4, 0, m, q, W)
        key_a = ctx.key_upload(pk[:n].contiguous(), q)
        bvec = ctx.empty_u32(m)
        assert ctx.lib.qf_vecmat_modq_dev(ctx.h, key_a.h, s.data_ptr(), bvec.data_ptr()) == 0
        pk[n] = ((bvec.to(torch.int64) & 0xFFFFFFFF) + (e.to(torch.int64) & 0xFFFFFFFF)).remainder(q).to(torch.int32)
        key_a.close()
    t_bcast = 0.0
    if world > 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(pk, src=0)
        dist.broadcast(s, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    key = ctx.key_upload(pk, q)
    w = ((q - (s.to(torch.int64) & 0xFFFFFFFF)) % q).to(torch.int32)

    # this rank's shard of the batch: rows [rank*B, (rank+1)*B) of the global generator streams
    xbits = ctx.sample_bits(SEED, 3, rank * B, B, m)
    msg = (ctx.sample_uniform(SEED, 5, rank * B, B, 2) & 1).to(torch.uint8)
    cipher = ctx.empty_u32(B, n + 1)
    bits = ctx.empty_u8(B)

    def step():
        ctx.matmul(key, xbits, X_BITS, B, msg=msg, out=cipher)
        ctx.dot_decode(w, q, cipher, out=bits)

    spin(step, args.spin_ms)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    assert torch.equal(bits, msg), "round trip failed: decrypted bits differ from the messages"

    # ---- timed region ------------------------------------------------------------------------
    # HIP events bracket the GEMM and the decode on every EV_EVERY-th step of the timed region (an
    # event record drains the queue between two kernels: on every step it costs ~25 us of a 290 us
    # step -- measured with tools/time_overlap.py -- so the kernels are sampled, not all stamped)
    EV_EVERY = 8 if args.steps >= 24 else 1
    ev = {k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
              torch.cuda.Event(enable_timing=True)) for k in range(0, args.steps, EV_EVERY)}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        e = ev.get(k)
        if e:
            e[0].record()
        ctx.matmul(key, xbits, X_BITS, B, msg=msg, out=cipher)
        if e:
            e[1].record()
        ctx.dot_decode(w, q, cipher, out=bits)
        if e:
            e[2].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.equal(bits, msg)

    gemm_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev.values()]))
    dec_ms = float(np.mean([b.elapsed_time(c) for _, b, c in ev.values()]))
    value = world * B * args.steps / elapsed
    achieved_tops = ops_per_enc * B / (gemm_ms * 1e-3) / 1e12

    out = {
        "metric": "encryptions/sec", "value": value, "unit": "encryptions/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8", "dtype_note": "signed radix-2^8 digits on int8 MFMA, exact i32/i64 accumulation, canonical Z_q results",
        "data": "synthetic (seeded SplitMix64 on device)",
        "config": {"workload": "Regev enc+dec round trip (BASELINE config 2)", "n": n, "m": m, "q": q,
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"batch-shard x{world}",
                   "key_broadcast_ms": t_bcast * 1e3},
        "roofline": {"kernel": "qf_gemm_packed_kernel<3,3> (bit operands, packed remainder row block)", "bound": "mfma", "achieved": achieved_tops,
                     "peak": INT8_PEAK_TOPS, "unit": "TFLOP/s", "frac": achieved_tops / INT8_PEAK_TOPS,
                     "traffic": pmc_traffic("qf_gemm_") if B == 65536 else None, "launch_ms": gemm_ms,
                     "note": "int8 multiply-accumulates counted as 2 ops; algorithmic ops = 2*La*Lb*(n+1)*m per encryption; "
                             "traffic = HBM bytes per launch from the committed PMC passes (profiles/r1/traffic.json); "
                             "sustained bare-MFMA rate on this data (tools/ubench_mfma): 3.4-3.65 POP/s at the 1.7-1.8 GHz the chip holds"},
        "decode": {"kernel": "qf_dot_decode_kernel", "bound": "hbm", "launch_ms": dec_ms,
                   "achieved": B * (n + 1) * 4 / (dec_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": B * (n + 1) * 4 / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }

    # ---- secondary headline: ring products n=1024 (BASELINE config 5 (i)) --------------------
    if not args.no_ring:
        nr, qr, Br = 1024, 1073479681, 1 << 16
        ring = ctx.ring(nr, qr)
        a = ctx.sample_uniform(SEED, 6, rank * 2 * Br * nr, Br * nr, qr).view(Br, nr)
        b = ctx.sample_uniform(SEED, 6, (rank * 2 + 1) * Br * nr, Br * nr, qr).view(Br, nr)
        c = torch.empty_like(a)
        spin(lambda: ring.polymul(a, b, out=c), args.spin_ms)
        torch.cuda.synchronize()
        rs = max(3, args.steps // 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if world > 1:
            dist.barrier()
        e0.record()
        for _ in range(rs):
            ring.polymul(a, b, out=c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / rs
        rate = Br / (ms * 1e-3)
        if world > 1:
            tt = torch.tensor([rate], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            rate = float(tt.item())
        gbs = Br * 3 * 4 * nr / (ms * 1e-3) / 1e9
        out["ring_polymul"] = {"metric": "ring-polymuls/sec (n=1024, q=2^30-2^18+1)", "value": rate, "unit": "polymuls/s",
                               "batch_per_gpu": Br, "launch_ms": ms,
                               "roofline": {"kernel": "qf_ring_wave_kernel<10,polymul,lazy>", "bound": "hbm", "achieved": gbs,
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                            "traffic": pmc_traffic("qf_ring_wave_kernel<10, 0") if Br == 65536 else None,
                                            "note": "algorithmic bytes = 12,288 per product (2 in, 1 out)"}}

    # ---- CPU baseline (rank 0, single-GPU run only) -----------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import coracle as co
        pk_h = ctx.to_host_u32(pk)
        s_h = ctx.to_host_u32(s)
        threads = co.num_threads()
        probe = 64
        xb_h = ctx.to_host_u8(xbits[:probe])
        t1 = time.perf_counter()
        co.matmul_batch(pk_h, q, xb_h, co.X_BITS, probe)
        per_msg_1t = (time.perf_counter() - t1) / probe
        sample = int(min(B, max(threads * 16, args.cpu_seconds / per_msg_1t * threads * 0.5)))
        xb_h = ctx.to_host_u8(xbits[:sample])
        msg_h = ctx.to_host_u8(msg[:sample])
        t1 = time.perf_counter()
        c_h = co.matmul_batch(pk_h, q, xb_h, co.X_BITS, sample, msg=msg_h, mt=True)
        b_h = co.dot_decode_batch(s_h, q, c_h, mt=True)
        cpu_t = time.perf_counter() - t1
        assert np.array_equal(b_h, msg_h)
        assert np.array_equal(c_h, ctx.to_host_u32(cipher[:sample])), "GPU ciphertexts differ from the CPU oracle"
        out["cpu_baseline"] = {"value": sample / cpu_t, "unit": "encryptions/s", "cores": threads, "kind": "port",
                               "sample": f"{sample} of the same {B} messages (enc+dec, one message per call as the reference does, "
                                         f"OpenMP over messages); single-thread rate {1.0 / per_msg_1t:.1f}/s on a {probe}-message probe; "
                                         "GPU ciphertexts of the sample verified bit-identical"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        if os.environ.get("QF_BENCH_REHEARSAL") == "1":
            out["rehearsal"] = "all ranks on cuda:0, collectives over gloo: not a measurement"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


SPIN_MS = 60.0


def _time_loop(fn, steps, warmup, dist, world):
    spin(fn, SPIN_MS)
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return el, float(np.mean([a.elapsed_time(b) for a, b in evs]))


def other_workloads(args):
    """Configs 3-5 of BASELINE.json (not the headline line): same conventions, one JSON line."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local, dist = _init_dist(world)
    from crypto_amd.device import Context, X_BITS, X_U32
    from crypto_amd import sharding
    ctx = Context(local)
    wl = args.workload

    def replicated(count, q_, shape):
        """A public key / SIS matrix: generated on rank 0, replicated with ONE RCCL broadcast."""
        t = ctx.empty_u32(*shape)
        if rank == 0:
            ctx.sample_uniform(SEED, 1, 0, count, q_, out=t.view(-1))
        if world > 1:
            dist.broadcast(t, src=0)
        return t

    out = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
           "vs_baseline": None, "data": "synthetic (seeded SplitMix64 on device)"}
    if wl == "sis":
        n, q = 512, 8380417
        m = n * 23
        Btot = 1 << 20
        lo, hi = sharding.shard_range(Btot, rank, world)
        B = hi - lo
        A = replicated(n * m, q, (n, m))
        key = ctx.key_upload(A, q)
        xb = ctx.sample_bits(SEED, 3, lo, B, m)
        dig = ctx.empty_u32(B, n)
        el, ms = _time_loop(lambda: ctx.matmul(key, xb, X_BITS, B, out=dig), args.steps, args.warmup, dist, world)
        ops = 2 * 3 * n * m
        tops = ops * B / (ms * 1e-3) / 1e12
        out.update({"metric": "SIS hashes/sec", "value": Btot * args.steps / el, "unit": "hashes/s", "scaling": "strong",
                    "ms_per_step": el / args.steps * 1e3, "dtype": "int8 digits (exact Z_q)",
                    "config": {"workload": "SIS hash, binary inputs (BASELINE config 3)", "n": n, "m": m, "q": q, "global_batch": Btot},
                    "roofline": {"kernel": "qf_gemm_kernel<3,1,*,bits>", "bound": "mfma", "achieved": tops, "peak": INT8_PEAK_TOPS,
                                 "unit": "TFLOP/s", "frac": tops / INT8_PEAK_TOPS, "traffic": None, "launch_ms": ms}})
        # secondary: inputs uniform in Z_q (hash/sis.rs:46), B = 2^16: recode to 3 digits + GEMM <3,3>
        B2 = min(B, 1 << 16)
        v = ctx.sample_uniform(SEED, 6, lo * m, B2 * m, q).view(B2, m)
        dig2 = ctx.empty_u32(B2, n)
        _, ms2 = _time_loop(lambda: ctx.matmul(key, v, X_U32, B2, out=dig2), max(3, args.steps // 10), 2, dist, world)
        out["full_width_inputs"] = {"metric": "SIS hashes/sec, inputs uniform in Z_q (recode + GEMM, 3x3 digit products)",
                                    "value": B2 / (ms2 * 1e-3), "unit": "hashes/s", "batch_per_gpu": B2, "call_ms": ms2,
                                    "achieved_tops": 2 * 9 * n * m * B2 / (ms2 * 1e-3) / 1e12}
        # linearity property at full size: H(x) + H(y) == H(x xor y) + 2*H(x and y)  (mod q)
        if rank == 0:
            k = min(B, 4096)
            x, y = xb[:k], ctx.sample_bits(SEED, 7, 0, k, m)
            hx = ctx.matmul(key, x, X_BITS, k).to(torch.int64)
            hy = ctx.matmul(key, y, X_BITS, k).to(torch.int64)
            hxor = ctx.matmul(key, x ^ y, X_BITS, k).to(torch.int64)
            hand = ctx.matmul(key, x & y, X_BITS, k).to(torch.int64)
            assert torch.equal((hx + hy) % q, (hxor + 2 * hand) % q), "SIS linearity violated"
    elif wl == "lpr":
        n, q, W = 1024, 1073741789, 2319
        Btot = 1 << 18
        lo, hi = sharding.shard_range(Btot, rank, world)
        B = hi - lo
        pk = replicated((n + 1) * n, q, (n + 1, n))
        key = ctx.key_upload(pk, q)
        r = ctx.sample_noise(SEED, 3, lo * n, B * n, q, W).view(B, n)
        e = ctx.sample_noise(SEED, 4, lo * (n + 1), B * (n + 1), q, W).view(B, n + 1)
        msg = (ctx.sample_uniform(SEED, 5, lo, B, 2) & 1).to(torch.uint8)
        c = ctx.empty_u32(B, n + 1)
        ctx.set_timing(False)
        el, ms = _time_loop(lambda: ctx.matmul(key, r, X_U32, B, addend=e, msg=msg, x_digits=2, out=c), args.steps, args.warmup, dist, world)
        ctx.set_timing(True)
        ctx.matmul(key, r, X_U32, B, addend=e, msg=msg, x_digits=2, out=c)
        gemm_ms = ctx.last_kernel_ms("gemm")
        ctx.set_timing(False)
        ops = 2 * 4 * 2 * (n + 1) * n
        tops = ops * B / (gemm_ms * 1e-3) / 1e12
        gbs = B * (4 * n + 8 * (n + 1)) / (ms * 1e-3) / 1e9
        out.update({"metric": "LPR encryptions/sec", "value": Btot * args.steps / el, "unit": "encryptions/s", "scaling": "strong",
                    "ms_per_step": el / args.steps * 1e3, "dtype": "int8 digits (exact Z_q)",
                    "config": {"workload": "LPR enc (BASELINE config 4), r recoded to 2 signed digits (verified on device)",
                               "n": n, "q": q, "global_batch": Btot},
                    "roofline": {"kernel": "qf_gemm_kernel<4,1,3,planes> (2 operand digits folded into the inner dimension)", "bound": "mfma", "achieved": tops, "peak": INT8_PEAK_TOPS,
                                 "unit": "TFLOP/s", "frac": tops / INT8_PEAK_TOPS, "traffic": None, "launch_ms": gemm_ms,
                                 "whole_call_ms": ms, "hbm_frac_whole_call": gbs / HBM_PEAK_GBS}})
    elif wl in ("ringlpr", "fdh"):
        n, q, W = 1024, 1073479681, 36
        ring = ctx.ring(n, q)
        if wl == "ringlpr":
            Btot = 1 << 18
            lo, hi = sharding.shard_range(Btot, rank, world)
            B = hi - lo
            keys = replicated(2 * n, q, (2, n))
            pk = ring.key(keys)
            r = ctx.sample_noise(SEED, 3, lo * n, B * n, q, W).view(B, n)
            e1 = ctx.sample_noise(SEED, 4, lo * n, B * n, q, W).view(B, n)
            e2 = ctx.sample_noise(SEED, 8, lo * n, B * n, q, W).view(B, n)
            mu = ctx.sample_bits(SEED, 5, lo, B, n, n // 8)
            u, v = torch.empty_like(r), torch.empty_like(r)
            el, ms = _time_loop(lambda: ring.ringlpr_enc(pk, r, e1, e2, mu, u=u, v=v), args.steps, args.warmup, dist, world)
            bytes_per = 5 * 4 * n + n // 8
            out.update({"metric": "RingLPR encryptions/sec (= 2 ring products each)", "value": Btot * args.steps / el, "unit": "encryptions/s"})
        else:
            mbar = 32
            Btot = 1 << 15
            lo, hi = sharding.shard_range(Btot, rank, world)
            B = hi - lo
            a = ring.key(replicated(mbar * n, q, (mbar, n)))
            sig = (ctx.sample_uniform(SEED, 7, lo * mbar * n, B * mbar * n, 201) - 100).view(B, mbar, n)
            uu = ctx.sample_uniform(SEED, 6, lo * n, B * n, q).view(B, n)
            el, ms = _time_loop(lambda: ring.fdh_verify(a, sig, uu, 10 ** 12), args.steps, args.warmup, dist, world)
            bytes_per = (mbar + 1) * 4 * n
            # the caller step in front of it: SHA-256 hash-to-R_q of B 32-byte messages on the device
            msgs = [("message %08d of the FDH verification batch" % (lo + i)) for i in range(B)]
            packed = ctx.pack_messages(msgs)
            hout = ctx.empty_u32(B, n)
            _, hms = _time_loop(lambda: ctx.hash_packed(packed, n, 1, q, out=hout), max(3, args.steps // 2), 1, dist, world)
            out["hash_to_ring"] = {"metric": "SHA-256 hash-to-R_q on the device (n = 1024 digests of a 43-byte message each)",
                                   "value": B / (hms * 1e-3), "unit": "messages/s", "launch_ms": hms,
                                   "digests_per_s": B * n / (hms * 1e-3)}
            out.update({"metric": "GPV-ring FDH verifications/sec (k+2 = 32 ring products each)", "value": Btot * args.steps / el, "unit": "verifications/s"})
        gbs = B * bytes_per / (ms * 1e-3) / 1e9
        out.update({"scaling": "strong", "ms_per_step": el / args.steps * 1e3, "dtype": "u32 (exact Z_q, Shoup NTT)",
                    "config": {"workload": f"{wl} n=1024 (BASELINE config 5)", "n": n, "q": q, "global_batch": Btot},
                    "roofline": {"kernel": "qf_ring_wave_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "launch_ms": ms}})
    elif wl == "regev_cycle":
        # BASELINE config 1 = the reference's own criterion bench (benches/regev.rs:15-57): one
        # gen + enc + dec cycle of ONE bit at new_from_n(n), n = 50 and the sweep 10..60.  The CPU
        # port of the reference's call structure beside the same cycle through the GPU host layer
        # (single-message path: kernel (c) + decode, key upload included -- as the reference
        # regenerates the key inside its timed loop).  A CPU reference point, no roofline.
        from oracle import coracle as co
        from crypto_amd.construction.pk_encryption import Regev
        rows = []
        for nn in (10, 20, 30, 40, 50, 60):
            r = Regev.new_from_n(nn, ctx=ctx)
            W = r.noise_width
            reps = 20
            t1 = time.perf_counter()
            for i in range(reps):
                assert co.regev_cycle(r.n, r.m, r.q, W, SEED + i, 1) == 1
            cpu_ms = (time.perf_counter() - t1) / reps * 1e3
            pk, sk = r.gen(); r.dec(sk, r.enc(pk, 1))
            t1 = time.perf_counter()
            for i in range(reps):
                pk, sk = r.gen()
                assert r.dec(sk, r.enc(pk, 1)) == 1
            gpu_ms = (time.perf_counter() - t1) / reps * 1e3
            rows.append({"n": r.n, "m": r.m, "q": r.q, "cpu_port_ms_per_cycle": cpu_ms, "gpu_host_layer_ms_per_cycle": gpu_ms})
        n50 = [x for x in rows if x["n"] == 50][0]
        out.update({"metric": "Regev gen+enc+dec cycles/sec, one bit, n=50 (benches/regev.rs default)", "value": 1e3 / n50["cpu_port_ms_per_cycle"],
                    "unit": "cycles/s (CPU port of the reference's call structure, 1 core)", "scaling": "weak", "ms_per_step": n50["cpu_port_ms_per_cycle"],
                    "dtype": "u32/u128 (exact Z_q)", "config": {"workload": "Regev single-bit cycle sweep n=10..60 (BASELINE config 1)"},
                    "sweep": rows, "roofline": None,
                    "note": "single-message calls are launch/PCIe-latency bound on a GPU; the GPU is for the batched forms (configs 2-5)"})
    elif wl == "matvec":
        # single-message SIS hash, matrix cache-cold: rotate over > 256 MiB of distinct keys
        n, q = 512, 8380417
        m = n * 23
        nkeys = 14
        keys = [ctx.key_upload(ctx.sample_uniform(SEED, 1, i * n * m, n * m, q).view(n, m), q) for i in range(nkeys)]
        x = ctx.sample_uniform(SEED, 6, 0, m, 2)
        y = ctx.empty_u32(n)
        state = {"i": 0}

        def one():
            k = keys[state["i"] % nkeys]
            state["i"] += 1
            assert ctx.lib.qf_matvec_modq_dev(ctx.h, k.h, x.data_ptr(), y.data_ptr()) == 0
        el, ms = _time_loop(one, max(args.steps, nkeys * 2), nkeys, dist, world)
        gbs = 4 * n * m / (ms * 1e-3) / 1e9
        out.update({"metric": "single-message SIS hashes/sec (HBM-streaming mat-vec, cache-cold keys)",
                    "value": world * max(args.steps, nkeys * 2) / el, "unit": "hashes/s", "scaling": "weak",
                    "ms_per_step": ms, "dtype": "u32 (exact Z_q)",
                    "config": {"workload": "SIS mat-vec n=512 m=11776, 14 keys = 338 MB rotated", "n": n, "m": m, "q": q},
                    "roofline": {"kernel": "qf_matvec_rows", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "launch_ms": ms}})
    if rank == 0:
        if os.environ.get("QF_BENCH_REHEARSAL") == "1":
            out["rehearsal"] = "all ranks on cuda:0, collectives over gloo: not a measurement"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    _a = parse()
    SPIN_MS = _a.spin_ms
    if _a.workload == "regev":
        main()
    else:
        other_workloads(_a)
