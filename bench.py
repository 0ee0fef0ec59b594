#!/usr/bin/env python3
"""Benchmarks of the hot path through the C ABI on MI355X.  Headline (BASELINE.json config 2):
batched Regev enc+dec, n=256, m=6168, q=16,777,213, 65,536 one-bit messages per GPU.

  python bench.py --gpus N --steps K --warmup W [--workload regev|sis|lpr|ringlpr|fdh|matvec|dualregev|regev_cycle]

With no --workload the ONE JSON line is the headline, and -- after the headline's own timed region,
in the same process -- every other configuration is run at BASELINE's size on the same clock and
appended as a sub-object of that line (`sis`, `lpr`, `ringlpr`, `fdh_verify`, `matvec`, `dual_regev`;
`ring_polymul` as before), each with its `value`, `ms_per_step`, `roofline` and a bounded
`cpu_baseline`.  The headline keys do not move.  --headline-only skips them; --workload X prints that
configuration's object as a line of its own (what the per-workload rocprofv3 passes under
profiles/r3/<X>/ run).

N > 1: when WORLD_SIZE is not set this process only LAUNCHES N fresh rank processes (one per GPU,
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment, rendezvous on 127.0.0.1) before anything
touches the GPU, relays rank 0's JSON line and exits with the ranks' status; under an external
launcher (python -m torch.distributed.run ... bench.py --gpus N) it is one of the ranks.  Ranks talk
over RCCL (backend "nccl"): ONE broadcast of the public key / SIS matrix / ring key from rank 0,
then every rank works on its own shard of the batch -- no collective on the data path.

A step = one pass of the hot path over one batch: encrypt B messages (kernel a, int8-MFMA mod-q
GEMM) and decrypt B ciphertexts (batched dot-decode).  Inputs are generated on the device by the
seeded generator and are resident in HBM when the timed region starts.  The batch buffers rotate
over R sets (x bits, messages, ciphertexts: 118 MB each, R = 6) and a step decrypts the set that was
encrypted R-1 steps earlier, so the decode streams its 67 MB from HBM, not from the 256 MiB
Infinity Cache where the GEMM just left them.

The timed region (EXACTLY K steps between barrier + synchronize on both sides) contains no event
records besides one at each end.  Per-kernel durations come from a second, untimed pass of the same
step loop in which sampled launches (every 4th step, the kernels taking turns) are bracketed by HIP
events on the launch stream; `roofline` divides the algorithmic work of the dominant kernel by that duration.  `cpu_baseline` is the C restatement of
the reference (oracle/, one message per call) on the host cores, on a bounded sample (rank 0, N=1).

Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5146414C4C
PROFILE_DIR = os.path.join(ROOT, "profiles", "r3")
INT8_PEAK_TOPS = 5033.0   # 2048 int8 op/clk/SIMD x 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md "Matrix cores": I8 = 2x BF16)
HBM_PEAK_GBS = 8000.0
WORKLOADS = ["regev", "sis", "lpr", "ringlpr", "fdh", "matvec", "dualregev", "regev_cycle"]
SUBS = [("sis", "sis"), ("lpr", "lpr"), ("ringlpr", "ringlpr"), ("fdh", "fdh_verify"), ("matvec", "matvec"), ("dualregev", "dual_regev")]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spin-ms", type=float, default=60.0,
                    help="untimed: keep the GPU busy this long before the W warmup steps (the clock the chip "
                         "holds under load settles ~20 ms after idle)")
    ap.add_argument("--batch", type=int, default=65536, help="messages per GPU per step (headline)")
    ap.add_argument("--sets", type=int, default=6, help="rotating batch buffer sets (headline): 6 x 67 MB of ciphertexts "
                                                        "exceed the 256 MiB Infinity Cache also when the decode runs alone")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-ring", action="store_true", help="skip the secondary ring-polymul measurement")
    ap.add_argument("--no-check", action="store_true", help="skip the small property-check launches (profile runs: keeps "
                                                             "the kernel-trace averages to full-size launches only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--bits-layout", choices=["kt", "rows"], default="kt",
                    help="resident form of the bit operands (Regev x, SIS inputs): k-tile-major planes (QF_X_BITS_KT) or row-major rows")
    ap.add_argument("--headline-only", action="store_true", help="do not append the other configurations to the headline line")
    ap.add_argument("--sub-steps", type=int, default=20, help="timed steps of each appended configuration (their steps take 1.6-3 ms)")
    ap.add_argument("--sub-cpu-seconds", type=float, default=8.0, help="CPU baseline budget of each appended configuration")
    ap.add_argument("--workload", default="regev", choices=WORKLOADS,
                    help="regev = the headline (BASELINE config 2); the others are the remaining configs, "
                         "each printing its own JSON line")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N with no WORLD_SIZE in the environment
# ---------------------------------------------------------------------------------------------
def launch_ranks(args) -> int:
    """Start N rank processes of this same script.  Nothing here touches the GPU: counting devices does
    not initialise HIP, and the ranks are fresh children (never an exec of a GPU-initialised process)."""
    import torch
    n = args.gpus
    rehearsal = os.environ.get("QF_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    if n > ndev and not rehearsal:
        sys.stderr.write(f"bench.py: --gpus {n} but this node exposes {ndev} GPU(s); refusing to report a "
                         f"{n}-GPU figure from fewer devices (QF_BENCH_REHEARSAL=1 rehearses the code path on one GPU over gloo)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), QF_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # Watchdog: the ranks meet in collectives, so ONE rank that dies (before the rendezvous or later) would
    # leave the others waiting for the process group's timeout.  Poll all of them; the first non-zero exit
    # ends the run: the siblings (fresh children of this process, never an exec) are killed and that status
    # is returned, also when rank 0 itself had finished cleanly.  Ranks still running 120 s after the first
    # clean exit are taken to be stuck.
    rc, first_done = 0, None
    while True:
        states = [p.poll() for p in procs]
        bad = [st for st in states if st not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(st == 0 for st in states):
            break
        if first_done is None and any(st == 0 for st in states):
            first_done = time.time()
        if first_done is not None and time.time() - first_done > 120:
            sys.stderr.write("bench.py: a rank is still running 120 s after another finished; killing the run\n")
            rc = 1
            break
        time.sleep(0.1)
    for p in procs:
        if p.poll() is None:
            p.kill()
    for p in procs:
        p.wait()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------
# helpers shared by the rank processes
# ---------------------------------------------------------------------------------------------
def pmc_traffic(workload, kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC passes of THIS workload
    (tools/profile_round.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench,
    FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when no profile is committed."""
    try:
        t = json.load(open(os.path.join(PROFILE_DIR, workload, "traffic.json")))
    except (OSError, ValueError):
        return None
    for k, v in t.items():
        if k.startswith(kernel_prefix) and v.get("fetch_bytes_per_launch_x2") is not None and v.get("write_bytes_per_launch") is not None:
            return v["fetch_bytes_per_launch_x2"] + v["write_bytes_per_launch"]
    return None


class Run:
    """One rank: device, process group, context, and the timing conventions of this file."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        if args.gpus != self.world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch one rank per GPU "
                             "(python bench.py --gpus N does it itself)")
        self.rehearsal = os.environ.get("QF_BENCH_REHEARSAL") == "1"
        local = 0 if self.rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        self.local = local
        self.dist = None
        self.backend = None
        self.rank_ms = None
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.rehearsal:   # ONE-GPU rehearsal of the multi-process path: every rank on cuda:0, gloo
                dist.init_process_group("gloo")
            else:                # RCCL
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            self.backend = dist.get_backend()
        from crypto_amd.device import Context
        self.ctx = Context(local)
        self.dev = self.ctx.device

    # -- collectives used outside the data path
    def meta(self):
        """What the process group really is: world size and every rank's device, gathered over it."""
        torch = self.torch
        if self.world == 1:
            return {"rccl_ranks": 1, "backend": None, "devices": [torch.cuda.get_device_name(self.local) + f" cuda:{self.local}"]}
        mine = torch.tensor([self.local, torch.cuda.get_device_properties(self.local).multi_processor_count],
                            dtype=torch.int64, device="cpu" if self.rehearsal else self.dev)
        allv = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allv, mine)
        return {"rccl_ranks": self.dist.get_world_size(), "backend": self.backend,
                "devices": [f"rank{r}: cuda:{int(v[0])} ({int(v[1])} CUs)" for r, v in enumerate(allv)]}

    def broadcast(self, *tensors):
        """The ONE collective of the path: rank 0's key material to every rank.  Returns seconds."""
        if self.world == 1:
            return 0.0
        torch = self.torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in tensors:
            if self.rehearsal:
                h = t.cpu()
                self.dist.broadcast(h, src=0)
                t.copy_(h)
            else:
                self.dist.broadcast(t, src=0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.rehearsal else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.rehearsal else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    # -- timing
    def spin(self, fn, ms=None):
        """Untimed: run fn back to back for about `ms` milliseconds (DVFS settle)."""
        ms = self.args.spin_ms if ms is None else ms
        if ms <= 0:
            return
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(10):
                fn()
            self.torch.cuda.synchronize()

    def timed(self, step, steps, warmup):
        """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over
        ranks.  One event at each end of the region (on the launch stream) cross-checks the wall clock.
        Returns (wall seconds, event milliseconds)."""
        torch = self.torch
        self.spin(step)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        self.barrier()
        el = time.perf_counter() - t0
        ems = e0.elapsed_time(e1)
        # every rank's own reading of the region (HIP events on its launch stream): min and max over ranks show stragglers
        self.rank_ms = (-self.max_over_ranks(-ems / steps), self.max_over_ranks(ems / steps)) if self.world > 1 else None
        return self.max_over_ranks(el), ems

    def sampled_ms(self, parts, steps, every=4):
        """Untimed second pass of the SAME step loop (all parts of a step, in order, `steps` times -- the clock
        the chip holds depends on the mix of kernels, so a kernel timed alone in a tight loop reads differently);
        on every `every`-th step ONE part is bracketed by two HIP events on the launch stream, the parts taking
        turns, and in one more turn the two events are recorded back to back with nothing between them: that
        reading is what the bracket itself costs (the command processor handles two markers around the launch),
        and it is subtracted.  Returns (mean ms per part, event-pair overhead ms)."""
        torch = self.torch
        n = len(parts)
        steps = max(steps, 16 * every * (n + 1))
        samples = [[] for _ in range(n + 1)]
        torch.cuda.synchronize()
        for k in range(steps):
            pick = (k // every) % (n + 1) if k % every == 0 else -1
            for j, p in enumerate(parts):
                if j == pick:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    p()
                    e1.record()
                    samples[j].append((e0, e1))
                else:
                    p()
            if pick == n:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e1.record()
                samples[n].append((e0, e1))
        torch.cuda.synchronize()
        raw = [sum(a.elapsed_time(b) for a, b in sm) / len(sm) for sm in samples]
        return [max(0.0, r - raw[n]) for r in raw[:n]], raw[n]

    def emit(self, out):
        if self.rank == 0:
            if self.rehearsal:
                out["rehearsal"] = "all ranks on cuda:0, collectives over gloo: not a measurement"
            print(json.dumps(out), flush=True)
        if self.world > 1:
            self.dist.destroy_process_group()


def cpu_sample_size(threads, per_unit_1t, seconds, cap, floor=None):
    floor = floor or threads * 4
    return int(min(cap, max(floor, seconds / per_unit_1t * threads * 0.5)))


# ---------------------------------------------------------------------------------------------
# headline: BASELINE config 2
# ---------------------------------------------------------------------------------------------
def headline(args):
    import numpy as np
    R_ = Run(args)
    torch, ctx, rank, world = R_.torch, R_.ctx, R_.rank, R_.world
    from crypto_amd.device import X_BITS, X_BITS_KT

    n, m, q = 256, 6168, 16777213
    B, NS = args.batch, max(2, args.sets)
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
    t_bcast = R_.broadcast(pk, s)
    key = ctx.key_upload(pk, q)
    w = ((q - (s.to(torch.int64) & 0xFFFFFFFF)) % q).to(torch.int32)

    # this rank's shard: rows [(rank*NS + i)*B, +B) of the global generator streams, for set i
    sets = []
    for i in range(NS):
        first = (rank * NS + i) * B
        sets.append({"x": ctx.sample_bits(SEED, 3, first, B, m),
                     "msg": (ctx.sample_uniform(SEED, 5, first, B, 2) & 1).to(torch.uint8),
                     "c": ctx.empty_u32(B, n + 1), "bits": ctx.empty_u8(B)})
        # the resident batch operand of the timed region: the same bits k-tile-major (QF_X_BITS_KT; layout of our boundary --
        # the reference has no packed form of x).  --bits-layout rows keeps the row-major rows.
        sets[i]["xk"] = ctx.bits_to_kt(sets[i]["x"], B, m) if args.bits_layout == "kt" else None
    state = {"k": 0}

    def enc(i, rows=False):
        if rows or args.bits_layout != "kt":
            ctx.matmul(key, sets[i]["x"], X_BITS, B, msg=sets[i]["msg"], out=sets[i]["c"])
        else:
            ctx.matmul(key, sets[i]["xk"], X_BITS_KT, B, msg=sets[i]["msg"], out=sets[i]["c"])

    def dec(i):
        ctx.dot_decode(w, q, sets[i]["c"], out=sets[i]["bits"])

    def step(rows=False):
        i = state["k"] % NS
        state["k"] += 1
        enc(i, rows)              # encrypt set i ...
        dec((i + 1) % NS)         # ... decrypt the set encrypted NS-1 steps ago (HBM-cold)

    def check_round_trip():
        torch.cuda.synchronize()
        for i in range(NS):
            dec(i)
        torch.cuda.synchronize()
        for i in range(NS):
            assert torch.equal(sets[i]["bits"], sets[i]["msg"]), "round trip failed: decrypted bits differ from the messages"

    for i in range(NS):
        enc(i)
    elapsed, region_ms = R_.timed(step, args.steps, args.warmup)
    hl_rank_ms = R_.rank_ms
    check_round_trip()
    value = world * B * args.steps / elapsed

    # per-kernel durations: second, untimed pass of the same step loop, sampled event pairs
    def p_enc():
        enc(state["k"] % NS)

    def p_dec():
        dec((state["k"] + 1) % NS)
        state["k"] += 1
    (gemm_ms, dec_ms), pair_ms = R_.sampled_ms([p_enc, p_dec], args.steps)
    achieved_tops = ops_per_enc * B / (gemm_ms * 1e-3) / 1e12
    dec_gbs = B * (n + 1) * 4 / (dec_ms * 1e-3) / 1e9

    # enc as the reference runs it: x is drawn inside the call (regev.rs:401) -- sampling on the device per step
    draw = {"k": 0}

    def step_sampling():
        i = state["k"] % NS
        draw["k"] += 1
        first = ((world + rank) * NS + draw["k"]) * B
        if args.bits_layout == "kt":
            ctx.sample_bits_kt(SEED, 3, first, B, m, out=sets[i]["xk"])   # the sampler writes the planes itself
            step()
        else:
            ctx.sample_bits(SEED, 3, first, B, m, out=sets[i]["x"])
            step(rows=True)
    k2 = max(5, args.steps // 2)
    el2, _ = R_.timed(step_sampling, k2, 2)
    check_round_trip()

    out = {
        "metric": "encryptions/sec", "value": value, "unit": "encryptions/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8",
        "dtype_note": "signed radix-2^8 digits on int8 MFMA, exact i32/i64 accumulation, canonical Z_q results",
        "data": "synthetic (seeded SplitMix64 on device)",
        "config": {"workload": "Regev enc+dec round trip (BASELINE config 2)", "n": n, "m": m, "q": q,
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"batch-shard x{world}",
                   "buffer_sets": NS, "key_broadcast_ms": t_bcast * 1e3,
                   "bits_layout": "k-tile-major (QF_X_BITS_KT)" if args.bits_layout == "kt" else "row-major (QF_X_BITS)"},
        "ms_per_step_events": region_ms / args.steps, "event_pair_overhead_ms": pair_ms,
        "roofline": {"kernel": "qf_gemm_packedx_kernel<3,4> (bit operands on v_mfma_i32_16x16x64_i8; two column tiles of 4 row blocks, the packed remainder block as a 13th fragment of the last)", "bound": "mfma",
                     "achieved": achieved_tops, "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": achieved_tops / INT8_PEAK_TOPS,
                     "traffic": pmc_traffic("regev", "qf_gemm_") if B == 65536 else None, "launch_ms": gemm_ms,
                     "note": "int8 multiply-accumulates counted as 2 ops; algorithmic ops = 2*La*Lb*(n+1)*m per encryption x B per launch; "
                             "launch_ms = mean of HIP-event pairs around sampled GEMM launches in a second, untimed pass of the same step loop, minus the reading of an empty event pair (event_pair_overhead_ms); "
                             "traffic = HBM bytes per launch from the committed PMC passes (profiles/r2/regev/traffic.json)"},
        "decode": {"kernel": "qf_dot_decode_kernel", "bound": "hbm", "launch_ms": dec_ms, "achieved": dec_gbs, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": dec_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("regev", "qf_dot_decode") if B == 65536 else None,
                   "note": f"ciphertexts written {NS - 1} steps earlier ({(NS - 1) * (B * (bit_stride_bytes(m) + 8 * (n + 1))) >> 20} MiB of traffic in between): HBM-cold"},
        "enc_with_sampling": {"metric": "encryptions/sec with x drawn on the device inside every step (regev.rs:401)",
                              "value": world * B * k2 / el2, "unit": "encryptions/s", "steps": k2, "ms_per_step": el2 / k2 * 1e3},
    }
    out.update(R_.meta())

    # ---- secondary headline: ring products n=1024 (BASELINE config 5 (i)) --------------------
    if not args.no_ring:
        nr, qr, Br = 1024, 1073479681, 1 << 16
        ring = ctx.ring(nr, qr)
        a = ctx.sample_uniform(SEED, 6, rank * 2 * Br * nr, Br * nr, qr).view(Br, nr)
        b = ctx.sample_uniform(SEED, 6, (rank * 2 + 1) * Br * nr, Br * nr, qr).view(Br, nr)
        c = torch.empty_like(a)
        rs = max(3, args.steps // 2)
        _, rms = R_.timed(lambda: ring.polymul(a, b, out=c), rs, 2)
        ms = rms / rs
        rate = R_.sum_over_ranks(Br / (ms * 1e-3))
        gbs = Br * 3 * 4 * nr / (ms * 1e-3) / 1e9
        out["ring_polymul"] = {"metric": "ring-polymuls/sec (n=1024, q=2^30-2^18+1)", "value": rate, "unit": "polymuls/s",
                               "batch_per_gpu": Br, "launch_ms": ms,
                               "roofline": {"kernel": "qf_ring_wave_kernel<10,polymul,lazy>", "bound": "hbm", "achieved": gbs,
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                            "traffic": pmc_traffic("regev", "qf_ring_wave_kernel<10, 0") if Br == 65536 else None,
                                            "note": "algorithmic bytes = 12,288 per product (2 in, 1 out); 805 MB per launch, HBM-cold by size"}}

    # ---- CPU baseline (rank 0, single-GPU run only) -----------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import coracle as co
        pk_h, s_h = ctx.to_host_u32(pk), ctx.to_host_u32(s)
        threads = co.num_threads()
        probe = 64
        xs, ms_, cs = sets[0]["x"], sets[0]["msg"], sets[0]["c"]
        ctx.sample_bits(SEED, 3, (rank * NS) * B, B, m, out=xs)    # set 0 as first generated
        if args.bits_layout == "kt":
            ctx.bits_to_kt(xs, B, m, out=sets[0]["xk"])
        enc(0)
        xb_h = ctx.to_host_u8(xs[:probe])
        t1 = time.perf_counter()
        co.matmul_batch(pk_h, q, xb_h, co.X_BITS, probe)
        per_msg_1t = (time.perf_counter() - t1) / probe
        sample = cpu_sample_size(threads, per_msg_1t, args.cpu_seconds, B, threads * 16)
        xb_h, msg_h = ctx.to_host_u8(xs[:sample]), ctx.to_host_u8(ms_[:sample])
        t1 = time.perf_counter()
        c_h = co.matmul_batch(pk_h, q, xb_h, co.X_BITS, sample, msg=msg_h, mt=True)
        b_h = co.dot_decode_batch(s_h, q, c_h, mt=True)
        cpu_t = time.perf_counter() - t1
        assert np.array_equal(b_h, msg_h)
        assert np.array_equal(c_h, ctx.to_host_u32(cs[:sample])), "GPU ciphertexts differ from the CPU oracle"
        out["cpu_baseline"] = {"value": sample / cpu_t, "unit": "encryptions/s", "cores": threads, "kind": "port",
                               "sample": f"{sample} of the same {B} messages (enc+dec, one message per call as the reference does, "
                                         f"OpenMP over messages); single-thread rate {1.0 / per_msg_1t:.1f}/s on a {probe}-message probe; "
                                         "GPU ciphertexts of the sample verified bit-identical"}
    elif rank == 0:
        out["cpu_baseline"] = None
    if R_.rank_ms is not None:
        out["ms_per_step_ranks"] = {"min": hl_rank_ms[0], "max": hl_rank_ms[1]}

    # ---- every other configuration on the same clock, as sub-objects of this line -------------
    if not args.headline_only:
        import types
        sets.clear()
        sub = types.SimpleNamespace(**vars(args))
        sub.steps, sub.warmup, sub.cpu_seconds = max(3, min(args.steps, args.sub_steps)), min(args.warmup, 3), args.sub_cpu_seconds
        for wl, name in SUBS:
            torch.cuda.empty_cache()
            t1 = time.perf_counter()
            o = workload(R_, wl, sub)
            o["wall_s"] = time.perf_counter() - t1
            out[name] = o
    R_.emit(out)


def bit_stride_bytes(cols):
    return ((cols + 127) // 128) * 16


# ---------------------------------------------------------------------------------------------
# the other configs
# ---------------------------------------------------------------------------------------------
def other_workloads(args):
    """--workload X: that configuration's object as the one JSON line."""
    R_ = Run(args)
    out = workload(R_, args.workload, args)
    out.update(R_.meta())
    R_.emit(out)


def workload(R_, wl, args):
    """Configs 1, 3, 4, 5 of BASELINE.json, the single-message mat-vec and DualRegev at secure128: same conventions
    as the headline; returns the object (a line of its own under --workload, a sub-object of the headline line otherwise)."""
    import numpy as np
    torch, ctx, rank, world = R_.torch, R_.ctx, R_.rank, R_.world
    from crypto_amd.device import X_BITS, X_U32
    from crypto_amd import sharding
    want_cpu = rank == 0 and world == 1 and not args.no_cpu
    co = None
    if want_cpu or wl == "regev_cycle":
        from oracle import coracle as co
    bcast = {"s": 0.0}

    def replicated(count, q_, shape):
        """A public key / SIS matrix / ring key: generated on rank 0, replicated with ONE broadcast."""
        t = ctx.empty_u32(*shape)
        if rank == 0:
            ctx.sample_uniform(SEED, 1, 0, count, q_, out=t.view(-1))
        bcast["s"] += R_.broadcast(t)
        return t

    out = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
           "vs_baseline": None, "data": "synthetic (seeded SplitMix64 on device)", "cpu_baseline": None}
    if wl == "sis":
        n, q = 512, 8380417
        m = n * 23
        Btot = 1 << 20
        lo, hi = sharding.shard_range(Btot, rank, world)
        B = hi - lo
        A = replicated(n * m, q, (n, m))
        key = ctx.key_upload(A, q)
        xb = ctx.sample_bits(SEED, 3, lo, B, m)
        xk = ctx.bits_to_kt(xb, B, m) if args.bits_layout == "kt" else None
        dig = ctx.empty_u32(B, n)
        if xk is not None:
            from crypto_amd.device import X_BITS_KT
            el, ems = R_.timed(lambda: ctx.matmul(key, xk, X_BITS_KT, B, out=dig), args.steps, args.warmup)
        else:
            el, ems = R_.timed(lambda: ctx.matmul(key, xb, X_BITS, B, out=dig), args.steps, args.warmup)
        ms = ems / args.steps
        ops = 2 * 3 * n * m
        tops = ops * B / (ms * 1e-3) / 1e12
        out.update({"metric": "SIS hashes/sec", "value": Btot * args.steps / el, "unit": "hashes/s", "scaling": "strong",
                    "ms_per_step": el / args.steps * 1e3, "dtype": "int8",
                    "config": {"workload": "SIS hash, binary inputs (BASELINE config 3)", "n": n, "m": m, "q": q, "global_batch": Btot,
                               "key_broadcast_ms": bcast["s"] * 1e3,
                               "bits_layout": "k-tile-major (QF_X_BITS_KT)" if xk is not None else "row-major (QF_X_BITS)"},
                    "roofline": {"kernel": "qf_gemm_kernel<3,1,4,1,bits> (v_mfma_i32_16x16x64_i8)", "bound": "mfma", "achieved": tops, "peak": INT8_PEAK_TOPS,
                                 "unit": "TOP/s", "frac": tops / INT8_PEAK_TOPS, "traffic": pmc_traffic("sis", "qf_gemm_kernel<3, 1, 4, 1, true>") if world == 1 else None,
                                 "launch_ms": ms, "note": "algorithmic int8 ops = 2*3*n*m per hash x B per launch; one launch per step"}})
        # secondary: inputs uniform in Z_q (hash/sis.rs:46), B = 2^16: recode to 3 digits + GEMM
        B2 = min(B, 1 << 16)
        v = ctx.sample_uniform(SEED, 6, lo * m, B2 * m, q).view(B2, m)
        dig2 = ctx.empty_u32(B2, n)
        k2 = max(3, args.steps // 10)
        _, ems2 = R_.timed(lambda: ctx.matmul(key, v, X_U32, B2, out=dig2), k2, 2)
        ms2 = ems2 / k2
        out["full_width_inputs"] = {"metric": "SIS hashes/sec, inputs uniform in Z_q (recode + GEMM, 3x3 digit products)",
                                    "value": B2 / (ms2 * 1e-3), "unit": "hashes/s", "batch_per_gpu": B2, "call_ms": ms2,
                                    "achieved_tops": 2 * 9 * n * m * B2 / (ms2 * 1e-3) / 1e12}
        # linearity property at full size: H(x) + H(y) == H(x xor y) + 2*H(x and y)  (mod q)
        if rank == 0 and not args.no_check:
            k = min(B, 4096)
            x, y = xb[:k], ctx.sample_bits(SEED, 7, 0, k, m)
            hx = ctx.matmul(key, x, X_BITS, k).to(torch.int64)
            hy = ctx.matmul(key, y, X_BITS, k).to(torch.int64)
            hxor = ctx.matmul(key, x ^ y, X_BITS, k).to(torch.int64)
            hand = ctx.matmul(key, x & y, X_BITS, k).to(torch.int64)
            assert torch.equal((hx + hy) % q, (hxor + 2 * hand) % q), "SIS linearity violated"
        if want_cpu:
            A_h = ctx.to_host_u32(A)
            threads = co.num_threads()
            probe = 16
            xb_h = ctx.to_host_u8(xb[:probe])
            t1 = time.perf_counter()
            co.matmul_batch(A_h, q, xb_h, co.X_BITS, probe)
            per1 = (time.perf_counter() - t1) / probe
            sample = cpu_sample_size(threads, per1, args.cpu_seconds, B)
            xb_h = ctx.to_host_u8(xb[:sample])
            t1 = time.perf_counter()
            d_h = co.matmul_batch(A_h, q, xb_h, co.X_BITS, sample, mt=True)
            cpu_t = time.perf_counter() - t1
            assert np.array_equal(d_h, ctx.to_host_u32(dig[:sample])), "GPU digests differ from the CPU oracle"
            out["cpu_baseline"] = {"value": sample / cpu_t, "unit": "hashes/s", "cores": threads, "kind": "port",
                                   "sample": f"{sample} of the {B} inputs (one mat-vec per call, OpenMP over inputs); single-thread "
                                             f"{1.0 / per1:.1f}/s on a {probe}-input probe; GPU digests of the sample verified bit-identical"}
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
        # the call as a host that hands over r as residues makes it: recode to 2 signed digits (verified) + product
        el_u, ems_u = R_.timed(lambda: ctx.matmul(key, r, X_U32, B, addend=e, msg=msg, x_digits=2, out=c), args.steps, args.warmup)
        c_u = c.clone()
        # the resident operand of the timed region: the same r as digit planes (QF_X_PLANES), recoded once, ahead of it
        from crypto_amd.device import X_PLANES
        rp = ctx.recode_planes(key, r, X_U32, B, 2)
        def enc():
            ctx.matmul(key, rp, X_PLANES, B, addend=e, msg=msg, x_digits=2, out=c)
        el, ems = R_.timed(enc, args.steps, args.warmup)
        assert torch.equal(c, c_u), "product on pre-recoded planes differs from the recode-inside call"
        del c_u
        ms = ems / args.steps
        # a step is ONE launch of the GEMM now: its mean duration is the HIP-event time of the timed region itself / steps
        # (a separate pass with an event pair and a host read-back per call leaves the GPU idle between launches and reads
        # 5-7 % longer: cold Infinity Cache, clock ramp)
        gemm_ms = ms
        ops = 2 * 4 * 2 * (n + 1) * n
        tops = ops * B / (gemm_ms * 1e-3) / 1e12
        gbs = B * (4 * n + 8 * (n + 1)) / (ms * 1e-3) / 1e9
        out.update({"metric": "LPR encryptions/sec", "value": Btot * args.steps / el, "unit": "encryptions/s", "scaling": "strong",
                    "ms_per_step": el / args.steps * 1e3, "dtype": "int8",
                    "config": {"workload": "LPR enc (BASELINE config 4), r resident as 2 signed-digit planes (QF_X_PLANES; recoded and verified on device ahead of the timed region)",
                               "n": n, "q": q, "global_batch": Btot, "key_broadcast_ms": bcast["s"] * 1e3},
                    "recode_inside": {"metric": "LPR encryptions/sec with r handed over as uint32 residues (recode + digit check inside every call)",
                                      "value": Btot * args.steps / el_u, "unit": "encryptions/s", "ms_per_step": el_u / args.steps * 1e3},
                    "roofline": {"kernel": "qf_gemm_kernel<4,1,3,1,planes> (2 operand digits folded into the inner dimension)", "bound": "mfma",
                                 "achieved": tops, "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": tops / INT8_PEAK_TOPS,
                                 "traffic": pmc_traffic("lpr", "qf_gemm_kernel") if world == 1 else None, "launch_ms": gemm_ms,
                                 "whole_call_ms": ms, "hbm_frac_whole_call": gbs / HBM_PEAK_GBS,
                                 "note": "algorithmic int8 ops = 2*4*2*(n+1)*n per encryption (SURVEY 8d's 2.52e7 assumes 3 operand digits)"}})
        # the decryption half of config 4: <[-s ; 1], c> + decision rule over the ciphertexts just written
        # (1.07 GB per 262,144: far beyond the Infinity Cache, so the read is HBM-cold by size)
        w_dec = ctx.sample_uniform(SEED, 2, 0, n, q)
        bits_dec = ctx.empty_u8(B)
        for _ in range(3):
            ctx.dot_decode(w_dec, q, c, out=bits_dec)
        ctx.set_timing(True)
        dms = []
        for _ in range(max(3, min(args.steps, 20))):
            ctx.dot_decode(w_dec, q, c, out=bits_dec)
            dms.append(ctx.last_kernel_ms("decode"))
        ctx.set_timing(False)
        dec_ms = float(np.mean(dms))
        dec_gbs = B * 4 * (n + 1) / (dec_ms * 1e-3) / 1e9
        out["decode"] = {"kernel": "qf_dot_decode_kernel (split sums: (n+1) q^2 > 2^64)", "bound": "hbm", "launch_ms": dec_ms, "achieved": dec_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dec_gbs / HBM_PEAK_GBS,
                         "decryptions_per_s": B * world / (dec_ms * 1e-3),
                         "enc_dec_round_trips_per_s": B * world / ((el / args.steps) + dec_ms * 1e-3),
                         "note": "lpr.rs:454-468 on the batch; algorithmic bytes = 4(n+1) per ciphertext"}
        if want_cpu:
            pk_h = ctx.to_host_u32(pk)
            threads = co.num_threads()
            probe = 16
            t1 = time.perf_counter()
            co.matmul_batch(pk_h, q, ctx.to_host_u32(r[:probe]), co.X_U32, probe, addend=ctx.to_host_u32(e[:probe]), msg=ctx.to_host_u8(msg[:probe]))
            per1 = (time.perf_counter() - t1) / probe
            sample = cpu_sample_size(threads, per1, args.cpu_seconds, B)
            r_h, e_h, m_h = ctx.to_host_u32(r[:sample]), ctx.to_host_u32(e[:sample]), ctx.to_host_u8(msg[:sample])
            t1 = time.perf_counter()
            c_h = co.matmul_batch(pk_h, q, r_h, co.X_U32, sample, addend=e_h, msg=m_h, mt=True)
            cpu_t = time.perf_counter() - t1
            assert np.array_equal(c_h, ctx.to_host_u32(c[:sample])), "GPU ciphertexts differ from the CPU oracle"
            out["cpu_baseline"] = {"value": sample / cpu_t, "unit": "encryptions/s", "cores": threads, "kind": "port",
                                   "sample": f"{sample} of the {B} messages (pk*r + e + msg per call, OpenMP over messages); single-thread "
                                             f"{1.0 / per1:.1f}/s on a {probe}-message probe; GPU ciphertexts of the sample verified bit-identical"}
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
            el, ems = R_.timed(lambda: ring.ringlpr_enc(pk, r, e1, e2, mu, u=u, v=v), args.steps, args.warmup)
            bytes_per = 5 * 4 * n + n // 8
            kname, kpre = "qf_ring_wave_kernel<10,rlpr_enc,lazy>", "qf_ring_wave_kernel<10, 1"
            out.update({"metric": "RingLPR encryptions/sec (= 2 ring products each)", "value": Btot * args.steps / el, "unit": "encryptions/s"})
            # the decryption half (ring_lpr.rs:491-507): v - s*u, threshold, bit packing -- one launch
            sk_ring = ring.key(ctx.sample_noise(SEED, 2, 0, n, q, W).view(1, n))
            mu_out = ctx.empty_u8(B, n // 8)
            for _ in range(3):
                ring.ringlpr_dec(sk_ring, u, v, out=mu_out)
            ctx.set_timing(True)
            dms = []
            for _ in range(max(3, min(args.steps, 20))):
                ring.ringlpr_dec(sk_ring, u, v, out=mu_out)
                dms.append(ctx.last_kernel_ms("ntt"))
            ctx.set_timing(False)
            dec_ms = float(np.mean(dms))
            dec_gbs = B * (8 * n + n // 8) / (dec_ms * 1e-3) / 1e9
            out["decrypt"] = {"kernel": "qf_ring_wave_kernel<10,rlpr_dec,lazy>", "bound": "hbm", "launch_ms": dec_ms, "achieved": dec_gbs,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dec_gbs / HBM_PEAK_GBS, "decryptions_per_s": B * world / (dec_ms * 1e-3),
                              "note": "algorithmic bytes = 8n + n/8 per ciphertext (u, v in; n message bits out)"}
            if want_cpu:
                k_h = ctx.to_host_u32(keys)
                threads = co.num_threads()
                probe = 4
                rp, e1p, e2p, mup = (ctx.to_host_u32(r[:probe]), ctx.to_host_u32(e1[:probe]), ctx.to_host_u32(e2[:probe]), ctx.to_host_u8(mu[:probe]))
                t1 = time.perf_counter()
                u_h, v_h = co.ringlpr_enc_batch(k_h[0], k_h[1], q, rp, e1p, e2p, mup)
                per_sb = (time.perf_counter() - t1) / probe
                assert np.array_equal(u_h, ctx.to_host_u32(u[:probe])) and np.array_equal(v_h, ctx.to_host_u32(v[:probe])), "GPU ciphertexts differ from the CPU oracle"
                t1 = time.perf_counter()
                co.polymul_batch(np.repeat(k_h[:1], 64, 0), ctx.to_host_u32(r[:64]), q, use_ntt=True)
                per1 = (time.perf_counter() - t1) / 64
                sample = cpu_sample_size(threads, 2 * per1, args.cpu_seconds, min(B, 65536))
                r_h = ctx.to_host_u32(r[:sample])
                ka, kb = np.repeat(k_h[:1], sample, 0), np.repeat(k_h[1:2], sample, 0)
                t1 = time.perf_counter()
                au = co.polymul_batch(ka, r_h, q, use_ntt=True, mt=True)
                bv = co.polymul_batch(kb, r_h, q, use_ntt=True, mt=True)
                cpu_t = time.perf_counter() - t1
                e1_h = ctx.to_host_u32(e1[:sample]).astype(np.uint64)
                assert np.array_equal(((au.astype(np.uint64) + e1_h) % q).astype(np.uint32), ctx.to_host_u32(u[:sample])), "GPU u differs from the CPU NTT oracle"
                del bv
                out["cpu_baseline"] = {"value": sample / cpu_t, "unit": "encryptions/s", "cores": threads, "kind": "port",
                                       "sample": f"{sample} of the {B} messages: the two products a*r, b*r per encryption with the oracle's CPU NTT, OpenMP "
                                                 f"over messages (additions not timed); single-thread NTT {0.5 / per1:.1f} enc/s; schoolbook (the parity "
                                                 f"oracle, 1 thread) {1.0 / per_sb:.2f} enc/s on {probe} messages, (u, v) verified bit-identical"}
        else:
            mbar = 32
            Btot = 1 << 15
            lo, hi = sharding.shard_range(Btot, rank, world)
            B = hi - lo
            Akey = replicated(mbar * n, q, (mbar, n))
            a = ring.key(Akey)
            sig = (ctx.sample_uniform(SEED, 7, lo * mbar * n, B * mbar * n, 201) - 100).view(B, mbar, n)
            uu = ctx.sample_uniform(SEED, 6, lo * n, B * n, q).view(B, n)
            el, ems = R_.timed(lambda: ring.fdh_verify(a, sig, uu, 10 ** 12), args.steps, args.warmup)
            bytes_per = (mbar + 1) * 4 * n
            kname, kpre = "qf_ring_wave_kernel<10,fdh,lazy>", "qf_ring_wave_kernel<10, 3"
            # the caller step in front of it: SHA-256 hash-to-R_q of B 43-byte messages on the device
            msgs = [("message %08d of the FDH verification batch" % (lo + i)) for i in range(B)]
            packed = ctx.pack_messages(msgs)
            hout = ctx.empty_u32(B, n)
            kh = max(3, args.steps // 2)
            _, hems = R_.timed(lambda: ctx.hash_packed(packed, n, 1, q, out=hout), kh, 1)
            hms = hems / kh
            out["hash_to_ring"] = {"metric": "SHA-256 hash-to-R_q on the device (n = 1024 digests of a 43-byte message each)",
                                   "value": B / (hms * 1e-3), "unit": "messages/s", "launch_ms": hms,
                                   "digests_per_s": B * n / (hms * 1e-3)}
            out.update({"metric": "GPV-ring FDH verifications/sec (k+2 = 32 ring products each)", "value": Btot * args.steps / el, "unit": "verifications/s"})
            if want_cpu:
                A_h = ctx.to_host_u32(Akey)
                threads = co.num_threads()
                probe = 2
                sig_h, u_h = ctx.to_host_u32(sig[:probe]).view(np.int32), ctx.to_host_u32(uu[:probe])
                ok_g, t_g = ring.fdh_verify(a, sig[:probe].contiguous(), uu[:probe].contiguous(), 10 ** 12, want_t=True)
                t1 = time.perf_counter()
                t_h, ok_h = co.fdh_ring_verify_batch(A_h, q, sig_h, u_h, 10 ** 12)
                per_sb = (time.perf_counter() - t1) / probe
                assert np.array_equal(t_h, ctx.to_host_u32(t_g)) and np.array_equal(ok_h, ctx.to_host_u8(ok_g)), "GPU verification differs from the CPU oracle"
                sred = (ctx.to_host_u32(sig[:4]).view(np.int32).astype(np.int64) % q).astype(np.uint32).reshape(4 * mbar, n)
                t1 = time.perf_counter()
                co.polymul_batch(np.tile(A_h, (4, 1)), sred, q, use_ntt=True)
                per1 = (time.perf_counter() - t1) / 4
                sample = cpu_sample_size(threads, per1, args.cpu_seconds, min(B, 4096), threads)
                sred = (ctx.to_host_u32(sig[:sample]).view(np.int32).astype(np.int64) % q).astype(np.uint32).reshape(sample * mbar, n)
                Arep = np.tile(A_h, (sample, 1))
                t1 = time.perf_counter()
                prod = co.polymul_batch(Arep, sred, q, use_ntt=True, mt=True)
                cpu_t = time.perf_counter() - t1
                tsum = (prod.reshape(sample, mbar, n).astype(np.uint64).sum(axis=1) % q).astype(np.uint32)
                _, t_all = ring.fdh_verify(a, sig[:sample].contiguous(), uu[:sample].contiguous(), 10 ** 12, want_t=True)
                assert np.array_equal(tsum, ctx.to_host_u32(t_all)), "GPU f_a differs from the CPU NTT oracle"
                out["cpu_baseline"] = {"value": sample / cpu_t, "unit": "verifications/s", "cores": threads, "kind": "port",
                                       "sample": f"{sample} of the {B} signatures: the 32 products of f_a per signature with the oracle's CPU NTT, OpenMP over "
                                                 f"products (sums, norm and comparison not timed); single-thread NTT {1.0 / per1:.2f} verifications/s; schoolbook "
                                                 f"(the parity oracle, 1 thread) {1.0 / per_sb:.3f}/s on {probe} signatures; t verified bit-identical on the sample"}
        ms = ems / args.steps
        gbs = B * bytes_per / (ms * 1e-3) / 1e9
        out.update({"scaling": "strong", "ms_per_step": el / args.steps * 1e3, "dtype": "u32",
                    "config": {"workload": f"{wl} n=1024 (BASELINE config 5)", "n": n, "q": q, "global_batch": Btot,
                               "key_broadcast_ms": bcast["s"] * 1e3},
                    "roofline": {"kernel": kname, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(wl, kpre) if world == 1 else None,
                                 "launch_ms": ms, "note": f"algorithmic bytes = {bytes_per} per unit x {B} per launch; one launch per step, exact Z_q (Shoup NTT)"}})
    elif wl == "dualregev":
        # DualRegev at the reference's secure128 set (dual_regev.rs:301-303): the batched shape is the opposite of every
        # other config -- the key is the transposed pk, (m+1) x n = 5,314 x 230, the batch operand s is uniform in Z_q
        # (3 digits), and every encryption reads m+1 noise words and writes m+1 ciphertext words: 43 KB of HBM traffic
        # against 2.2e7 int8 ops (507 op/B, below the chip's 629 op/B balance) -- nominally HBM-bound.
        n, m, q, W = 230, 5313, 7764299, 2951
        Btot = 1 << 16
        lo, hi = sharding.shard_range(Btot, rank, world)
        B = hi - lo
        pk = ctx.empty_u32(n, m + 1)
        xsk = ctx.sample_bits(SEED, 2, 0, 1, m)
        if rank == 0:                                                           # gen: u = A x (dual_regev.rs:354)
            A = ctx.sample_uniform(SEED, 1, 0, n * m, q).view(n, m)
            x_h = np.unpackbits(ctx.to_host_u8(xsk), axis=1, bitorder="little")[0, :m].astype(np.uint32)
            pk[:, :m] = A
            pk[:, m] = ctx.to_dev_u32(ctx.matvec(ctx.key_upload(A, q), x_h))
        bcast["s"] += R_.broadcast(pk)
        key = ctx.key_upload(pk, q, transpose=True)                             # rows m+1, cols n
        sv = ctx.sample_uniform(SEED, 3, lo * n, B * n, q).view(B, n)
        e = ctx.sample_noise(SEED, 4, lo * (m + 1), B * (m + 1), q, W).view(B, m + 1)
        msg = (ctx.sample_uniform(SEED, 5, lo, B, 2) & 1).to(torch.uint8)
        c = ctx.empty_u32(B, m + 1)
        ctx.set_timing(False)
        el, ems = R_.timed(lambda: ctx.matmul(key, sv, X_U32, B, addend=e, msg=msg, out=c), args.steps, args.warmup)
        ms = ems / args.steps
        gms = []
        ctx.set_timing(True)
        for _ in range(max(3, min(args.steps, 20))):
            ctx.matmul(key, sv, X_U32, B, addend=e, msg=msg, out=c)
            gms.append(ctx.last_kernel_ms("gemm"))
        ctx.set_timing(False)
        gemm_ms = float(np.mean(gms))
        ops = 2 * 3 * 3 * (m + 1) * n
        bytes_per = 4 * n + 8 * (m + 1)
        tops = ops * B / (gemm_ms * 1e-3) / 1e12
        gbs = bytes_per * B / (gemm_ms * 1e-3) / 1e9
        out.update({"metric": "DualRegev encryptions/sec (secure128)", "value": Btot * args.steps / el, "unit": "encryptions/s", "scaling": "strong",
                    "ms_per_step": el / args.steps * 1e3, "dtype": "int8",
                    "config": {"workload": "DualRegev enc, secure128 (dual_regev.rs:301-303): c^t = s^t [A|u] + e^t, s uniform in Z_q (3 digits)",
                               "n": n, "m": m, "q": q, "global_batch": Btot, "key_broadcast_ms": bcast["s"] * 1e3},
                    "roofline": {"kernel": "qf_gemm_shortk_kernel<3,2,1> (3 operand digits folded: K' = 768; key tile resident in LDS, unsynchronised waves)", "bound": "hbm",
                                 "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic("dualregev", "qf_gemm_shortk_kernel") if world == 1 else None, "launch_ms": gemm_ms,
                                 "whole_call_ms": ms, "mfma_tops": tops, "mfma_frac": tops / INT8_PEAK_TOPS,
                                 "note": "algorithmic bytes = 4n (s) + 4(m+1) (e) + 4(m+1) (c) = 43,432 per encryption; algorithmic int8 ops = "
                                         "2*3*3*(m+1)*n = 21,999,960 per encryption: 507 op/B, below the 629 op/B balance of the chip"}})
        # decryption (dual_regev.rs:437-450): c^t [-x ; 1] and the decision rule -- 21 KB per ciphertext, HBM-cold by size
        w_dec = ((q - (ctx.to_dev_u32(np.unpackbits(ctx.to_host_u8(xsk), axis=1, bitorder="little")[0, :m].astype(np.uint32)).to(torch.int64))) % q).to(torch.int32)
        bits_dec = ctx.empty_u8(B)
        for _ in range(3):
            ctx.dot_decode(w_dec, q, c, out=bits_dec)
        assert torch.equal(bits_dec, msg), "DualRegev round trip failed"
        ctx.set_timing(True)
        dms = []
        for _ in range(max(3, min(args.steps, 20))):
            ctx.dot_decode(w_dec, q, c, out=bits_dec)
            dms.append(ctx.last_kernel_ms("decode"))
        ctx.set_timing(False)
        dec_ms = float(np.mean(dms))
        dec_gbs = B * 4 * (m + 1) / (dec_ms * 1e-3) / 1e9
        out["decode"] = {"kernel": "qf_dot_decode_kernel", "bound": "hbm", "launch_ms": dec_ms, "achieved": dec_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dec_gbs / HBM_PEAK_GBS, "decryptions_per_s": B * world / (dec_ms * 1e-3),
                         "note": "algorithmic bytes = 4(m+1) per ciphertext; every one of the batch decrypts to its message"}
        if want_cpu:
            pkT = np.ascontiguousarray(ctx.to_host_u32(pk).T)
            threads = co.num_threads()
            probe = 8
            t1 = time.perf_counter()
            co.matmul_batch(pkT, q, ctx.to_host_u32(sv[:probe]), co.X_U32, probe, addend=ctx.to_host_u32(e[:probe]), msg=ctx.to_host_u8(msg[:probe]))
            per1 = (time.perf_counter() - t1) / probe
            sample = cpu_sample_size(threads, per1, args.cpu_seconds, B)
            s_h, e_h, m_h = ctx.to_host_u32(sv[:sample]), ctx.to_host_u32(e[:sample]), ctx.to_host_u8(msg[:sample])
            t1 = time.perf_counter()
            c_h = co.matmul_batch(pkT, q, s_h, co.X_U32, sample, addend=e_h, msg=m_h, mt=True)
            cpu_t = time.perf_counter() - t1
            assert np.array_equal(c_h, ctx.to_host_u32(c[:sample])), "GPU ciphertexts differ from the CPU oracle"
            out["cpu_baseline"] = {"value": sample / cpu_t, "unit": "encryptions/s", "cores": threads, "kind": "port",
                                   "sample": f"{sample} of the {B} messages (s^t*pk + e^t + msg per call, OpenMP over messages); single-thread "
                                             f"{1.0 / per1:.1f}/s on a {probe}-message probe; GPU ciphertexts of the sample verified bit-identical"}
    elif wl == "regev_cycle":
        # BASELINE config 1 = the reference's own criterion bench (benches/regev.rs:15-57): one
        # gen + enc + dec cycle of ONE bit at new_from_n(n), n = 50 and the sweep 10..60.  The CPU
        # port of the reference's call structure beside the same cycle through the GPU host layer
        # (single-message path: kernel (c) + decode, key upload included -- as the reference
        # regenerates the key inside its timed loop).  A CPU reference point, no roofline.
        from crypto_amd.construction.pk_encryption import Regev
        rows = []
        for nn in (10, 20, 30, 40, 50, 60):
            r = Regev.new_from_n(nn, ctx=ctx, seed=SEED, noise_model="irwin_hall")   # the noise model the CPU port draws
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
            pk, sk = r.gen()
            t1 = time.perf_counter()
            for i in range(reps):
                assert r.dec(sk, r.enc(pk, 1)) == 1
            encdec_ms = (time.perf_counter() - t1) / reps * 1e3
            rows.append({"n": r.n, "m": r.m, "q": r.q, "cpu_port_ms_per_cycle": cpu_ms, "gpu_host_layer_ms_per_cycle": gpu_ms,
                         "gpu_host_layer_ms_enc_dec_only": encdec_ms})
        n50 = [x for x in rows if x["n"] == 50][0]
        out.update({"metric": "Regev gen+enc+dec cycles/sec, one bit, n=50 (benches/regev.rs default)", "value": 1e3 / n50["cpu_port_ms_per_cycle"],
                    "unit": "cycles/s (CPU port of the reference's call structure, 1 core)", "scaling": "weak", "ms_per_step": n50["cpu_port_ms_per_cycle"],
                    "dtype": "u32", "config": {"workload": "Regev single-bit cycle sweep n=10..60 (BASELINE config 1)"},
                    "sweep": rows, "roofline": None,
                    "cpu_baseline": {"value": 1e3 / n50["cpu_port_ms_per_cycle"], "unit": "cycles/s", "cores": 1, "kind": "port",
                                     "sample": "20 cycles per n; this config IS the CPU reference point"},
                    "note": "single-message calls are launch/PCIe-latency bound on a GPU; the GPU is for the batched forms (configs 2-5)"})
    elif wl == "matvec":
        # single-message SIS hash, matrix cache-cold: rotate over 24 keys = 434 MB of digit planes
        n, q = 512, 8380417
        m = n * 23
        nkeys = 24
        keys = [ctx.key_upload(ctx.sample_uniform(SEED, 1, i * n * m, n * m, q).view(n, m), q) for i in range(nkeys)]
        x = ctx.sample_uniform(SEED, 6, 0, m, 2)
        y = ctx.empty_u32(n)
        state = {"i": 0}

        def one():
            k = keys[state["i"] % nkeys]
            state["i"] += 1
            assert ctx.lib.qf_matvec_modq_dev(ctx.h, k.h, x.data_ptr(), y.data_ptr()) == 0
        steps = max(args.steps, nkeys * 4)
        el, ems = R_.timed(one, steps, nkeys)
        ms = ems / steps
        alg = n * m * 3                      # n*m*ceil(bits(q)/8) = 18.1 MB per hash
        gbs = alg / (ms * 1e-3) / 1e9
        out.update({"metric": "single-message SIS hashes/sec (HBM-streaming mat-vec, cache-cold keys)",
                    "value": world * steps / el, "unit": "hashes/s", "scaling": "weak", "steps": steps,
                    "ms_per_step": el / steps * 1e3, "dtype": "int8",
                    "config": {"workload": "SIS mat-vec n=512 m=11776, 24 keys rotated (434 MB of int8 digit planes)", "n": n, "m": m, "q": q},
                    "roofline": {"kernel": "qf_matvec_planes_kernel<3,2>", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("matvec", "qf_matvec_planes") if world == 1 else None,
                                 "launch_ms": ms,
                                 "note": "algorithmic bytes = n*m*3 = 18,087,936 per hash (one launch); launch_ms = HIP events around the "
                                         "whole timed region / launches, i.e. it includes the ~1.5 us launch-to-launch gap of back-to-back calls"}})
        if want_cpu:
            A_h = ctx.to_host_u32(ctx.sample_uniform(SEED, 1, 0, n * m, q).view(n, m))
            x_h = ctx.to_host_u32(x)
            assert ctx.lib.qf_matvec_modq_dev(ctx.h, keys[0].h, x.data_ptr(), y.data_ptr()) == 0
            reps = 0
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < min(args.cpu_seconds, 5.0):
                y_h = co.matvec(A_h, x_h, q)
                reps += 1
            cpu_t = time.perf_counter() - t1
            assert np.array_equal(y_h, ctx.to_host_u32(y)), "GPU digest differs from the CPU oracle"
            out["cpu_baseline"] = {"value": reps / cpu_t, "unit": "hashes/s", "cores": 1, "kind": "port",
                                   "sample": f"{reps} single-message hashes of one key (one mat-vec per call, the reference is single-threaded; "
                                             "key warm in the CPU caches); GPU digest verified bit-identical"}
    if R_.rank_ms is not None:
        out["ms_per_step_ranks"] = {"min": R_.rank_ms[0], "max": R_.rank_ms[1]}
    return out


if __name__ == "__main__":
    _a = parse()
    if _a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(_a))
    if _a.workload == "regev":
        headline(_a)
    else:
        other_workloads(_a)
