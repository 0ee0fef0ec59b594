#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh output directory: per-kernel mean duration (kernel trace) and
HBM traffic per launch from the PMC passes.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so reads are doubled
(MI355X_MICROARCH.md, HBM section).  Writes kernel_stats.csv + traffic.json into the directory."""
import csv, glob, json, os, sys, collections

d = sys.argv[1]
def find(pat):
    f = glob.glob(os.path.join(d, pat), recursive=True)
    return f[0] if f else None

def short(name):
    return name.split("(")[0].replace("void ", "")[:60]

dur = collections.defaultdict(list)
f = find("trace/**/*kernel_trace.csv")
if f:
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = {}
for key, pat in (("fetch_kib", "pmc_fetch/**/*counter_collection.csv"), ("write_kib", "pmc_write/**/*counter_collection.csv")):
    f = find(pat)
    acc = collections.defaultdict(list)
    if f:
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], short(r["Kernel_Name"]))] += float(r["Counter_Value"])
        for (_, k), v in per.items():
            acc[k].append(v)
    ctr[key] = acc
rows = []
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    v = dur[k]
    fe, wr = ctr["fetch_kib"].get(k, []), ctr["write_kib"].get(k, [])
    tail = lambda x: x[len(x) // 2:] if len(x) > 1 else x   # steady-state half (skips first-touch launches)
    rows.append({"kernel": k, "calls": len(v), "total_us": round(sum(v), 1), "mean_us": round(sum(v) / len(v), 2),
                 "min_us": round(min(v), 2), "max_us": round(max(v), 2),
                 "fetch_bytes_per_launch_x2": round(2 * 1024 * sum(tail(fe)) / max(1, len(tail(fe)))) if fe else None,
                 "write_bytes_per_launch": round(1024 * sum(tail(wr)) / max(1, len(tail(wr)))) if wr else None})
# matrix-pipe pass (GEMM workloads): MFMA busy share and LDS conflict share per kernel
f = find("pmc_mfma/**/*counter_collection.csv")
if f:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    tmp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        tmp[(r["Dispatch_Id"], short(r["Kernel_Name"]))][r["Counter_Name"]] = float(r["Counter_Value"])
    for (_, k), m in tmp.items():
        for n, v in m.items():
            per[k][n].append(v)
    mf = {}
    for k, c in per.items():
        m = {n: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for n, v in c.items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8          # the counter sums the 8 XCDs
        if not cyc or not m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            continue
        mf[k] = {"launches": len(c["GRBM_GUI_ACTIVE"]), "cycles_per_launch": round(cyc),
                 "mfma_busy_cycles": round(m["SQ_VALU_MFMA_BUSY_CYCLES"]),
                 # SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy, summed over the 1,024 SIMDs
                 "mfma_busy_share_of_simd_cycles": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),
                 "insts_lds": round(m.get("SQ_INSTS_LDS", 0)), "lds_bank_conflict_cycles": round(m.get("SQ_LDS_BANK_CONFLICT", 0)),
                 "mfma_mops_i8": round(m.get("SQ_INSTS_VALU_MFMA_MOPS_I8", 0)), "wave_cycles": round(m.get("SQ_WAVE_CYCLES", 0)),
                 "effective_clock_ghz_if_launch_ms_known": None}
    json.dump(mf, open(os.path.join(d, "mfma_util.json"), "w"), indent=1)
    for k, v in mf.items():
        print("MFMA", k, v)
# L2 pass (GEMM workloads): hit / miss counts and fabric read requests per launch
f = find("pmc_l2/**/*counter_collection.csv")
if f:
    tmp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], short(r["Kernel_Name"]))
        tmp[key][r["Counter_Name"]] = tmp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for (_, k), m in tmp.items():
        for n, v in m.items():
            per[k][n].append(v)
    l2 = {}
    for k, c in per.items():
        m = {n: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for n, v in c.items()}
        hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
        if hit + miss < 1e5:
            continue
        l2[k] = {"launches": len(next(iter(c.values()))), **{n: round(v) for n, v in m.items()},
                 "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
    json.dump(l2, open(os.path.join(d, "l2.json"), "w"), indent=1)
    for k, v in l2.items():
        print("L2", k, v)
with open(os.path.join(d, "kernel_stats.csv"), "w", newline="") as fo:
    w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()) if rows else ["kernel"])
    w.writeheader(); w.writerows(rows)
json.dump({r["kernel"]: r for r in rows}, open(os.path.join(d, "traffic.json"), "w"), indent=1)
for r in rows[:12]:
    print(r)
