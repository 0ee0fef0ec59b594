#!/bin/bash
# Runs ON THE GPU BOX: one PMC pass (SQ issue/wait counters + GRBM_GUI_ACTIVE) over tools/time_ring.py; per-kernel means.
set -e
out=$PWD/gpurun_out/pmc_ring
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out -o p -- python3 tools/time_ring.py > $out/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "wave_kernel" not in k: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, len(c["SQ_WAVE_CYCLES"]))
    print("   ", {n: round(v) for n, v in m.items()})
    g = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if g:
        print("    cycles/launch %.0f  VALU active share of SIMD time: %.3f  (ACTIVE_INST_VALU*4 / (1024 SIMDs * cycles))" % (g, m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * g)),
              " wait_any/wave_cycles %.3f  wait_inst/wave_cycles %.3f  active_any/wave_cycles %.3f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
