#!/bin/bash
# GPU BOX: cycles per VALU instruction (per SIMD, 8 waves/SIMD) for the integer ops of the NTT butterflies
set -e
out=$PWD/gpurun_out/pmc_ubench; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 60 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out -o p -- ./tools/ubench > $out/run.log 2>&1
cat $out/run.log | grep -v "^$" | tail -8
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    acc[(r["Dispatch_Id"], r["Kernel_Name"][:40])][r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), m in sorted(acc.items(), key=lambda t: int(t[0][0])):
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(d, k, "cycles %.0f  VALU insts %.3g  SIMD-cycles per VALU inst %.2f" % (cyc, m["SQ_INSTS_VALU"], cyc * 1024 / m["SQ_INSTS_VALU"]))
PY
