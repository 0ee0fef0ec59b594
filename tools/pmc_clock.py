"""Average shader clock and MFMA-busy share of the GEMM launches of a rocprofv3 --pmc run
(GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES, csv output): python3 tools/pmc_clock.py DIR [DIR ...].
GRBM_GUI_ACTIVE is the sum over the 8 XCDs; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs), as tools/profile_summary.py."""
import csv, sys, glob, collections, statistics
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = {}
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:48], r["Dispatch_Id"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    rows = [k for k in acc if "qf_gemm" in k[0]]
    rows = rows[len(rows) // 2:]          # the second half: the larger of the script's two shapes
    us = statistics.mean(dur[k] for k in rows)
    gui = statistics.mean(sum(acc[k]["GRBM_GUI_ACTIVE"]) for k in rows) / 8.0
    mf = statistics.mean(sum(acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) for k in rows)
    print("%s  %s  n=%d  %.1f us  clock %.3f GHz  MFMA busy %.3f" % (d, rows[0][0], len(rows), us, gui / us / 1e3, mf / (gui * 1024.0)))
