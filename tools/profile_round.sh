#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): for ONE bench workload, a kernel-trace stats pass, two HBM-traffic PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate runs, never together with a trace) and, for the GEMM
# workloads, one matrix-pipe pass -- all of the same `python3 bench.py --workload W --no-cpu` command,
# the program directly after `--`.  Summaries go to gpurun_out/prof_<tag>/<workload>/ ; copy them to
# profiles/<round>/<workload>/.      usage: tools/profile_round.sh TAG WORKLOAD [bench args...]
tag=$1; wl=$2; shift; shift
out=$PWD/gpurun_out/prof_$tag/$wl
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
args="--workload $wl --steps 40 --warmup 5 --no-cpu --no-check --headline-only $*"
echo "python3 bench.py $args" > $out/command.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py $args > $out/bench_trace.json 2> $out/trace.log || { echo "trace pass failed"; tail -5 $out/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py $args > $out/bench_fetch.json 2> $out/fetch.log || { echo "fetch pass failed"; tail -5 $out/fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py $args > $out/bench_write.json 2> $out/write.log || { echo "write pass failed"; tail -5 $out/write.log; exit 1; }
case $wl in regev|sis|lpr|dualregev)
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -o m -- python3 bench.py $args > $out/bench_mfma.json 2> $out/mfma.log || { echo "mfma pass failed"; tail -5 $out/mfma.log; exit 1; }
# L2 pass: hits, misses and the read requests that left the L2 for the fabric (MALL / HBM), 4 TCC counters
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/pmc_l2 -o l -- python3 bench.py $args > $out/bench_l2.json 2> $out/l2.log || { echo "l2 pass failed"; tail -5 $out/l2.log; exit 1; }
;; esac
python3 tools/profile_summary.py $out
# keep only the summaries (the raw per-dispatch CSVs are tens of MB)
cp $out/trace/*kernel_stats.csv $out/rocprofv3_kernel_stats.csv 2>/dev/null || cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats.csv
rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_l2
