#!/bin/bash
# Runs ON THE GPU BOX: every bench workload once (plain, un-profiled), one JSON line each into gpurun_out/bench_$1_<workload>.json
tag=$1; shift
mkdir -p gpurun_out
for wl in regev sis lpr ringlpr fdh matvec regev_cycle; do
  timeout -k 10 400 python3 bench.py --workload $wl "$@" > gpurun_out/bench_${tag}_$wl.json 2> gpurun_out/bench_${tag}_$wl.err
  rc=$?
  echo "$wl rc=$rc $(head -c 300 gpurun_out/bench_${tag}_$wl.json)"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/bench_${tag}_$wl.err; exit $rc; fi
done
