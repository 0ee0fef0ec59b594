#!/bin/bash
# copies the summaries of gpurun_out/prof_<tag>/<workload>/ into profiles/<round>/<workload>/   usage: tools/collect_profiles.sh r2 r2
tag=$1; round=$2
for d in gpurun_out/prof_$tag/*/; do
  wl=$(basename $d)
  mkdir -p profiles/$round/$wl
  for f in kernel_stats.csv traffic.json mfma_util.json l2.json rocprofv3_kernel_stats.csv bench_trace.json command.txt; do
    [ -f $d/$f ] && cp $d/$f profiles/$round/$wl/$f
  done
done
ls -R profiles/$round | head -40
