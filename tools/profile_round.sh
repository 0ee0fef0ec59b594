#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + two PMC passes (FETCH_SIZE, WRITE_SIZE are
# collected in separate runs, never together with a trace) of the headline bench, then summarises
# them into gpurun_out/prof_$1/.   usage: tools/profile_round.sh r1b [bench args...]
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 10 --warmup 2 --no-cpu $*"
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py $args > $out/bench_trace.json 2> $out/trace.log
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py $args > $out/bench_fetch.json 2> $out/fetch.log
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py $args > $out/bench_write.json 2> $out/write.log
python3 tools/profile_summary.py $out
