#!/bin/bash
# Diagnostic builds of libqfgpu: tools/build_variant.sh NAME [-DQF_STAMP] [-DQF_ABL_...]  -> tools/abl/libqfgpu_NAME.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
for f in qf_api qf_gemm qf_matvec qf_ntt qf_ntt_wave qf_sampler qf_sha256; do
  flags=""
  [ $f = qf_gemm ] && flags="$*"
  [ $f = qf_ntt_wave ] && flags="$*"
  rem=""
  [ $f = qf_gemm ] && rem="-Rpass-analysis=kernel-resource-usage"
  /opt/rocm/bin/hipcc $flags $rem -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -c $root/crypto_amd/csrc/$f.hip -mllvm -pragma-unroll-threshold=262144 -o $tmp/$f.o 2> $tmp/$f.remarks &
done
wait
# a GEMM kernel that spills must never run (see crypto_amd/csrc/Makefile): the NOLDS variant of the
# digit-plane kernels does, and faulted the GPU when it was launched -- refuse such a build
python3 $root/crypto_amd/csrc/check_noscratch.py $tmp/qf_gemm.remarks qf_gemm "$QF_ALLOW_SPILL" || { rm -rf $tmp; exit 1; }
mkdir -p $root/tools/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/abl/libqfgpu_$name.so $tmp/*.o
rm -rf $tmp
