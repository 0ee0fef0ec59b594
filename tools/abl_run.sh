#!/bin/bash
# runs tools/time_gemm.py for the default library and every ablation build in tools/abl (GPU box)
echo "== default"; timeout -k 10 120 python tools/time_gemm.py regev sis 2>/dev/null | grep -E "default"
for f in tools/abl/libqfgpu_*.so; do
  case $f in *STAMP*) continue;; esac
  echo "== $f"; QF_LIB_PATH=$PWD/$f timeout -k 10 120 python tools/time_gemm.py regev sis 2>/dev/null | grep -E "default"
done
