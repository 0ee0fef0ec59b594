#!/bin/bash
# A/B of epilogue cache policies and raster chunks on the DualRegev and LPR shapes (GPU box)
for lib in "" NTLD NTST NTBOTH; do
  if [ -n "$lib" ]; then export QF_LIB_PATH=$PWD/tools/abl/libqfgpu_$lib.so; else unset QF_LIB_PATH; fi
  for mc in 16 32 64 128; do
    echo "== lib=${lib:-default} mchunk=$mc"
    QF_GEMM_MCHUNK=$mc timeout -k 10 120 python tools/time_dualregev.py 2>/dev/null | grep "with addend" || exit 1
  done
  for mc in 32 64; do
    echo "== lib=${lib:-default} mchunk=$mc"
    QF_GEMM_MCHUNK=$mc timeout -k 10 120 python tools/time_lpr.py 2>/dev/null | grep "with addend" || exit 1
  done
done
