#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
out=gpurun_out/lab/gxt_il.txt
: > $out
for st in 1 16 0 17 16 1; do
  NMFX_GXR=1 NMFX_GX_STAGGER=$st timeout -k 10 200 python tools/lab/gxr_time.py 256:0 512:0 >> $out 2>&1 || { tail -20 $out; exit 1; }
done
grep -v amdgpu.ids $out
