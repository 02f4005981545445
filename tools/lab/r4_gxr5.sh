#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
out=gpurun_out/lab/gxr_exp2.txt
: > $out
for st in 1 3 9 11 0; do
  NMFX_GXR=1 NMFX_GX_STAGGER=$st timeout -k 10 200 python tools/lab/gxr_time.py 256:1 >> $out 2>&1 || { tail -20 $out; exit 1; }
done
grep -v amdgpu.ids $out
