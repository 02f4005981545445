#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
out=gpurun_out/lab/gxr_ab.txt
: > $out
for env in "NMFX_GXR=1 NMFX_GX_STAGGER=1" "NMFX_GXR=1 NMFX_GX_STAGGER=0" "NMFX_GXR=0 NMFX_GX_STAGGER=1" "NMFX_GXR=0 NMFX_GX_STAGGER=0" "NMFX_GXR=1 NMFX_GX_STAGGER=1"; do
  env $env timeout -k 10 200 python tools/lab/gxr_time.py 256:0 256:1 512:0 >> $out 2>&1 || { tail -20 $out; exit 1; }
done
grep -v amdgpu.ids $out
