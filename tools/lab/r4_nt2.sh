#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
out=gpurun_out/lab/gxt_nt.txt
: > $out
for l in 1 0 1 0; do
  NMFX_GXT_NT=$l timeout -k 10 200 python tools/lab/gxr_time.py 256:0 512:0 >> $out 2>&1 || { tail -20 $out; exit 1; }
done
grep -v amdgpu.ids $out | cut -c1-250
