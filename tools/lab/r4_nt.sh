#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
out=gpurun_out/lab/gxr_nt.txt
: > $out
for l in nt2 nt nt2 nt; do
  echo "lib $l" >> $out
  NMFX_LIB=$PWD/tools/lab/ab/libnmfx_$l.so timeout -k 10 200 python tools/lab/gxr_time.py 256:0 256:1 >> $out 2>&1 || { tail -20 $out; exit 1; }
done
grep -v amdgpu.ids $out | cut -c1-330
