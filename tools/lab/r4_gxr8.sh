#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
out=gpurun_out/lab/gxr_pf.txt
: > $out
timeout -k 10 300 env NMFX_GX_STAGGER=33 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu -k "mur or kl" > gpurun_out/lab/gxr_tests.log 2>&1
rc=$?
tail -5 gpurun_out/lab/gxr_tests.log
[ $rc -ne 0 ] && exit $rc
for st in 33 1 33 1; do
  NMFX_GX_STAGGER=$st timeout -k 10 200 python tools/lab/gxr_time.py 256:0 256:1 >> $out 2>&1 || { tail -20 $out; exit 1; }
done
grep -v amdgpu.ids $out
