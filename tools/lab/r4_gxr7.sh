#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
out=gpurun_out/lab/fuse_upd.txt
: > $out
timeout -k 10 700 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu > gpurun_out/lab/gxr_tests.log 2>&1
rc=$?
tail -5 gpurun_out/lab/gxr_tests.log
[ $rc -ne 0 ] && exit $rc
for f in 1 0 1 0; do
  NMFX_GX_FUSE_UPDATE=$f timeout -k 10 200 python tools/lab/gxr_time.py 256:0 512:0 >> $out 2>&1 || { tail -20 $out; exit 1; }
done
grep -v amdgpu.ids $out
