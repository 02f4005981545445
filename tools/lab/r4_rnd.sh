#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_bigk.py tests/test_gpu_dist.py -x -q -m gpu -k "ao or admm or k160 or 160" > gpurun_out/lab/rnd_tests.log 2>&1
rc=$?
tail -6 gpurun_out/lab/rnd_tests.log
[ $rc -ne 0 ] && exit $rc
for f in 0 1 0 1; do
  if [ $f = 1 ]; then export NMFX_GX_ROUNDS_F32=1; else unset NMFX_GX_ROUNDS_F32; fi
  echo "f32 rounds: $f"
  timeout -k 10 240 python tools/lab/bigk_perf.py 2>&1 | grep ao_admm | cut -c1-600
done | tee gpurun_out/lab/rnd_perf.txt
