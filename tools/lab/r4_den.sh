#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_bigk.py tests/test_gpu_knobs.py tests/test_gpu_dist.py -x -q -m gpu -k "mur or GX or k160" > gpurun_out/lab/den_tests.log 2>&1
rc=$?
tail -5 gpurun_out/lab/den_tests.log
[ $rc -ne 0 ] && exit $rc
for d in 1 0 1 0; do
  NMFX_GX_DEN_BF16=$d timeout -k 10 200 python tools/lab/gxr_time.py 256:0 512:0 2>&1 | grep -v amdgpu.ids | cut -c1-260
done | tee gpurun_out/lab/den.txt
