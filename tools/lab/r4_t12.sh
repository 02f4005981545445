#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_bigk.py tests/test_gpu_dist.py tests/test_gpu_knobs.py -x -q -m gpu -k "anls or ANLS" > gpurun_out/lab/t12.log 2>&1
rc=$?
tail -6 gpurun_out/lab/t12.log
exit $rc
