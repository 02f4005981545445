#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 1000 python -m pytest tests/test_gpu_bigk.py tests/test_gpu_dist.py tests/test_gpu_knobs.py -x -q -m gpu > gpurun_out/lab/t9.log 2>&1
rc=$?
tail -8 gpurun_out/lab/t9.log
exit $rc
