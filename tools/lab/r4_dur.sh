#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 1150 python -m pytest tests/test_gpu_dist.py tests/test_gpu_fullsize.py tests/test_gpu_bigk.py -x -q -m gpu --durations=30 > gpurun_out/lab/dur.log 2>&1
rc=$?
tail -45 gpurun_out/lab/dur.log
exit $rc
