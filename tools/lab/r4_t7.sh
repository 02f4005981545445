#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_bigk.py -x -q -m gpu -k "k160 or unfused or row_sharded or admm or anls" > gpurun_out/lab/t7.log 2>&1
rc=$?
tail -25 gpurun_out/lab/t7.log
exit $rc
