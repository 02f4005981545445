#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_bigk.py tests/test_gpu_prox.py tests/test_gpu_admm.py -x -q -m gpu -k "l1inf or prox" > gpurun_out/lab/t8.log 2>&1
rc=$?
tail -25 gpurun_out/lab/t8.log
exit $rc
