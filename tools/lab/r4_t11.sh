#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu -s -k "16384x8192 or 4096x2048" > gpurun_out/lab/t11.log 2>&1
rc=$?
grep "PARITY\|passed\|failed\|Error\|assert" gpurun_out/lab/t11.log | tail -12
exit $rc
