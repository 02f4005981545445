#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_knobs.py -x -q -m gpu -k "GXR or GXT or GX_" > gpurun_out/lab/t10.log 2>&1
rc=$?
tail -8 gpurun_out/lab/t10.log
exit $rc
