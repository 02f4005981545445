#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_bigk.py tests/test_gpu_knobs.py -x -q -m gpu -k "mur_beyond" > gpurun_out/lab/den_tests.log 2>&1
rc=$?
tail -5 gpurun_out/lab/den_tests.log
exit $rc
