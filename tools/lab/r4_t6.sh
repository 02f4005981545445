#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 1100 python -m pytest tests/test_gpu_grid.py tests/test_gpu_mur.py tests/test_gpu_bigk.py "tests/test_gpu_fullsize.py::test_config2_tight_tolerance_stop_is_refereed_in_float64" -x -q -m gpu -s > gpurun_out/lab/t6.log 2>&1
rc=$?
grep -v "^\[" gpurun_out/lab/t6.log | tail -12
exit $rc
