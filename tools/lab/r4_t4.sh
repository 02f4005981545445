#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 1100 python -m pytest tests/test_gpu_aoadmm.py tests/test_gpu_admm.py tests/test_gpu_knobs.py tests/test_gpu_dist.py tests/test_gpu_edge_shapes.py tests/test_gpu_mur.py -x -q -m gpu > gpurun_out/lab/t4.log 2>&1
rc=$?
tail -15 gpurun_out/lab/t4.log
exit $rc
