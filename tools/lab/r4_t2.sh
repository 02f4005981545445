#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 1000 python -m pytest tests/test_gpu_aoadmm.py tests/test_gpu_knobs.py tests/test_gpu_grid.py tests/test_gpu_admm.py tests/test_gpu_prox.py "tests/test_gpu_dist.py::test_factorize_takes_the_native_exchange_only_on_request_and_falls_back_together" "tests/test_gpu_dist.py::test_bench_two_ranks_on_one_gpu_runs_both_sharded_legs" -x -q -m gpu > gpurun_out/lab/t2.log 2>&1
rc=$?
tail -25 gpurun_out/lab/t2.log
exit $rc
