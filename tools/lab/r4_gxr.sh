#!/bin/bash
# the persistent short-contraction kernel (gxr_kernel) against gxb_gemm_kernel: parity tests beyond k = 128, then per-launch times
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 700 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu > gpurun_out/lab/gxr_tests.log 2>&1
rc=$?
tail -15 gpurun_out/lab/gxr_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 240 python tools/lab/bigk_perf.py > gpurun_out/lab/gxr_perf_on.txt 2>&1 || { tail -20 gpurun_out/lab/gxr_perf_on.txt; exit 1; }
NMFX_GXR=0 timeout -k 10 240 python tools/lab/bigk_perf.py > gpurun_out/lab/gxr_perf_off.txt 2>&1 || { tail -20 gpurun_out/lab/gxr_perf_off.txt; exit 1; }
echo ON; cat gpurun_out/lab/gxr_perf_on.txt; echo OFF; cat gpurun_out/lab/gxr_perf_off.txt
