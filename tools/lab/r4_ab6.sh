#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
export NMF_AMD_QUIET=1
timeout -k 10 400 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu -k "mur or aoadmm" > gpurun_out/lab/t9.log 2>&1 || { tail -20 gpurun_out/lab/t9.log; exit 1; }
tail -2 gpurun_out/lab/t9.log
for rep in 1 2; do
echo "--- swizzled (this build)"
timeout -k 10 300 python tools/lab/bigk_perf.py 2>&1 | grep '"distance"' | cut -c1-330
echo "--- plain (x, y) = blockIdx"
NMFX_LIB=$PWD/tools/lab/ab/libnmfx_noswz.so NMFX_LIB_LAX=1 timeout -k 10 300 python tools/lab/bigk_perf.py 2>&1 | grep '"distance"' | cut -c1-330
done
