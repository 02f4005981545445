#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_aoadmm.py tests/test_gpu_admm.py -x -q -m gpu -k "kl or KL" > gpurun_out/lab/t_kl.log 2>&1
rc=$?
tail -25 gpurun_out/lab/t_kl.log
[ $rc -ne 0 ] && exit $rc
export NMF_AMD_QUIET=1
timeout -k 10 300 python tools/lab/kl_admm_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lab/kl_time_new.txt
NMFX_KL_BF16=0 timeout -k 10 300 python tools/lab/kl_admm_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lab/kl_time_old.txt
