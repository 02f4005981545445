#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
NMFX_EXTRA_DEFS=-DNMFX_EXP_STAMPS python -m nmf_amd.build > gpurun_out/lab/build_st.log 2>&1 || { tail -20 gpurun_out/lab/build_st.log; exit 1; }
export NMF_AMD_QUIET=1
echo "== k = 128, 16384 x 16384 (config 5, one rank's shard at N = 8)"
timeout -k 10 300 python tools/lab/stamps.py 16384 16384 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lab/stamps_k128.txt
echo "== k = 64, 16384 x 8192 (config 2)"
timeout -k 10 300 python tools/lab/stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lab/stamps_k64.txt
