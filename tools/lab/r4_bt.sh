#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
NMFX_EXTRA_DEFS=-DNMFX_EXP_BLOCKTIME python -m nmf_amd.build > gpurun_out/lab/build_bt.log 2>&1 || { tail -20 gpurun_out/lab/build_bt.log; exit 1; }
export NMF_AMD_QUIET=1
for c in 5 8 12; do
echo "--- cross = $c"
NMFX_SK_CROSS=$c timeout -k 10 300 python tools/lab/sk_block_times.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lab/sk_block_times.txt
done
