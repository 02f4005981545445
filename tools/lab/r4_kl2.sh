#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
export NMF_AMD_QUIET=1
for l in dvnt base dvnt base; do
  echo "lib $l"
  NMFX_LIB=$PWD/tools/lab/ab/libnmfx_$l.so timeout -k 10 300 python tools/lab/kl_admm_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-900
done | tee gpurun_out/lab/kl_nt.txt
