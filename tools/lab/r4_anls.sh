#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
for b in 1 0; do NMFX_GX_ANLS_BF16=$b timeout -k 10 300 python tools/lab/anls_bigk_time.py 256 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/lab/anls_bigk.txt
