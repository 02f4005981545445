#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 600 python tools/lab/bench_legs.py aoadmm_kl_on_cfg3_shape aoadmm_kl_exact_f32 admm_kl_on_cfg3_shape 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lab/legs.txt | cut -c1-700
