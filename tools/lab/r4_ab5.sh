#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 500 python tools/lab/ab_iter.py cfg2 nmf_amd/lib/libnmfx.so tools/lab/ab/libnmfx_vfront.so --rounds 3 2>&1 | tee gpurun_out/lab/ab5.txt
