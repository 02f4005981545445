#!/bin/bash
# r4: whole MUR iterations, this build vs the round-3 library (tools/lab/ab/libnmfx_r3.so), interleaved on one box
set -o pipefail
mkdir -p gpurun_out/lab
for cfg in c5shard8 cfg2 cfg3shape; do
  echo "== $cfg"
  timeout -k 10 400 python tools/lab/ab_iter.py $cfg nmf_amd/lib/libnmfx.so tools/lab/ab/libnmfx_r3.so --rounds 2 2>&1 | tee -a gpurun_out/lab/ab4.txt || exit 1
done
