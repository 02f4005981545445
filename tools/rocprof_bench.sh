#!/bin/bash
# Collect the rocprofv3 kernel-trace summary of the bench command (GPU box).
# usage: tools/rocprof_bench.sh <tag>   -> gpurun_out/prof_<tag>/  (+ stats csv copied to profiles/ by hand)
set -e
tag=${1:-r01}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --no-cpu --no-traffic --steps 40 --warmup 5 --profile-steps 0 > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
find "$out" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
head -12 "$out/kernel_stats.csv"
cat "$out/bench.json"
