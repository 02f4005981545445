#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab/shardprof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $GRAFT_REPO_ROOT/tools/lab/shard_step_probe.py 2>&1 | grep -v amdgpu.ids | tee $GRAFT_REPO_ROOT/gpurun_out/lab/shard_step.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lab/shardprof -o sp -- python3 $GRAFT_REPO_ROOT/tools/lab/shard_step_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/lab/shardprof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
