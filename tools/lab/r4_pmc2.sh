#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROBES="k256" bash tools/collect_pmc.sh r04b > gpurun_out/collect_pmc_r04b.log 2>&1 || { tail -20 gpurun_out/collect_pmc_r04b.log; exit 1; }
tail -30 gpurun_out/collect_pmc_r04b.log | cut -c1-300
