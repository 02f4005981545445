#!/bin/bash
# round-4 collection: bench line, kernel trace, PMC of config 2 (tools/collect_profiles.sh), then the PMC probes of the other kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1 || { tail -20 gpurun_out/collect_r04.log; exit 1; }
tail -5 gpurun_out/collect_r04.log | cut -c1-300
PROBES="cfg3 k128 aokl k256" bash tools/collect_pmc.sh r04 > gpurun_out/collect_pmc_r04.log 2>&1 || { tail -20 gpurun_out/collect_pmc_r04.log; exit 1; }
tail -30 gpurun_out/collect_pmc_r04.log | cut -c1-260
