#!/bin/bash
# round-5 collection: bench line + detail, kernel trace, PMC of config 2 (tools/collect_profiles.sh), then the PMC probes of the other
# kernels -- config 4 (kl) included this round (r04's summary had no config-4 row)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1 || { tail -20 gpurun_out/collect_r05.log; exit 1; }
tail -5 gpurun_out/collect_r05.log | cut -c1-300
PROBES="kl cfg3 k128 aokl" bash tools/collect_pmc.sh r05 > gpurun_out/collect_pmc_r05.log 2>&1 || { tail -20 gpurun_out/collect_pmc_r05.log; exit 1; }
tail -30 gpurun_out/collect_pmc_r05.log | cut -c1-260
