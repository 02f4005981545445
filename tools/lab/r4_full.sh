#!/bin/bash
# the round-end check as the driver runs it: the whole GPU suite in one process, then smoke()
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu > gpurun_out/lab/full.log 2>&1
rc=$?
tail -8 gpurun_out/lab/full.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
