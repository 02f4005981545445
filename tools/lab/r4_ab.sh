#!/bin/bash
# r4: parity tests of the AO-ADMM paths, then config 3 with and without the side-job overlap (same box)
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 700 python -m pytest tests/test_gpu_aoadmm.py tests/test_gpu_knobs.py -x -q -m gpu > gpurun_out/lab/t_ao.log 2>&1
rc=$?
tail -5 gpurun_out/lab/t_ao.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/bench_configs.py cfg3 > gpurun_out/lab/cfg3_new.json 2>&1 || { tail -5 gpurun_out/lab/cfg3_new.json; exit 1; }
NMFX_AO_OVERLAP=0 timeout -k 10 200 python tools/bench_configs.py cfg3 > gpurun_out/lab/cfg3_old.json 2>&1 || { tail -5 gpurun_out/lab/cfg3_old.json; exit 1; }
timeout -k 10 200 python tools/bench_configs.py cfg3 > gpurun_out/lab/cfg3_new2.json 2>&1 || exit 1
tail -n 1 gpurun_out/lab/cfg3_new.json; tail -n 1 gpurun_out/lab/cfg3_old.json; tail -n 1 gpurun_out/lab/cfg3_new2.json
