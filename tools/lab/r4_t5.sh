#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu > gpurun_out/lab/t5.log 2>&1
rc=$?
tail -12 gpurun_out/lab/t5.log
[ $rc -ne 0 ] && exit $rc
export NMF_AMD_QUIET=1
timeout -k 10 300 python tools/lab/one_config.py aoadmm_k256_on_cfg2_shape 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['config'], round(d['ms_per_step'],4), d.get('mean_inner_rounds_h_w'), {k:v['us_per_launch'] for k,v in d['kernels'].items()})
"
NMFX_GX_ROUNDS=4 timeout -k 10 300 python tools/lab/one_config.py aoadmm_k256_on_cfg2_shape 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('old', d['config'], round(d['ms_per_step'],4), d.get('mean_inner_rounds_h_w'), {k:v['us_per_launch'] for k,v in d['kernels'].items()})
"
