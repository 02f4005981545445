#!/bin/bash
# r4: config 3 (bench.py's leg): default (no pack launch) vs NMFX_AO_PACK=1 vs NMFX_AO_OVERLAP=0, interleaved on one box
set -o pipefail
mkdir -p gpurun_out/lab
export NMF_AMD_QUIET=1
timeout -k 10 600 python -m pytest tests/test_gpu_aoadmm.py -x -q -m gpu > gpurun_out/lab/t_ao.log 2>&1 || { tail -20 gpurun_out/lab/t_ao.log; exit 1; }
tail -2 gpurun_out/lab/t_ao.log
for rep in 1 2; do
  timeout -k 10 200 python tools/lab/one_config.py cfg3 > gpurun_out/lab/oc3_new$rep.json 2>gpurun_out/lab/oc3_new$rep.err || { tail -5 gpurun_out/lab/oc3_new$rep.err; exit 1; }
  NMFX_AO_PACK=1 timeout -k 10 200 python tools/lab/one_config.py cfg3 > gpurun_out/lab/oc3_pack$rep.json 2>gpurun_out/lab/oc3_pack$rep.err || { tail -5 gpurun_out/lab/oc3_pack$rep.err; exit 1; }
  NMFX_AO_OVERLAP=0 timeout -k 10 200 python tools/lab/one_config.py cfg3 > gpurun_out/lab/oc3_old$rep.json 2>gpurun_out/lab/oc3_old$rep.err || { tail -5 gpurun_out/lab/oc3_old$rep.err; exit 1; }
done
timeout -k 10 200 python tools/bench_configs.py cfg3 > gpurun_out/lab/cfg3_new.json 2>&1 || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lab/oc3_*.json")):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        print(f, d.get("ms_per_step"), d.get("value"))
d=json.loads(open("gpurun_out/lab/cfg3_new.json").read().strip().splitlines()[-1])
print(round(d["ms_per_iter"],4), {k:v["us"] for k,v in d["kernels"].items()})
PY
