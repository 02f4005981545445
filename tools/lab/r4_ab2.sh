#!/bin/bash
# r4: config 3 with and without the side-job overlap, interleaved on one box (bench.py's own cfg3 leg and tools/bench_configs.py)
set -o pipefail
mkdir -p gpurun_out/lab
export NMF_AMD_QUIET=1
for rep in 1 2; do
  timeout -k 10 200 python tools/lab/one_config.py cfg3 > gpurun_out/lab/oc3_new$rep.json 2>gpurun_out/lab/oc3_new$rep.err || { tail -5 gpurun_out/lab/oc3_new$rep.err; exit 1; }
  NMFX_AO_OVERLAP=0 timeout -k 10 200 python tools/lab/one_config.py cfg3 > gpurun_out/lab/oc3_old$rep.json 2>gpurun_out/lab/oc3_old$rep.err || { tail -5 gpurun_out/lab/oc3_old$rep.err; exit 1; }
done
timeout -k 10 200 python tools/bench_configs.py cfg3 > gpurun_out/lab/cfg3_new.json 2>&1 || exit 1
NMFX_AO_OVERLAP=0 timeout -k 10 200 python tools/bench_configs.py cfg3 > gpurun_out/lab/cfg3_old.json 2>&1 || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lab/oc3_*.json")):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        print(f, {k: d[k] for k in d if k in ("name","value","ms_per_step","kernels_us","inner_rounds")} if isinstance(d, dict) else d)
for f in ("gpurun_out/lab/cfg3_new.json","gpurun_out/lab/cfg3_old.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_iter"],4), {k:v["us"] for k,v in d["kernels"].items()})
PY
