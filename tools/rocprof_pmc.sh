#!/bin/bash
# PMC passes over a short bench run (GPU box).  usage: tools/rocprof_pmc.sh <tag> "<counters>" [bench args]
set -e
tag=$1; shift
ctr=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out" -o pmc -- python3 bench.py --no-cpu --steps 6 --warmup 2 --profile-steps 0 "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not any(s in k for s in ("wphase", "hphase", "xyt")):
        continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
