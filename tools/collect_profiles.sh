#!/bin/bash
# Round profiles (GPU box): kernel-trace summary of the whole bench (config 2 + other_configs), PMC passes for the
# dominant kernels (separate rocprofv3 runs, --pmc with --kernel-trace only), and the plain bench line.
#   usage: tools/collect_profiles.sh r02      -> gpurun_out/profiles_r02/  (copy what is to be judged into profiles/)
set -e
tag=${1:-r02}
export TMPDIR=/tmp
out=gpurun_out/profiles_$tag
rm -rf "$out"; mkdir -p "$out"
# 1. the bench line as the driver runs it (no profiler attached)
NMFX_BENCH_DETAIL="$out/bench_n1_detail.json" python3 bench.py --steps 20 --warmup 5 > "$out/bench_n1.json" 2> "$out/bench_n1.err" || { tail -5 "$out/bench_n1.err"; exit 1; }
grep -v "^bench_detail: " "$out/bench_n1.err" > "$out/bench_n1.err.tmp"; mv "$out/bench_n1.err.tmp" "$out/bench_n1.err"      # (the full record is in bench_n1_detail.json)
# 2. kernel trace of the same command (CPU leg and PMC children off: they are not kernels of the product)
NMFX_BENCH_DETAIL="$out/bench_under_rocprof_detail.json" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 bench.py --no-cpu --no-traffic --steps 200 --warmup 5 \
    > "$out/bench_under_rocprof.json" 2> "$out/trace.err" || { tail -5 "$out/trace.err"; exit 1; }
find "$out/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
rm -rf "$out/trace"
# 3. PMC passes (config 2 only), one counter set per run
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    name=$(echo "$ctr" | tr ' ' '+')
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/pmc_$name" -o pmc -- python3 bench.py --pmc-child --steps 6 --warmup 2 \
        > /dev/null 2> "$out/pmc_$name.err" || { echo "pmc pass $name failed"; tail -3 "$out/pmc_$name.err"; continue; }
    find "$out/pmc_$name" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$out/pmc_$name.csv"
    rm -rf "$out/pmc_$name"
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xyt32_bf16_kernel" not in k:
            continue
        key = "W phase" if "<true, 3" in k else "H phase" if "<false, 3" in k else k[:60]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024      # FETCH_SIZE x2: gfx950 correction (MI355X_MICROARCH.md, HBM)
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
head -14 "$out/kernel_stats.csv" | cut -c1-200
tail -c 1500 "$out/bench_n1.json"
