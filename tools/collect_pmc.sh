#!/bin/bash
# PMC passes over the product kernels of every measured workload (GPU box; separate rocprofv3 runs, --pmc with --kernel-trace
# only, as MI355X_MICROARCH.md prescribes), one summary with the derived figures:
#   hbm_bytes_per_launch = 2 x FETCH_SIZE KB (gfx950: wide streaming reads are counted half) + WRITE_SIZE KB
#   mfma_busy            = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)     (the same normalisation for every kernel)
#   (r4: one row per (kernel, grid) -- the composed path launches one kernel name at V-sized and at Gram-sized shapes)
#   usage: tools/collect_pmc.sh r04      -> gpurun_out/pmc_r04/summary.json  (copy into profiles/)
tag=${1:-r04}
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
for probe in ${PROBES:-cfg2 k128 kl cfg3 pair k256 aokl}; do
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
    name=${probe}_$(echo "$ctr" | tr ' ' '+')
    if [ "$probe" = cfg2 ]; then
      rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python3 bench.py --pmc-child --steps 6 --warmup 2 > /dev/null 2> "$out/$name.err" || { echo "pass $name failed"; tail -3 "$out/$name.err"; continue; }
    else
      rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python3 tools/lab/pmc_probe.py $probe > /dev/null 2> "$out/$name.err" || { echo "pass $name failed"; tail -3 "$out/$name.err"; continue; }
    fi
    find "$out/$name" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$out/$name.csv"
    rm -rf "$out/$name" "$out/$name.err"
    echo "pass $name done"
  done
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/*.csv")):
    probe = os.path.basename(f).split("_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("xyt32_bf16_kernel", "xyt_bf16_kernel", "gxb_gemm_kernel", "gxt_gemm_kernel", "gxt2_gemm_kernel", "gxr_kernel")):
            continue
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        acc[probe + ": " + short + " grid " + str(grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches_seen"] = max(len(v) for v in d.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CYCLES"):
        e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * e["SQ_BUSY_CYCLES"])
    if e.get("SQ_WAVE_CYCLES"):
        e["wave_cycles_waiting_to_issue"] = e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
    if e.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = e.get("SQ_INSTS_VALU", 0.0) / e["SQ_INSTS_MFMA"]
        e["lds_per_mfma"] = e.get("SQ_INSTS_LDS", 0.0) / e["SQ_INSTS_MFMA"]
    res[k] = e
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, e in res.items():
    print(k, {c: (round(v, 3) if v < 100 else round(v)) for c, v in e.items() if c in ("hbm_bytes_per_launch", "mfma_busy", "wave_cycles_waiting_to_issue", "valu_per_mfma", "lds_per_mfma")})
PY
