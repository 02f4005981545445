#!/bin/bash
# PMC passes over the k = 128 product kernels (separate rocprofv3 runs; --pmc with --kernel-trace only)
export TMPDIR=/tmp
out=gpurun_out/k128_pmc; rm -rf $out; mkdir -p $out
for ctr in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo "$ctr" | tr ' ' '+')
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$name -o pmc -- python3 tools/lab/k128_probe.py > /dev/null 2> $out/$name.err || { echo "pass $name failed"; tail -3 $out/$name.err; continue; }
  find $out/$name -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/$name.csv; rm -rf $out/$name
done
python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xyt32_bf16_kernel" not in k: continue
        key = "W phase k128" if "<true, 3, 0, false, 128>" in k else "H phase k128" if "<false, 3, 0, false, 128>" in k else k[:50]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
