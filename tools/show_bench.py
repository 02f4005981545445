"""Pretty-print the one-line JSON of bench.py (value, ms/step, loop, kernels): from the files named on the
command line, or from stdin when there are none."""
import fileinput, json, sys
for ln in fileinput.input():
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    print(d["n_gpus"], d["config"].get("rows_per_gpu"), d["config"].get("loop"), round(d["value"], 1), "iter/s",
          round(d["ms_per_step"] * 1000, 1), "us/step",
          {k: round(v["ms_per_launch"] * 1000, 1) for k, v in (d.get("kernels") or {}).items()})
