#!/bin/bash
# r5: the KL-loss ADMM variants' S / dual_v in the auxiliaries kernel's register order -- tests of the KL paths, then the two
# compositions timed alternately with the build under test and tools/lab/ab/libnmfx_base.so (same box).
# usage (on the GPU box): bash tools/lab/r5_kl_layout.sh > gpurun_out/r5/s_layout.log
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_aoadmm.py tests/test_gpu_admm.py -m gpu -q -x -k "kl" 2>&1 | tail -3 || exit 1
python -m pytest tests/test_gpu_knobs.py -m gpu -q -x -k "KL_BF16" 2>&1 | tail -2 || exit 1
for lib in nmf_amd/lib/libnmfx.so tools/lab/ab/libnmfx_base.so nmf_amd/lib/libnmfx.so tools/lab/ab/libnmfx_base.so; do
    [ -f $lib ] || continue
    NMFX_LIB=$lib NMFX_LIB_LAX=1 python tools/lab/bench_legs.py aoadmm_kl_on_cfg3_shape admm_kl_on_cfg3_shape 2>/dev/null | LIB=$lib python -c '
import sys, json, os
for ln in sys.stdin:
    d = json.loads(ln); k = d["kernels"]
    print(os.environ["LIB"], d["config"], round(d["ms_per_step"], 3), "ms; kl_vaux", k["kl_vaux"]["us_per_launch"], "transpose", k["transpose"]["us_per_launch"],
          "products", k["hphase"]["us_per_launch"], k["wphase_noobj"]["us_per_launch"])
'
done
