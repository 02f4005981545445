#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 900 python -m pytest "tests/test_gpu_fullsize.py::test_config5_shard_through_the_sharded_entry_points_vs_oracle" -x -q -m gpu -s > gpurun_out/lab/t3.log 2>&1
rc=$?
tail -15 gpurun_out/lab/t3.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/lab/bench_r4a.json 2> gpurun_out/lab/bench_r4a.err
rc=$?
tail -3 gpurun_out/lab/bench_r4a.err
python tools/show_bench.py gpurun_out/lab/bench_r4a.json 2>/dev/null | head -60
exit $rc
