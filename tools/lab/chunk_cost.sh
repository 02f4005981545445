#!/bin/bash
# What the chunked exchange costs on ONE GPU (world of one: the collectives are no-ops, the chunk launches and packs are not)
for shape in 16384x16384x128 2048x8192x64; do
  for ch in 1 2 4; do
    NMFX_DIST_CHUNKS=$ch NMFX_BENCH_FORCE_SHARDED=1 NMFX_BENCH_SHAPE=$shape python3 bench.py --no-cpu --no-traffic --no-others --steps 50 --warmup 5 --tol-max-iter 0 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$shape chunks $ch:', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v['ms_per_launch']*1e3,1) for k,v in d['kernels'].items()})"
  done
done
