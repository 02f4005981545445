#!/usr/bin/env python3
"""Time the device top-k SVD at config size (GPU box); optionally LAPACK on the host beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from nmf_amd import synth as R
from nmf_amd.engine import Engine
m, n = 16384, 8192
for k in (64, 128):
    v = R.planted_matrix(m, n, k, seed=0, dtype=np.float32)
    with Engine(m, n, k) as eng:
        eng.upload_v(v)
        eng.synchronize()
        t0 = time.perf_counter()
        u, s, vt, sweeps, resid = eng.topk_svd(k)
        dt = time.perf_counter() - t0
    print(f"planted {m}x{n} k={k}: device top-k SVD {dt:.2f} s, {sweeps} sweeps, residual {resid:.1e}, s[:3]={s[:3]}", flush=True)
v = np.random.RandomState(1).rand(m, n).astype(np.float32)
with Engine(m, n, 64) as eng:
    eng.upload_v(v)
    t0 = time.perf_counter()
    u, s, vt, sweeps, resid = eng.topk_svd(64, max_sweeps=400)
    print(f"uniform {m}x{n} k=64: device {time.perf_counter() - t0:.2f} s, {sweeps} sweeps, residual {resid:.1e}", flush=True)
if "--host" in sys.argv:
    t0 = time.perf_counter()
    np.linalg.svd(v.astype(np.float64), full_matrices=False)
    print(f"numpy.linalg.svd (f64, {os.cpu_count()} host threads): {time.perf_counter() - t0:.1f} s")
