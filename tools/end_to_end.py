#!/usr/bin/env python3
"""Wall time of whole API calls at config size (GPU box): upload, NNDSVD on the device, iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from nmf_amd.synth import planted_matrix
from nmf_amd import NMF
v = planted_matrix(16384, 8192, 64, seed=0, dtype=np.float32)
for method, k, kw in (("ao_admm", 128, dict(reg_w=(0.1, "l1n"), reg_h=(0.1, "l1n"), min_iter=50, max_iter=50)),
                      ("anls", 64, dict(min_iter=20, max_iter=20)),
                      ("mur", 64, dict(distance_type="eu", min_iter=500, max_iter=500, nndsvd_init=(True, "mean")))):
    t0 = time.perf_counter()
    nmf = NMF(v, k)
    nmf.factorize(method=method, **kw)
    dt = time.perf_counter() - t0
    print(f"{method} k={k}: {nmf.results.i + 1} iterations, objective {nmf.results.obj_history[0]:.4g} -> {nmf.results.obj_history[-1]:.4g}, "
          f"wall {dt:.2f} s (NNDSVD init, upload, iterations, download)", flush=True)
