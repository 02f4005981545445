#!/usr/bin/env python3
"""WH / objective error of both arithmetic modes against the f64 oracle (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.mur import mur
for (m, n, k, iters) in [(512, 384, 40, 40), (640, 1000, 64, 40), (2048, 1024, 64, 100)]:
    v = R.planted_matrix(m, n, k, seed=1, dtype=np.float32)
    np.random.seed(7); ref = R.mur(v.astype(np.float64), k, distance_type="eu", min_iter=iters, max_iter=iters)
    for p in ("f32", "bf16"):
        os.environ["NMFX_PRECISION"] = p
        np.random.seed(7); res = mur(v.copy(), k, distance_type="eu", min_iter=iters, max_iter=iters)
        err = np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
        orel = np.max(np.abs(np.asarray(res.obj_history) - ref.obj_history) / np.abs(ref.obj_history))
        print(f"{m}x{n} k={k} iters={iters} {p:5s} WH err {err:.3e}  obj max rel {orel:.3e}")
