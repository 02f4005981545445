#!/usr/bin/env python3
"""Parity of the split-bf16 paths on awkward shapes (GPU box): MUR-eu and AO-ADMM against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.mur import mur
from nmf_amd.ao_admm import ao_admm
bad = 0
for (m, n, k) in [(70, 90, 40), (130, 64, 64), (1000, 129, 100), (129, 1000, 128), (64, 64, 33), (257, 255, 65), (2049, 130, 50)]:
    v = R.planted_matrix(m, n, min(k, min(m, n)), seed=m + n, dtype=np.float32)
    np.random.seed(3); res = mur(v.copy(), k, distance_type="eu", min_iter=15, max_iter=15, lambda_w=0.01)
    np.random.seed(3); ref = R.mur(v.astype(np.float64), k, distance_type="eu", min_iter=15, max_iter=15, lambda_w=0.01)
    e1 = np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    o1 = np.max(np.abs(np.asarray(res.obj_history) / np.asarray(ref.obj_history) - 1))
    kw = dict(distance_type="eu", reg_w=(0.05, "l1n"), reg_h=(0, "nn"), min_iter=5, max_iter=5, nndsvd_init=(True, "zero"))
    try:
        r2 = ao_admm(v.copy(), k, **kw); f2 = R.ao_admm(v.astype(np.float64), k, **kw)
        e2 = np.linalg.norm(r2.w @ r2.h - f2.w @ f2.h) / np.linalg.norm(v.astype(np.float64))
        same = [tuple(t) for t in ao_admm.last_inner_counts] == [tuple(t) for t in f2.trace["inner"]]
    except Exception as ex:          # noqa: BLE001
        e2, same = float("nan"), repr(ex)[:80]
    ok = e1 < 1e-4 and o1 < 5e-4 and (e2 < 1e-4)
    bad += not ok
    print(f"{m}x{n} k={k}: MUR WH {e1:.2e} obj {o1:.1e} | AO-ADMM WH {e2:.2e} inner-counts-equal {same} {'ok' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
