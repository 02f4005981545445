#!/usr/bin/env python3
"""Which paths of the hinted inner rounds a few AO-ADMM problems exercise, and whether the inner counts are the oracle's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np  # noqa: E402
from oracle import nmf_ref as R  # noqa: E402  (lab tool: compares against the oracle like the tests do)
from nmf_amd.ao_admm import ao_admm  # noqa: E402

CASES = [((300, 520, 24), (0.02, "l1n"), (0, "nn"), 16, 30), ((300, 520, 24), (0.02, "l1n"), (0, "nn"), 10, 30),
         ((300, 520, 40), (0.02, "l1n"), (0, "nn"), 16, 30), ((300, 520, 40), (0.02, "l1n"), (0, "nn"), 10, 30),
         ((320, 448, 100), (0.02, "l1n"), (0, "nn"), 16, 24), ((320, 448, 100), (0.02, "l1n"), (0, "nn"), 9, 24)]
for shape, rw, rh, T, it in CASES:
    m, n, k = shape
    v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
    kw = dict(distance_type="eu", reg_w=rw, reg_h=rh, min_iter=it, max_iter=it, admm_iter=T, nndsvd_init=(True, "zero"))
    ref = R.ao_admm(v.astype(np.float64), k, **kw)
    res = ao_admm(v.copy(), k, **kw)
    same = [tuple(r) for r in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]]
    print(shape, rw, rh, T, it, "paths", ao_admm.last_inner_paths, "counts equal", same, flush=True)
    print("   h:", [int(r[0]) for r in ao_admm.last_inner_counts], flush=True)
    print("   w:", [int(r[1]) for r in ao_admm.last_inner_counts], flush=True)
    if not same:
        print("   ref h:", [int(t[0]) for t in ref.trace["inner"]], "\n   ref w:", [int(t[1]) for t in ref.trace["inner"]], flush=True)
