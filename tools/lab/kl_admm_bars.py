#!/usr/bin/env python3
"""How close the KL variants of ADMM / AO-ADMM come to the oracle at a few shapes (to set the bars of the tests)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np  # noqa: E402
from oracle import nmf_ref as R  # noqa: E402  (lab tool)
from nmf_amd.ao_admm import ao_admm  # noqa: E402
from nmf_amd.admm import admm  # noqa: E402

CASES = [("ao", (96, 80, 8), dict(reg_w=(0, "nn"), reg_h=(0, "nn"), admm_iter=10), 8),
         ("ao", (200, 160, 12), dict(reg_w=(0.05, "l1n"), reg_h=(0.05, "l1n"), admm_iter=8), 6),
         ("ao", (320, 256, 40), dict(reg_w=(0.05, "l1n"), reg_h=(0, "nn"), admm_iter=6), 5),
         ("ao", (256, 384, 100), dict(reg_w=(0, "nn"), reg_h=(0.05, "l1n"), admm_iter=5), 4),
         ("admm", (200, 160, 12), dict(reg_w=(0.05, "l1n"), reg_h=(0.05, "l1n"), rho=1.0), 12),
         ("admm", (320, 256, 40), dict(reg_w=(0, "nn"), reg_h=(0.05, "l1n"), rho=2.0), 10),
         ("admm", (256, 384, 100), dict(reg_w=(0.05, "l2n"), reg_h=(0, "nn"), rho=1.0), 8)]
for kind, (m, n, k), kw, it in CASES:
    v = R.planted_matrix(m, n, min(k, 32), seed=m + n + k, dtype=np.float32)
    kw = dict(kw, distance_type="kl", min_iter=it, max_iter=it, nndsvd_init=(True, "zero"))
    t0 = time.time()
    with np.errstate(all="ignore"):
        ref = (R.ao_admm if kind == "ao" else R.admm)(v.astype(np.float64), k, **kw)
    t1 = time.time()
    res = (ao_admm if kind == "ao" else admm)(v.copy(), k, **kw)
    err = float(np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)))
    od = float(np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history)))
    inner = ""
    if kind == "ao":
        inner = "inner equal %s" % ([tuple(r) for r in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]])
    print(kind, (m, n, k), kw.get("reg_w"), kw.get("reg_h"), "oracle %.1fs" % (t1 - t0), "WH %.2e" % err, "obj %.2e" % od, "i", res.i, ref.i, inner, flush=True)
