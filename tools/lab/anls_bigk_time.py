#!/usr/bin/env python3
"""ANLS beyond 128 components on the config-2 matrix: ms per iteration and per-launch times (NMFX_GX_ANLS_BF16=0: exact-f32 products)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

m, n, k = 16384, 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 256
NEVER = 10 ** 12
v = planted_matrix(m, n, 64, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
with Engine(m, n, k) as e:
    e.upload_v(v)
    e.set_factors(rs.rand(m, k), rs.rand(k, n))
    e.anls_run(0.0, 0.0, NEVER, 1e-3, 1e-3, 0, 3)
    e.synchronize()
    t0 = time.perf_counter()
    e.anls_run(0.0, 0.0, NEVER, 1e-3, 1e-3, 3, 5)
    e.synchronize()
    dt = (time.perf_counter() - t0) / 5
    e.profile_enable(True); e.profile_reset()
    e.anls_run(0.0, 0.0, NEVER, 1e-3, 1e-3, 8, 2)
    e.synchronize()
    prof = {}
    for kn in ("objective", "wphase", "hphase", "gram_tn", "gram_nt", "images", "nnls", "small"):
        ms, cnt = e.profile_get(kn)
        if cnt:
            prof[kn] = (round(ms / cnt * 1e3, 1), cnt / 2)
    _, _, n_obj = e.state()
    obj = e.objectives(0, n_obj)
    print(json.dumps({"k": k, "ms_per_iter": round(dt * 1e3, 3), "bf16": os.environ.get("NMFX_GX_ANLS_BF16", "1"), "obj_first_last": [float(obj[0]), float(obj[-1])],
                      "kernels_us_and_launches": prof}), flush=True)
