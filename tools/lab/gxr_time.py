#!/usr/bin/env python3
"""MUR beyond 128 components on the config-2 matrix: per-launch times of the composed path's products (k, distance from argv)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

m, n = 16384, 8192
NEVER = 10 ** 12
cases = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or [(256, 0), (256, 1)]
v = planted_matrix(m, n, 64, seed=0, dtype=np.float32)
for k, dist in cases:
    rs = np.random.RandomState(0)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    with Engine(m, n, k) as e:
        e.upload_v(v)
        e.set_factors(w0, h0)
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 0, 10)
        e.synchronize()
        steps = 40
        t0 = time.perf_counter()
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 10, steps)
        e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        e.profile_enable(True)
        e.profile_reset()
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 10 + steps, 10)
        e.synchronize()
        prof = {}
        for name in ("objective", "wphase", "hphase", "gram_tn", "gram_nt", "w_update", "h_update", "images", "small"):
            ms, cnt = e.profile_get(name)
            if cnt:
                prof[name] = round(ms / cnt * 1e3, 1)
        hist = e.objective_history(10 + steps + 10) if hasattr(e, "objective_history") else None
        print(json.dumps({"k": k, "distance": "eu" if dist == 0 else "kl", "ms_per_iter": round(dt * 1e3, 4), "kernels_us": prof,
                          "env": {x: os.environ.get(x) for x in ("NMFX_GXR", "NMFX_GX_STAGGER")}}), flush=True)
