#!/usr/bin/env python3
"""The composed path at config 5's shape (131072 x 16384) with k = 256: index arithmetic of the persistent kernels at 2^31 elements.
MUR-eu and MUR-kl, a few iterations: finite, decreasing objectives, and the recorded Euclidean objective against the device's float64
referee (nmfx_objective_f64) of the returned pair."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402

m, n, k = 131072, 16384, 256
NEVER = 10 ** 12
rs = np.random.RandomState(0)
t0 = time.perf_counter()
a = np.abs(rs.randn(m, 32)).astype(np.float32)
b = np.abs(rs.randn(32, n)).astype(np.float32)
v = a @ b
v += 0.01 * np.abs(rs.randn(*v.shape).astype(np.float32))
print("V drawn in %.1f s" % (time.perf_counter() - t0), flush=True)
w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
for dist in (0, 1):
    with Engine(m, n, k) as e:
        e.upload_v(v)
        e.set_factors(w0, h0)
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 0, 3)
        e.synchronize()
        t0 = time.perf_counter()
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 3, 3)
        e.finish_a(dist, 6) if hasattr(e, "finish_a") else None
        e.synchronize()
        dt = (time.perf_counter() - t0) / 3
        _, _, n_obj = e.state()
        obj = e.objectives(0, n_obj)
        out = {"distance": "eu" if dist == 0 else "kl", "ms_per_iter": round(dt * 1e3, 2), "objectives": [float(x) for x in obj]}
        assert np.all(np.isfinite(obj)) and np.all(np.diff(obj) < 0), obj
        print(json.dumps(out), flush=True)
print("ok")
