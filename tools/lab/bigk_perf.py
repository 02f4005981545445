#!/usr/bin/env python3
"""MUR beyond 128 components on the config-2 matrix: iterations/s and per-launch times of the generic path."""
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
v = planted_matrix(m, n, 64, seed=0, dtype=np.float32)
for k, dist in ((256, 0), (256, 1), (160, 0), (512, 0)):
    rs = np.random.RandomState(0)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    with Engine(m, n, k) as e:
        e.upload_v(v)
        e.set_factors(w0, h0)
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 0, 10)
        e.synchronize()
        steps = 30
        t0 = time.perf_counter()
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 10, steps)
        e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        e.profile_enable(True)
        e.profile_reset()
        e.mur_run(dist, 0.0, 0.0, NEVER, 1e-5, 1e-5, 10 + steps, 10)
        e.synchronize()
        prof = {}
        for name in ("objective", "wphase", "hphase", "gram_tn", "gram_nt", "w_update", "h_update", "small"):
            ms, cnt = e.profile_get(name)
            if cnt:
                prof[name] = round(ms / cnt * 1e3, 1)
        kp = -(-k // 128) * 128
        flops = (4.0 if dist == 0 else 8.0) * m * n * kp + (2.0 * m * n * kp if dist == 0 else 0.0)
        print(json.dumps({"k": k, "distance": "eu" if dist == 0 else "kl", "ms_per_iter": dt * 1e3, "iter_per_s": 1 / dt,
                          "executed_tflops_v_sized": flops / dt / 1e12, "kernels_us": prof}), flush=True)

# AO-ADMM (least squares, prox nn / l1n) beyond 128 components: per outer iteration
from nmf_amd import _lib as L  # noqa: E402
for k in (256,):
    rs = np.random.RandomState(0)
    w0, h0 = 0.05 * np.abs(rs.randn(m, k)), 0.05 * np.abs(rs.randn(k, n))      # (a start of the data's scale: from |randn| the iteration blows up -> "not positive definite")
    with Engine(m, n, k) as e:
        e.upload_v(v)
        e.set_factors(w0, h0)
        e.aoadmm_run(L.EU, L.PROX['nn'], 0.0, L.PROX['l1n'], 0.1, 10, NEVER, 1e-5, 1e-5, 0, 3)
        e.synchronize()
        steps = 8
        t0 = time.perf_counter()
        e.aoadmm_run(L.EU, L.PROX['nn'], 0.0, L.PROX['l1n'], 0.1, 10, NEVER, 1e-5, 1e-5, 3, steps)
        e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        e.profile_enable(True)
        e.profile_reset()
        e.aoadmm_run(L.EU, L.PROX['nn'], 0.0, L.PROX['l1n'], 0.1, 10, NEVER, 1e-5, 1e-5, 3 + steps, 4)
        e.synchronize()
        prof = {}
        for name in ("objective", "wphase", "hphase", "gram_tn", "gram_nt", "images", "prepare", "inner", "small"):
            ms, cnt = e.profile_get(name)
            if cnt:
                prof[name] = round(ms / cnt * 1e3, 1)
        e.state()                                       # (raises on "not positive definite")
        print(json.dumps({"k": k, "solver": "ao_admm eu nn/l1n", "ms_per_iter": dt * 1e3, "iter_per_s": 1 / dt,
                          "inner_rounds": (e.inner_counts(3, steps) & 0xFFFF).tolist(), "kernels_us": prof}), flush=True)

# ADMM (Euclidean, fixed rho, prox nn / l1n) beyond 128 components
for k in (256,):
    rs = np.random.RandomState(0)
    w0, h0 = 0.05 * np.abs(rs.randn(m, k)), 0.05 * np.abs(rs.randn(k, n))
    with Engine(m, n, k) as e:
        e.upload_v(v)
        e.set_factors(w0, h0)
        e.admm_run(L.EU, 1.0, L.PROX['nn'], 0.0, L.PROX['l1n'], 0.1, NEVER, 1e-5, 1e-5, 0, 3)
        e.synchronize()
        steps = 8
        t0 = time.perf_counter()
        e.admm_run(L.EU, 1.0, L.PROX['nn'], 0.0, L.PROX['l1n'], 0.1, NEVER, 1e-5, 1e-5, 3, steps)
        e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        _, _, n_obj = e.state()
        obj = e.objectives(0, n_obj)                     # (r3 read slot 3 + steps, which only a finish call writes: "objective": [0.0])
        assert np.all(np.isfinite(obj)) and obj[-1] < obj[0], obj
        print(json.dumps({"k": k, "solver": "admm eu nn/l1n rho=1", "ms_per_iter": dt * 1e3, "iter_per_s": 1 / dt,
                          "objective_first_last": [float(obj[0]), float(obj[-1])]}), flush=True)
