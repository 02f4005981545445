#!/usr/bin/env python3
"""Throughput of the pair mode (two k = 64 MUR-eu problems per pass over V) against two separate runs, config-2 shape."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

m, n, k = 16384, 8192, 64
NEVER = 10 ** 12
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
wa, ha = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
wb, hb = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = {}
with Engine(m, n, 64) as e:
    e.upload_v(v)
    for tag, (w, h, lw, lh) in {"a": (wa, ha, 0.0, 0.0), "b": (wb, hb, 0.1, 0.05)}.items():
        e.set_factors(w, h)
        e.mur_run(0, lw, lh, NEVER, 1e-5, 1e-5, 0, 400)
        e.synchronize()
        e.set_factors(w, h)
        e.mur_run(0, lw, lh, NEVER, 1e-5, 1e-5, 0, 10)
        e.synchronize()
        t0 = time.perf_counter()
        e.mur_run(0, lw, lh, NEVER, 1e-5, 1e-5, 10, steps)
        e.synchronize()
        out["single_" + tag + "_us"] = (time.perf_counter() - t0) / steps * 1e6
with Engine(m, n, 128) as e:
    e.upload_v(v)
    w0 = np.concatenate([wa, wb], axis=1)
    h0 = np.concatenate([ha, hb], axis=0)
    e.set_factors(w0, h0)
    e.mur_pair_run([0.0, 0.1], [0.0, 0.05], NEVER, 1e-5, 1e-5, 0, 400)
    e.synchronize()
    e.set_factors(w0, h0)
    e.mur_pair_run([0.0, 0.1], [0.0, 0.05], NEVER, 1e-5, 1e-5, 0, 10)
    e.synchronize()
    t0 = time.perf_counter()
    e.mur_pair_run([0.0, 0.1], [0.0, 0.05], NEVER, 1e-5, 1e-5, 10, steps)
    e.synchronize()
    out["pair_us"] = (time.perf_counter() - t0) / steps * 1e6
    e.profile_enable(True)
    e.profile_reset()
    e.mur_pair_run([0.0, 0.1], [0.0, 0.05], NEVER, 1e-5, 1e-5, 10 + steps, 20)
    e.synchronize()
    prof = {}
    for name in ("wphase", "hphase", "gram_tn", "gram_nt", "sum_hht", "w_update", "pack", "h_update", "small", "images"):
        ms, cnt = e.profile_get(name)
        if cnt:
            prof[name] = round(ms / cnt * 1e3, 1)
    out["pair_kernels_us"] = prof
out["sequential_us_for_both"] = out["single_a_us"] + out["single_b_us"]
out["per_problem_speedup"] = out["sequential_us_for_both"] / out["pair_us"]
print(json.dumps(out))
