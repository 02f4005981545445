#!/usr/bin/env python3
"""ANLS per-kernel breakdown at a config-size shape (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from nmf_amd.synth import planted_matrix
from nmf_amd.engine import Engine
m, n, k = 16384, 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 64
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0); w0 = rs.rand(m, k) + 0.01; h0 = rs.rand(k, n) / k + 0.01
eng = Engine(m, n, k); eng.upload_v(v); eng.set_factors(w0, h0)
run = lambda f, c: eng.anls_run(0.0, 0.0, 10**12, 1e-3, 1e-3, f, c)
run(0, 2); eng.synchronize(); t0 = time.perf_counter(); run(2, 5); eng.synchronize()
dt = (time.perf_counter() - t0) / 5
eng.profile_enable(True); eng.profile_reset(); run(7, 3); eng.synchronize()
prof = {}
for kn in ("wphase", "wphase_noobj", "objective", "hphase", "gram_nt", "gram_tn", "pack", "nnls", "small", "sums"):
    ms, cnt = eng.profile_get(kn)
    if cnt: prof[kn] = (round(ms / cnt * 1e3, 1), cnt / 3)
print(f"ANLS {m}x{n} k={k}: {1/dt:.1f} iter/s {dt*1e3:.2f} ms", prof, "nnls fallbacks (problems, half-steps) after 10 iterations:", eng.nnls_fallbacks())
if "--stats" in sys.argv:          # library built with NMFX_EXTRA_DEFS=-DNMFX_NNLS_STATS
    import ctypes
    from nmf_amd import _lib
    out = (ctypes.c_ulonglong * 8)()
    _lib.load().nmfx_debug_nnls_stats(out)
    print("nnls: problems", out[2], "mean iterations", out[0] / max(out[2], 1), "max", out[1], "back-up exchanges per problem", out[3] / max(out[2], 1),
          "pivots per problem", out[4] / max(out[2], 1), "mean final support", out[5] / max(out[2], 1))
