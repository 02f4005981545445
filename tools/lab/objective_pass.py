"""Time of the objective-only pass (ADMM's objective of (w, h), the closing objective of a run): 32-row kernel without its A stages
against the 16-row kernel (NMFX_XYT16=1 forces the latter for EVERY product: compare the 'objective' scope only)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix

m, n = 16384, 8192
for k in (64, 128):
    v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
    rs = np.random.RandomState(0)
    with Engine(m, n, k) as eng:
        eng.upload_v(v)
        eng.set_factors(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n)))
        eng.admm_run(0, 1.0, 0, 0.0, 0, 0.0, 10 ** 9, 1e-3, 1e-3, 0, 3)
        eng.synchronize()
        eng.profile_enable(True)
        eng.admm_run(0, 1.0, 0, 0.0, 0, 0.0, 10 ** 9, 1e-3, 1e-3, 3, 20)
        eng.synchronize()
        ms, cnt = eng.profile_get("objective")
        _, _, n_obj = eng.state()
        obj = eng.objectives(0, n_obj)
        print(f"k={k}: objective pass {1e3 * ms / max(cnt, 1):.1f} us x {cnt}; objective[-1] = {obj[-1]:.6f}")
