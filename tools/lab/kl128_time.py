"""MUR-KL at k = 128 (16-row kernel today) against k = 64 (32-row KL kernel) on the config-2 matrix: per-phase times."""
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
        eng.mur_run(1, 0.0, 0.0, 10 ** 9, 1e-5, 1e-5, 0, 5)
        eng.synchronize()
        eng.profile_enable(True)
        eng.mur_run(1, 0.0, 0.0, 10 ** 9, 1e-5, 1e-5, 5, 20)
        eng.synchronize()
        out = []
        for name in ("wphase", "hphase", "w_update", "h_update", "pack", "images", "sums"):
            ms, cnt = eng.profile_get(name)
            if cnt:
                out.append(f"{name} {1e3 * ms / cnt:.1f} us x{cnt // 20}")
        _, _, n_obj = eng.state()
        obj = eng.objectives(0, n_obj)
        print(f"k={k}: " + ", ".join(out) + f"; obj[-1] = {obj[-1]:.6f}")
