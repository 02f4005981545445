"""r4: where the time of the KL-loss ADMM variants goes (exact-f32 products today): per-kernel times on the config-3 shape."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix

m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 8192, 128)
T = 10
NEVER = 10 ** 12
v = planted_matrix(m, n, min(k, 64), seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01
KERNELS = ("wphase", "wphase_noobj", "objective", "hphase", "gram_nt", "gram_tn", "sum_hht", "w_update", "pack", "h_update", "images",
           "prepare", "inner_h", "inner_w", "sums", "kl_vaux", "kl_vaux_fused", "small", "kl_round_h", "kl_round_w", "transpose")
for name, queue in (("ao_admm kl", lambda e, f, c: e.aoadmm_run(1, 0, 0.0, 0, 0.0, T, NEVER, 1e-3, 1e-3, f, c)),
                    ("admm kl", lambda e, f, c: e.admm_run(1, 1.0, 0, 0.0, 0, 0.0, NEVER, 1e-3, 1e-3, f, c))):
    with Engine(m, n, k) as e:
        e.upload_v(v); e.set_factors(w0, h0)
        queue(e, 0, 3); e.synchronize()
        t0 = time.perf_counter()
        queue(e, 3, 5); e.synchronize()
        dt = (time.perf_counter() - t0) / 5
        _, _, n_obj = e.state()
        obj = e.objectives(0, n_obj)
        e.profile_enable(True); e.profile_reset()
        queue(e, 8, 2); e.synchronize()
        prof = {}
        for kn in KERNELS:
            ms, cnt = e.profile_get(kn)
            if cnt:
                prof[kn] = (round(ms / cnt * 1e3, 1), cnt / 2)
        inner = (e.inner_counts(0, 8) & 0xFFFF).mean(axis=0).tolist() if name.startswith("ao") else None
        print(json.dumps({"solver": name, "shape": [m, n, k], "precision": e.precision(), "ms_per_iter": round(dt * 1e3, 3), "inner": inner,
                          "obj_first_last": [float(obj[0]), float(obj[-1])], "kernels_us_and_launches_per_iter": prof}), flush=True)
