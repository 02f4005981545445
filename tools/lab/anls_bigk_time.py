"""Time of the generic (k > 128) ANLS path: per-kernel profile of a few iterations."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import nmf_ref as R
from nmf_amd.engine import Engine

m, n, k, iters = (int(a) for a in (sys.argv[1:5] + ["2048", "1536", "160", "3"][len(sys.argv) - 1:]))
v = R.planted_matrix(m, n, 24, seed=1, dtype=np.float32)
rs = np.random.RandomState(0)
with Engine(m, n, k) as eng:
    eng.upload_v(v)
    eng.set_factors(rs.rand(m, k), rs.rand(k, n))
    eng.anls_set_distance(0)
    eng.profile_enable(True)
    t0 = time.time()
    eng.anls_run(0.05, 0.02, 10 ** 9, 1e-3, 1e-3, 0, iters)
    eng.synchronize()
    print(f"{m}x{n} k={k}: {iters} iterations in {time.time() - t0:.3f} s")
    for name in ("nnls", "wphase", "hphase", "gram_nt", "gram_tn", "objective"):
        ms, cnt = eng.profile_get(name)
        if cnt:
            print(f"  {name:10s} {ms / cnt:10.3f} ms x {cnt}")
    print("  diagnostics (evicted, capped):", eng.diagnostics())
