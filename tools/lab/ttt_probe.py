"""How many MUR-eu iterations does config 2 need to the reference's DEFAULT tolerances (tol1 = tol2 = 1e-5, min_iter = 100)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
import bench
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix
m, n, k = 16384, 8192, 64
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
with Engine(m, n, k) as e:
    e.upload_v(v)
    for tol in (1e-3, 1e-4, 1e-5):
        t0 = time.time()
        rule, stop_i, done, secs, n_obj = bench.converge_on_device(e, w0, h0, tol, int(sys.argv[1]) if len(sys.argv) > 1 else 400000)
        print(tol, rule, stop_i, done, round(secs, 2), e.objectives(n_obj - 1, 1)[0], flush=True)
