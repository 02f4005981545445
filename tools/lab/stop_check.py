"""oracle_stop_check of bench.py at another tolerance:  python tools/lab/stop_check.py 1e-3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
import bench
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
m, n, k = 16384, 8192, 64
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
with Engine(m, n, k) as e:
    e.upload_v(v)
    rule, stop_i, done, secs, n_obj = bench.converge_on_device(e, w0, h0, tol, 400000)
    dev_tail = e.objectives(max(0, stop_i - 20), min(n_obj, stop_i + 2) - max(0, stop_i - 20))
    print("device", rule, stop_i, round(secs, 2), "decreases", (-np.diff(dev_tail))[-6:])
    chk = bench.oracle_stop_check(e, v, w0, h0, tol, rule, stop_i, lead=40, span=120)
    print(chk)
