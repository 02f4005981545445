"""The f64-evaluated decrease of the device's own iterates near the stop (tol 1e-3, config 2): how smooth is it, where does the
reference's rule fire on it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
import bench
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
guard = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-5
m, n, k = 16384, 8192, 64
NEVER = 10 ** 12
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
with Engine(m, n, k) as e:
    e.upload_v(v)
    e.set_factors(w0, h0)
    e.set_stop_guard(guard)
    done, rule = 0, 0
    while not rule:
        e.mur_run(0, 0.0, 0.0, 100, tol, tol, done, 256); done += 256
        rule, stop_i, n_obj = e.state()
    j = stop_i + 1
    print("candidate: rule", rule, "stop_i", stop_i)
    e.resume()
    t0 = time.time(); o_old = e.objective_f64(); print("f64 objective:", o_old, "in", round((time.time() - t0) * 1e3, 2), "ms; device value", e.objectives(j, 1)[0])
    dec = []
    fired = None
    for i in range(j, j + 700):
        e.mur_run(0, 0.0, 0.0, NEVER, tol, tol, i, 1)
        o_new = e.objective_f64()
        dec.append(o_old - o_new)
        if fired is None and o_new >= o_old - tol:
            fired = i
        o_old = o_new
    dec = np.array(dec)
    print("f64 rule fires at", fired)
    k0 = (fired or j) - j
    print("decreases around it:", dec[max(0, k0 - 4):k0 + 4])
    print("jitter of the f64 decreases (std of second differences):", np.std(np.diff(dec)))
    dev = -np.diff(e.objectives(j, 600))
    print("jitter of the device decreases:", np.std(np.diff(dev)))
