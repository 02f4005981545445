"""r5: config 3 (AO-ADMM l1n, 16384 x 8192, k = 128, admm_iter 10) per outer iteration for different numbers of
stream-K workers (NMFX_SK_WORKERS is read once per process: one child each)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %(root)r); os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix
m, n, k, T = 16384, 8192, 128, 10
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
eng = Engine(m, n, k); eng.upload_v(v)
w0, h0 = rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01
NEVER = 10 ** 12
eng.set_factors(w0, h0); eng.aoadmm_run(0, 1, 0.1, 1, 0.1, T, NEVER, 1e-3, 1e-3, 0, 200); eng.synchronize()
ts = []
for rep in range(3):
    eng.set_factors(w0, h0); eng.aoadmm_run(0, 1, 0.1, 1, 0.1, T, NEVER, 1e-3, 1e-3, 0, 5); eng.synchronize()
    t0 = time.perf_counter(); eng.aoadmm_run(0, 1, 0.1, 1, 0.1, T, NEVER, 1e-3, 1e-3, 5, 100); eng.synchronize()
    ts.append((time.perf_counter() - t0) / 100 * 1e6)
print("SK " + json.dumps(ts))
'''
for w in sys.argv[1:] or ["0", "255", "254", "252", "248", "240", "224", "192"]:
    env = dict(os.environ)
    if w != "0":
        env["NMFX_SK_WORKERS"] = w
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True)
    line = [x for x in p.stdout.splitlines() if x.startswith("SK ")]
    print("workers", w if w != "0" else "default (ncu - 1)", "us per outer iteration:", [round(x, 1) for x in json.loads(line[0][3:])] if line else p.stderr[-300:])
