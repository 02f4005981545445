#!/usr/bin/env python3
"""Config-3 shape, 30 iterations: the hinted inner rounds against NMFX_AO_HINT=0 (child processes) -- objectives, inner counts
and factors must be identical bit for bit."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json, hashlib
sys.path.insert(0, %(root)r)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix
m, n, k, T = 16384, 8192, 128, 10
v = planted_matrix(m, n, 32, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
with Engine(m, n, k) as eng:
    eng.upload_v(v)
    eng.set_factors(rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01)
    eng.aoadmm_run(0, 1, 0.1, 1, 0.1, T, 10 ** 12, 1e-3, 1e-3, 0, 30)
    eng.synchronize()
    _, _, nobj = eng.state()
    obj = eng.objectives(0, nobj)
    inner = eng.inner_counts(0, 30) & 0xFFFF
    w, h = eng.get_factors()
    print(json.dumps({"obj": hashlib.sha1(obj.tobytes()).hexdigest(), "inner": inner.tolist(), "w": hashlib.sha1(w.tobytes()).hexdigest(),
                      "h": hashlib.sha1(h.tobytes()).hexdigest(), "paths": eng.inner_paths(), "last": float(obj[-1])}))
'''
out = {}
for mode in ("1", "0"):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=dict(os.environ, NMFX_AO_HINT=mode), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    print("hint", mode, "paths", out[mode]["paths"], "last objective", out[mode]["last"], "inner h", [c[0] for c in out[mode]["inner"]], "w", [c[1] for c in out[mode]["inner"]], flush=True)
same = all(out["1"][key] == out["0"][key] for key in ("obj", "inner", "w", "h"))
print("identical:", same)
sys.exit(0 if same else 1)
