#!/usr/bin/env python3
"""WH error of AO-ADMM against the oracle for the product forms (child process per form: the knobs are read once)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.ao_admm import ao_admm
m, n, k, T, it = (int(a) for a in sys.argv[1:6])
rw, rh = json.loads(sys.argv[6])
v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
kw = dict(distance_type="eu", reg_w=tuple(rw), reg_h=tuple(rh), min_iter=it, max_iter=it, admm_iter=T, nndsvd_init=(True, "zero"))
res = ao_admm(v.copy(), k, **kw)
ref = R.ao_admm(v.astype(np.float64), k, **kw)
err = float(np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)))
print(json.dumps({"err": err, "obj_rel": float(abs(res.obj_history[-1] - ref.obj_history[-1]) / ref.obj_history[-1])}))
'''
cases = [((320, 448, 100), 16, 14, [[0.02, "l1n"], [0, "nn"]]), ((320, 448, 100), 8, 14, [[0.02, "l1n"], [0, "nn"]]),
         ((384, 640, 100), 10, 7, [[0.1, "l1n"], [0.05, "l1n"]]), ((320, 448, 100), 16, 14, [[0.1, "l1n"], [0.1, "l1n"]])]
for shape, T, it, regs in cases:
    for env in ({}, {"NMFX_BF16_TERMS": "4"}, {"NMFX_PRECISION": "f32"}):
        out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, *[str(a) for a in shape], str(T), str(it), json.dumps(regs)],
                             env=dict(os.environ, **env), capture_output=True, text=True)
        print(shape, T, it, regs, env, out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-500:], flush=True)
