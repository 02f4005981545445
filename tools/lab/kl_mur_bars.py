#!/usr/bin/env python3
"""Objective closeness of MUR-KL to the oracle / goldens (to set the bars of tests/test_gpu_mur.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.mur import mur
out = []
for shape in [(512, 384, 40), (300, 520, 100), (200, 130, 33)]:
    m, n, k = shape
    v = R.planted_matrix(m, n, k, seed=m + n, dtype=np.float32)
    kw = dict(distance_type="kl", min_iter=25, max_iter=25, lambda_w=0.02, lambda_h=0.01)
    np.random.seed(11); res = mur(v.copy(), k, **kw)
    np.random.seed(11); ref = R.mur(v.astype(np.float64), k, **kw)
    od = np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history)
    out.append((shape, float(od.max()), int(od.argmax()), float(ref.obj_history[-1] / ref.obj_history[0])))
from gpu_common import run_fixture
for name in ("mur_kl", "mur_kl_lambda", "mur_kl_sparse"):
    z, meta, v, res = run_fixture(name, mur)
    od = np.abs(np.asarray(res.obj_history) - z["obj_history"]) / np.abs(z["obj_history"])
    out.append((name, float(od.max()), int(od.argmax()), float(z["obj_history"][-1] / z["obj_history"][0])))
print(json.dumps(out))
'''
for env in ({"NMFX_PRECISION": "bf16"}, {"NMFX_PRECISION": "f32"}):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=dict(os.environ, **env), capture_output=True, text=True)
    print(env, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-800:], flush=True)
