"""Accuracy of a MUR-KL build against the float64 oracle: WH error and objective history after `iters` iterations.
    NMFX_LIB=tools/lab/ab/libnmfx_q1.so python tools/lab/kl_q1_check.py 4096 16384 64 10"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.mur import mur  # noqa: E402
from oracle import nmf_ref as R  # noqa: E402

m, n, k, iters = (int(x) for x in sys.argv[1:5])
v = R.planted_matrix(m, n, k, seed=3, dtype=np.float32)
kw = dict(distance_type="kl", min_iter=iters, max_iter=iters)
np.random.seed(0)
res = mur(v.copy(), k, **kw)
np.random.seed(0)
ref = R.mur(v.astype(np.float64), k, **kw)
num = den = 0.0
for a in range(0, m, 2048):
    d = res.w[a:a + 2048] @ res.h - ref.w[a:a + 2048] @ ref.h
    num += float(np.sum(d * d)); den += float(np.sum(v[a:a + 2048].astype(np.float64) ** 2))
obj = np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history))
print(f"{os.environ.get('NMFX_LIB', 'default lib')}: {m}x{n} k={k}, {iters} iterations: WH {np.sqrt(num / den):.3e}, objective max rel diff {obj:.3e}, "
      f"W rel {np.linalg.norm(res.w - ref.w) / np.linalg.norm(ref.w):.3e}, H rel {np.linalg.norm(res.h - ref.h) / np.linalg.norm(ref.h):.3e}")
