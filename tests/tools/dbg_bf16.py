"""Self-consistency probe for the split-bf16 MUR path: after every iteration compare the
GPU factors with an f64 recomputation from the PREVIOUS GPU factors, and say which rows /
columns disagree (localises a wrong image: H images -> W wrong, W^T images -> H wrong)."""
import os, sys
sys.path.insert(0, "/root/repo"); os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.engine import Engine
m, n, k = (int(a) for a in (sys.argv[1:4] or (512, 384, 40)))
v = R.planted_matrix(m, n, k, seed=1, dtype=np.float32)
rs = np.random.RandomState(7); w0 = np.abs(rs.randn(m, k)); h0 = np.abs(rs.randn(k, n))
e = Engine(m, n, k); e.upload_v(v); e.set_factors(w0, h0)
vd = v.astype(np.float64)
w, h = w0, h0
for j in range(4):
    e.mur_run(0, 0, 0, 10**9, 1e-5, 1e-5, j, 1)
    wg, hg = e.get_factors()
    we = w * (vd @ h.T) / (w @ (h @ h.T) + 1e-9)
    he = h * (wg.T @ vd) / ((wg.T @ wg) @ h + 1e-9)
    rw = np.abs(wg - we) / (np.abs(we) + 1e-30); rh = np.abs(hg - he) / (np.abs(he) + 1e-30)
    bw = np.argwhere(rw > 1e-3); bh = np.argwhere(rh > 1e-3)
    print(f"it {j}: W max rel {rw.max():.2e} bad {len(bw)} rows {sorted(set(bw[:,0].tolist()))[:12]} cols {sorted(set(bw[:,1].tolist()))[:12]}")
    print(f"it {j}: H max rel {rh.max():.2e} bad {len(bh)} rows {sorted(set(bh[:,0].tolist()))[:12]} cols {sorted(set(bh[:,1].tolist()))[:12]}")
    w, h = wg, hg
print("obj", e.objectives(0, 5))
print("ref", [0.5 * np.sum((vd - a @ b) ** 2) for a, b in [(w0, h0)]])
