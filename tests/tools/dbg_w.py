import os, sys
sys.path.insert(0, "/root/repo"); os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.engine import Engine
m, n, k = 512, 384, 40
v = R.planted_matrix(m, n, k, seed=1, dtype=np.float32)
rs = np.random.RandomState(7); w0 = np.abs(rs.randn(m, k)); h0 = np.abs(rs.randn(k, n))
e = Engine(m, n, k); e.upload_v(v); e.set_factors(w0, h0)
vd = v.astype(np.float64)
e.mur_run(0, 0, 0, 10**9, 1e-5, 1e-5, 0, 1)
wg, hg = e.get_factors()
A = vd @ h0.T; D = w0 @ (h0 @ h0.T)
we = w0 * A / (D + 1e-9)
rw = np.abs(wg - we) / (np.abs(we) + 1e-30)
bad = ~(rw < 1e-3)
np.set_printoptions(linewidth=200, precision=4)
print("bad by row%64:", bad.reshape(-1, 64, k).sum(axis=(0, 2)))
print("bad by col:", bad.sum(axis=0))
print("nan count", np.isnan(wg).sum())
r = int(np.argwhere(bad)[0][0])
print("row", r, "gpu", wg[r, :12]); print("exp", we[r, :12])
dimp = w0[r] * A[r] / wg[r] - 1e-9
print("implied D", dimp[:12]); print("true D   ", D[r, :12])
Dimp = w0 * A / wg - 1e-9
print("ratio implied D / true D, rows 0-7, cols 0-35")
print((Dimp / D)[:8, :36])
# does implied D match D of some other (row, col)?
for (r, c) in [(1, 0), (1, 5), (2, 3), (0, 17), (4, 20)]:
    val = Dimp[r, c]
    cand = np.argwhere(np.abs(D[:64] - val) / val < 2e-6)
    hh = h0 @ h0.T
    print((r, c), "implied", val, "true", D[r, c], "matches D at", cand[:4].tolist())
