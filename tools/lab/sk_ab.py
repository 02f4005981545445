"""r4: the two V-sized products of AO-ADMM (config 3) as the round-3 launches and as stream-K launches without / with the side job,
back to back in ONE process (nmfx_profile_repeat), interleaved: us per launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix

m, n, k, T = 16384, 8192, 128, 10
if len(sys.argv) > 3:
    m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
eng = Engine(m, n, k)
eng.upload_v(v); eng.set_factors(rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01)
eng.aoadmm_run(0, 1, 0.1, 1, 0.1, T, 10 ** 12, 1e-3, 1e-3, 0, 30)
eng.synchronize()
names = ("ao_hphase", "sk_hphase", "sk_hphase_side", "ao_wphase", "sk_wphase", "sk_wphase_side")
res = {nm: [] for nm in names}
for rep in range(4):
    for nm in names:
        res[nm].append(eng.profile_repeat(nm, 100) * 1e3)
for nm in names:
    print("%-16s %s" % (nm, " ".join("%6.1f" % x for x in res[nm])))
