import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from nmf_amd.synth import planted_matrix
from nmf_amd.engine import Engine
m, n, k = 16384, 8192, 128
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0); w0 = rs.rand(m, k) + 0.01; h0 = rs.rand(k, n) / k + 0.01
for prec in ("f32", "bf16"):
    os.environ["NMFX_PRECISION"] = prec
    eng = Engine(m, n, k); eng.upload_v(v); eng.set_factors(w0, h0)
    run = lambda f, c: eng.admm_run(0, 1.0, 1, 0.1, 1, 0.1, 10**12, 1e-3, 1e-3, f, c)
    run(0, 3); eng.synchronize(); t0 = time.perf_counter(); run(3, 20); eng.synchronize()
    dt = (time.perf_counter() - t0) / 20
    _, _, nobj = eng.state(); obj = eng.objectives(0, nobj)
    print(prec, f"{1/dt:.1f} iter/s {dt*1e3:.3f} ms", obj[0], obj[-1], flush=True)
    eng.close()
