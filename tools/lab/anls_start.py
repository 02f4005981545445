"""ANLS from the reference's uniform random start at the config-2 shape: time of the first iterations (sparse iterates: large
complements) and of the settled ones; NNLS fallback counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from nmf_amd.synth import planted_matrix
from nmf_amd.engine import Engine
m, n, k = 16384, 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 64
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rs = np.random.RandomState(seed); w0, h0 = rs.rand(m, k), rs.rand(k, n)
eng = Engine(m, n, k); eng.upload_v(v); eng.set_factors(w0, h0)
run = lambda f, c: eng.anls_run(0.0, 0.0, 10**12, 1e-3, 1e-3, f, c)
run(0, 1); eng.synchronize()            # (first-call allocations)
eng.set_factors(w0, h0); eng.synchronize()
out = []
done = 0
for cnt in (3, 5, 12):
    t0 = time.perf_counter(); run(done, cnt); eng.synchronize(); dt = (time.perf_counter() - t0) / cnt
    done += cnt
    w, h = eng.get_factors()
    out.append((done, round(dt * 1e3, 3), eng.nnls_fallbacks(), "support W %.3f H %.3f" % ((w > 0).mean(), (h > 0).mean())))
print("ANLS k=%d from rand start: (iterations done, ms per iteration in the last chunk, cumulative fallbacks)" % k, out)
