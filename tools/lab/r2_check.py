"""The recorded (device) objective of a MUR-eu run against nmfx_objective_f64 of the same iterates -- how much a cheaper residual
product costs in objective accuracy.    NMFX_LIB=<build> python tools/lab/r2_check.py [m n k]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 16384, 128)
v = planted_matrix(m, n, k, seed=5, dtype=np.float32)
rs = np.random.RandomState(0)
NEVER = 10 ** 12
with Engine(m, n, k) as e:
    e.upload_v(v)
    e.set_factors(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n)))
    done, rows = 0, []
    for upto in (3, 10, 50, 200, 1000, 3000):
        e.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, done, upto - done)
        f64 = e.objective_f64()                            # the pair after `upto` iterations, product and sum in float64
        e.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, upto, 1)  # its objective as the LOOP records it: fused into the W phase of iteration `upto`
        e.synchronize()
        done = upto + 1
        rec = e.objectives(upto, 1)[0]
        hist = e.objectives(max(0, upto - 12), min(12, upto))
        rows.append((upto, rec, f64, (rec - f64) / f64, float(np.std(np.diff(hist, n=2))) / rec if len(hist) > 4 else 0.0))
    print(os.environ.get("NMFX_LIB", "default lib"), f"{m}x{n} k={k}")
    for r in rows:
        print("  after %5d iterations: recorded %.9g  f64 %.9g  relative difference %+.3e   (spread of the last second differences / objective %.2e)" % r)
