"""A few iterations of one workload for rocprofv3 --pmc passes (the program after `--` must be python itself):
    python3 tools/lab/pmc_probe.py k128   MUR-eu, one rank's shard of config 5 (16384 x 16384, k = 128)
    python3 tools/lab/pmc_probe.py kl     MUR-kl, config 4 (32768 x 16384, k = 64)
    python3 tools/lab/pmc_probe.py cfg3   AO-ADMM, config 3 (16384 x 8192, k = 128, planted start)
    python3 tools/lab/pmc_probe.py pair   two k = 64 MUR-eu problems per pass, config-2 shape
    python3 tools/lab/pmc_probe.py k256   MUR-eu beyond 128 components (16384 x 8192, k = 256): the split-bf16 NT kernel over operand planes
    python3 tools/lab/pmc_probe.py aokl   AO-ADMM with the KL loss on the config-3 shape (r4: auxiliaries / products / objective on split bf16)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "k128"
NEVER = 10 ** 12
m, n, k = {"k128": (16384, 16384, 128), "kl": (32768, 16384, 64), "cfg3": (16384, 8192, 128), "pair": (16384, 8192, 128),
           "k256": (16384, 8192, 256), "aokl": (16384, 8192, 128)}[mode]
v = planted_matrix(m, n, 32, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
with Engine(m, n, k) as eng:
    eng.upload_v(v)
    if mode == "cfg3":
        eng.set_factors(rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01)
        eng.aoadmm_run(0, 1, 0.1, 1, 0.1, 10, NEVER, 1e-3, 1e-3, 0, 6)
    elif mode == "aokl":
        eng.set_factors(rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01)
        eng.aoadmm_run(1, 0, 0.0, 0, 0.0, 4, NEVER, 1e-3, 1e-3, 0, 2)
    elif mode == "pair":
        eng.set_factors(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n)))
        eng.mur_pair_run([0.0, 0.1], [0.0, 0.05], NEVER, 1e-5, 1e-5, 0, 6)
    else:
        eng.set_factors(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n)))
        eng.mur_run(1 if mode == "kl" else 0, 0, 0, NEVER, 1e-5, 1e-5, 0, 6)
    eng.synchronize()
