"""A few MUR-eu iterations at k = 128 on one rank's shard of config 5 (16384 x 16384), for rocprofv3 --pmc passes:
    rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d out -- python3 tools/lab/k128_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

m, n, k = 16384, 16384, 128
v = planted_matrix(m, n, 32, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
with Engine(m, n, k) as eng:
    eng.upload_v(v)
    eng.set_factors(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n)))
    eng.mur_run(0, 0, 0, 10 ** 12, 1e-5, 1e-5, 0, 6)
    eng.synchronize()
