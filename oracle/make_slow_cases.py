#!/usr/bin/env python3
"""Write tests/golden/slow/*.npz: the ORACLE's outcome (oracle/nmf_ref.py, pinned against the reference by
tests/test_oracle_golden.py) for the GPU-suite cases whose oracle leg takes tens of seconds of scipy NNLS (ANLS with 100+
components).  CPU only, ~3 minutes:

    python oracle/make_slow_cases.py

The cases are defined HERE exactly as the tests define them (seeded inputs, keywords); each file carries the signature of its
inputs (tests/gpu_common.slow_signature) and a test whose definition has drifted from this list finds a mismatching
signature and simply computes its oracle again.  TEST INFRASTRUCTURE ONLY -- fixtures are data."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NMFX_WRITE_SLOW_ORACLE"] = "1"

from gpu_common import slow_oracle, slow_signature  # noqa: E402
from oracle import nmf_ref as R  # noqa: E402


def main():
    # tests/test_gpu_anls.py::test_anls_k64_k128_both_precisions_vs_oracle
    for (m, n, k) in ((300, 220, 40), (260, 400, 100)):
        v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
        kw = dict(lambda_w=0.05, lambda_h=0.02, min_iter=4, max_iter=4, nndsvd_init=(True, "zero"))
        slow_oracle(f"anls_{m}x{n}_k{k}", slow_signature(v, k, kw), lambda: R.anls(v.astype(np.float64), k, **kw))
    # tests/test_gpu_anls.py::test_anls_rank_deficient_passive_set_at_lambda_zero[dead-*]
    for k, (m, n) in ((6, (200, 150)), (40, (300, 260)), (100, (500, 700))):
        iters = 3
        v = R.planted_matrix(m, n, k, seed=k, dtype=np.float32)
        rs = np.random.RandomState(k)
        w0, h0 = rs.rand(m, k), rs.rand(k, n)
        h0[2] = 0.0
        kw = dict(lambda_w=0, lambda_h=0, min_iter=iters, max_iter=iters)
        slow_oracle(f"anls_dead_{m}x{n}_k{k}", slow_signature(v, k, kw, w0, h0),
                    lambda: R.anls(v.astype(np.float64), k, w0=w0, h0=h0, **kw))
    # tests/test_gpu_bigk.py::test_anls_beyond_128_components_vs_oracle
    for (m, n), k, distance in (((520, 400), 160, "eu"), ((300, 420), 144, "kl")):
        v = R.planted_matrix(m, n, 24, seed=m + k, dtype=np.float32)
        kw = dict(distance_type=distance, lambda_w=0.05, lambda_h=0.02, min_iter=3, max_iter=3, nndsvd_init=(True, "zero"))
        slow_oracle(f"anls_{distance}_{m}x{n}_k{k}", slow_signature(v, k, kw), lambda: R.anls(v.astype(np.float64), k, **kw))
    # tests/test_gpu_dist.py::test_sharded_aoadmm_anls_device_path[*-anls_k160]
    m, n, k = 240, 200, 160
    kw = dict(lambda_w=0.1, lambda_h=0.05, min_iter=1, max_iter=2, tol1=1e-3, tol2=1e-3)
    v = R.planted_matrix(m, n, 24, seed=39, dtype=np.float32)
    w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
    slow_oracle("dist_anls_k160", slow_signature(v, k, kw, w0, h0), lambda: R.anls(v.astype(np.float64), k, w0=w0, h0=h0, **kw))
    d = os.path.join(ROOT, "tests", "golden", "slow")
    for f in sorted(os.listdir(d)):
        print(f, os.path.getsize(os.path.join(d, f)))


if __name__ == "__main__":
    main()
