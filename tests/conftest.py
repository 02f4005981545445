import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's BLAS pool = the CPUs the container may use: a 256-thread pool under a 16-CPU quota gets the whole
    # process throttled (nmf_amd.synth.usable_cpus), which made the full-size oracle runs several times slower
    from nmf_amd.synth import limit_blas_threads
    limit_blas_threads()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"])) if "meta" in z.files else {}
    return z, meta


def solver_fixture_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and f != "functions.npz")


def fix_kwargs(kw):
    """JSON turned tuples into lists; the solvers index them, so either works,
    but keep tuples for fidelity."""
    out = {}
    for k, v in kw.items():
        out[k] = tuple(v) if isinstance(v, list) else v
    return out


@pytest.fixture(scope="session")
def golden_loader():
    return load_golden
