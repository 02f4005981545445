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


# NMFX_RECORD_BARS=<file>: every np.testing.assert_allclose of the run appends (test id, rtol asked, largest relative
# difference seen) to the file -- how far inside its bar each comparison sits (used to set the bars).
_BARS = os.environ.get("NMFX_RECORD_BARS")
if _BARS:
    _orig_allclose = np.testing.assert_allclose

    def _recording_allclose(actual, desired, rtol=1e-7, atol=0, *args, **kwargs):
        try:
            a, d = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
            if a.shape == d.shape and a.size:
                with np.errstate(all="ignore"):
                    rel = np.abs(a - d) / np.maximum(np.abs(d), 1e-300)
                worst = float(np.nanmax(np.where(np.abs(d) > 0, rel, 0.0)))
                test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
                with open(_BARS, "a") as fh:
                    fh.write(json.dumps({"test": test, "rtol": rtol, "atol": atol, "worst_rel": worst, "n": int(a.size)}) + "\n")
        except Exception:  # noqa: BLE001  (a diagnostic must never change a test's outcome)
            pass
        return _orig_allclose(actual, desired, rtol=rtol, atol=atol, *args, **kwargs)

    np.testing.assert_allclose = _recording_allclose


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(worker, args, nprocs):
    """torch.multiprocessing.spawn of `worker(rank, *args)`; args[1] (pass None there) becomes the rendezvous: a `file://`
    init_method in a fresh temporary directory.  (Round 2 probed a free TCP port in the parent and let rank 0 bind it later;
    one run in ~2000 lost the port in between and died with EADDRINUSE.  A FileStore has no such window.)"""
    import shutil
    import tempfile
    import torch.multiprocessing as mp
    d = tempfile.mkdtemp(prefix="nmfx_rdzv_")
    try:
        a = list(args)
        a[1] = "file://" + os.path.join(d, "store")
        mp.spawn(worker, args=tuple(a), nprocs=nprocs, join=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
