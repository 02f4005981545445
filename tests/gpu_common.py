"""Helpers for the GPU parity tests (HIP path vs oracle / golden fixtures)."""
import numpy as np

from conftest import fix_kwargs, load_golden
from oracle import nmf_ref as R

WH_TOL = 1e-4     # north_star: ||W_g H_g - W_r H_r||_F / ||V||_F < 1e-4


def _record_wh(err):
    """NMFX_RECORD_BARS=<file> (tests/conftest.py): the WH errors go to the same file as the assert_allclose records."""
    import json
    import os
    path = os.environ.get("NMFX_RECORD_BARS")
    if path:
        with open(path, "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "wh": float(err)}) + "\n")
    return err


def wh_error(w, h, w_ref, h_ref, v):
    return _record_wh(np.linalg.norm(w @ h - w_ref @ h_ref) / np.linalg.norm(np.asarray(v, dtype=np.float64)))


def run_fixture(name, solver, **override):
    z, meta = load_golden(name)
    v = R.fixture_matrix(meta["vspec"])
    kw = fix_kwargs(meta["kwargs"])
    kw.update(override)
    np.random.seed(meta["seed"])
    res = solver(v, meta["k"], **kw)
    return z, meta, v, res


def snapshot_errors(name, solver):
    """WH error after 1, 2, 10 ... iterations (localises a divergence)."""
    z, meta = load_golden(name)
    out = {}
    for key in z.files:
        if key.startswith("snap") and key.endswith("_w"):
            s = int(key[4:-2])
            v = R.fixture_matrix(meta["vspec"])
            kw = fix_kwargs(meta["kwargs"])
            kw.update(max_iter=s, min_iter=s + 5)
            np.random.seed(meta["seed"])
            res = solver(v, meta["k"], **kw)
            out[s] = wh_error(res.w, res.h, z[f"snap{s}_w"], z[f"snap{s}_h"], v)
    return out


def wh_error_blocked(w, h, w_ref, h_ref, v, block=2048):
    """wh_error without m x n float64 temporaries (full-size configs): row blocks."""
    num = den = 0.0
    for a in range(0, v.shape[0], block):
        b = min(v.shape[0], a + block)
        d = w[a:b] @ h - w_ref[a:b] @ h_ref
        num += float(np.sum(d * d))
        vb = np.asarray(v[a:b], dtype=np.float64)
        den += float(np.sum(vb * vb))
    return _record_wh(np.sqrt(num / den))


def direct_objective(v, w, h, kind="eu", block=2048):
    """nmf/utils.py:18-33 evaluated in float64 from the returned factors, by row blocks."""
    tot = 0.0
    for a in range(0, v.shape[0], block):
        b = min(v.shape[0], a + block)
        tot += float(R.objective(np.asarray(v[a:b], dtype=np.float64), w[a:b] @ h, kind))
    return tot


def oracle_after(ref_fn, v, k, seed, iterations, **kw):
    """(w, h, obj_history) of the oracle after exactly `iterations` outer iterations (its stop rule off): what a device run that
    stopped one iteration off the oracle's index is compared with -- the stop tests accept |i - i_ref| <= 1 where the firing
    decrease sits at the resolution of an f32-grade objective, and must still pin the iterate."""
    kw = dict(kw)
    kw.update(min_iter=iterations, max_iter=iterations)
    np.random.seed(seed)
    with np.errstate(all="ignore"):
        out = ref_fn(np.asarray(v, dtype=np.float64), k, **kw)
    return out.w, out.h, np.asarray(out.obj_history)
