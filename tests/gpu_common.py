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


# ---- oracle results that take tens of seconds (scipy's per-column NNLS beyond 100 components) --------------------------------
# The oracle run of such a case is committed as data under tests/golden/slow/ (made by oracle/make_slow_cases.py from the very
# functions the tests call: same seeded input, same keywords) and only re-computed when the file is missing or was made for other
# inputs -- the GPU suite has a 900 s limit and spent 170 s of it in these.
SLOW_DIR = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "slow")


class SlowOutcome:
    def __init__(self, w, h, i, obj_history):
        self.w, self.h, self.i, self.obj_history = w, h, int(i), list(np.asarray(obj_history, dtype=np.float64))


def slow_signature(v, k, kw, w0=None, h0=None):
    """What pins a case: shape, rank, keywords and checksums of the inputs (JSON text)."""
    import json
    sig = {"shape": list(v.shape), "k": int(k), "kw": {key: (list(val) if isinstance(val, tuple) else val) for key, val in sorted(kw.items())},
           "vsum": repr(float(np.sum(np.asarray(v, dtype=np.float64))))}
    if w0 is not None:
        sig["w0sum"], sig["h0sum"] = repr(float(np.sum(w0))), repr(float(np.sum(h0)))
    return json.dumps(sig, sort_keys=True)


def slow_oracle(name, signature, compute):
    """The oracle's outcome for the case `name`: from tests/golden/slow/<name>.npz when that was made for `signature`, else
    compute() (and, with NMFX_WRITE_SLOW_ORACLE=1 -- oracle/make_slow_cases.py --, written there)."""
    import os
    path = os.path.join(SLOW_DIR, name + ".npz")
    if os.path.exists(path) and os.environ.get("NMFX_WRITE_SLOW_ORACLE") != "1":
        z = np.load(path, allow_pickle=False)
        if str(z["signature"]) == signature:
            return SlowOutcome(z["w"], z["h"], z["i"], z["obj_history"])
    with np.errstate(all="ignore"):
        out = compute()
    if os.environ.get("NMFX_WRITE_SLOW_ORACLE") == "1":
        os.makedirs(SLOW_DIR, exist_ok=True)
        np.savez_compressed(path, w=out.w, h=out.h, i=out.i, obj_history=np.asarray(out.obj_history, dtype=np.float64), signature=signature)
    return SlowOutcome(out.w, out.h, out.i, out.obj_history)
