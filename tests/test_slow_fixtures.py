"""tests/golden/slow/*.npz (oracle outcomes of the GPU suite's slowest oracle legs, oracle/make_slow_cases.py): every file carries
its input signature, and the small case is recomputed here and must come back bit for bit."""
import json
import os

import numpy as np

from gpu_common import SLOW_DIR, slow_oracle, slow_signature
from oracle import nmf_ref as R


def test_every_slow_fixture_is_well_formed():
    names = sorted(f for f in os.listdir(SLOW_DIR) if f.endswith(".npz"))
    assert len(names) >= 8
    for f in names:
        z = np.load(os.path.join(SLOW_DIR, f), allow_pickle=False)
        sig = json.loads(str(z["signature"]))
        m, n = sig["shape"]
        k = sig["k"]
        assert z["w"].shape == (m, k) and z["h"].shape == (k, n)
        assert len(z["obj_history"]) == int(z["i"]) + 2
        assert np.isfinite(z["w"]).all() and np.isfinite(z["h"]).all() and (z["w"] >= 0).all() and (z["h"] >= 0).all()


def test_small_slow_fixture_equals_a_fresh_oracle_run(monkeypatch):
    monkeypatch.delenv("NMFX_WRITE_SLOW_ORACLE", raising=False)
    k, (m, n), iters = 6, (200, 150), 3
    v = R.planted_matrix(m, n, k, seed=k, dtype=np.float32)
    rs = np.random.RandomState(k)
    w0, h0 = rs.rand(m, k), rs.rand(k, n)
    h0[2] = 0.0
    kw = dict(lambda_w=0, lambda_h=0, min_iter=iters, max_iter=iters)
    sig = slow_signature(v, k, kw, w0, h0)
    calls = []
    got = slow_oracle(f"anls_dead_{m}x{n}_k{k}", sig, lambda: calls.append(1) or R.anls(v.astype(np.float64), k, w0=w0, h0=h0, **kw))
    assert not calls                                             # served from the file: the signature matches this definition
    fresh = R.anls(v.astype(np.float64), k, w0=w0, h0=h0, **kw)
    np.testing.assert_array_equal(got.w, fresh.w)
    np.testing.assert_array_equal(got.h, fresh.h)
    assert got.i == fresh.i and list(got.obj_history) == list(fresh.obj_history)
    # another input -> the file is not used
    other = slow_oracle(f"anls_dead_{m}x{n}_k{k}", slow_signature(v + 1, k, kw, w0, h0),
                        lambda: calls.append(1) or R.anls(v.astype(np.float64) + 1, k, w0=w0, h0=h0, **kw))
    assert calls == [1] and not np.array_equal(other.w, got.w)
