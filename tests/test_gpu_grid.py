"""Pair mode: two MUR-Euclidean problems in one pass over V -- against separate calls and against the oracle."""
import numpy as np
import pytest

from gpu_common import WH_TOL, oracle_after, wh_error
from oracle import nmf_ref as R

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("shape,ka,kb", [((520, 300), 40, 24), ((384, 640), 64, 64), ((200, 130), 5, 33)])
def test_mur_pair_equals_two_separate_calls_and_the_oracle(shape, ka, kb):
    """nmf_amd.mur.mur_pair: problem 0 in the factor columns [0, 64), problem 1 in [64, 128) of a k = 128 engine; lambda, objective,
    Gram matrices, stop test per problem.  Same RNG draws as two consecutive mur() calls; each result within 1e-6 of its separate
    call (the k = 128 kernels sum in another order than the k = 64 ones) and inside north_star's bar against the f64 oracle."""
    from nmf_amd.mur import mur, mur_pair
    m, n = shape
    v = R.planted_matrix(m, n, 12, seed=m + ka, dtype=np.float32)
    par = [dict(k=ka, lambda_w=0.01, lambda_h=0.0), dict(k=kb, lambda_w=0.1, lambda_h=0.05)]
    kw = dict(min_iter=30, max_iter=30)
    np.random.seed(11)
    got = mur_pair(v.copy(), max(ka, kb), par, **kw)
    np.random.seed(11)
    sep = [mur(v.copy(), p["k"], distance_type="eu", lambda_w=p["lambda_w"], lambda_h=p["lambda_h"], **kw) for p in par]
    np.random.seed(11)
    ref = [R.mur(v.astype(np.float64), p["k"], distance_type="eu", lambda_w=p["lambda_w"], lambda_h=p["lambda_h"], **kw) for p in par]
    for g, s, r, p in zip(got, sep, ref, par):
        assert g.w.shape == (m, p["k"]) and g.h.shape == (p["k"], n)
        assert g.i == s.i == r.i and len(g.obj_history) == g.i + 2
        assert g.experiment == s.experiment
        assert _rel(g.w @ g.h, s.w @ s.h) < 1e-6, _rel(g.w @ g.h, s.w @ s.h)
        np.testing.assert_allclose(g.obj_history, s.obj_history, rtol=1e-5)      # (the objective of a 1 % residual is ~100 x as sensitive as WH: measured 1.5e-6)
        assert wh_error(g.w, g.h, r.w, r.h, v) < WH_TOL
        np.testing.assert_allclose(g.obj_history, r.obj_history, rtol=4e-5)


def test_mur_pair_problems_stop_independently():
    """The two problems of a pair meet the reference's stop rule (nmf/mur.py:131, nmf/utils.py:4-15) at DIFFERENT outer iterations:
    the one that stops first keeps the iterate the reference returns (its half of the W ping-pong buffer (stop_i + 1) & 1 is not
    written again, its rows of H are frozen) while the other one goes on; each equals its own run and the oracle's stop index."""
    from nmf_amd.mur import mur, mur_pair
    v = R.planted_matrix(300, 260, 36, seed=77, dtype=np.float32)
    par = [dict(k=36, lambda_w=0.0, lambda_h=0.0), dict(k=10, lambda_w=0.05, lambda_h=0.05)]
    kw = dict(min_iter=5, max_iter=600, tol1=1e-9, tol2=5e-3)
    np.random.seed(3)
    got = mur_pair(v.copy(), 36, par, **kw)
    np.random.seed(3)
    ref = [R.mur(v.astype(np.float64), p["k"], distance_type="eu", lambda_w=p["lambda_w"], lambda_h=p["lambda_h"], **kw) for p in par]
    np.random.seed(3)
    sep = [mur(v.copy(), p["k"], distance_type="eu", lambda_w=p["lambda_w"], lambda_h=p["lambda_h"], **kw) for p in par]
    assert ref[0].i != ref[1].i and max(ref[0].i, ref[1].i) < 599, (ref[0].i, ref[1].i)
    for g, s, r, p in zip(got, sep, ref, par):
        assert abs(g.i - r.i) <= 1 and len(g.obj_history) == g.i + 2, (g.i, s.i, r.i)
        if g.i == s.i:
            assert _rel(g.w @ g.h, s.w @ s.h) < 2e-6
        if g.i == r.i:
            assert wh_error(g.w, g.h, r.w, r.h, v) < WH_TOL
            np.testing.assert_allclose(g.obj_history, r.obj_history, rtol=4e-5)
    # whatever the index, the iterate is pinned: the oracle after exactly g.i + 1 iterations, both problems from the RNG stream
    # the pair consumed (the |i - i_ref| <= 1 branch used to skip the comparison)
    np.random.seed(3)
    kw_n = {key: val for key, val in kw.items() if key not in ("min_iter", "max_iter")}
    for g, p in zip(got, par):
        with np.errstate(all="ignore"):
            at = R.mur(v.astype(np.float64), p["k"], distance_type="eu", lambda_w=p["lambda_w"], lambda_h=p["lambda_h"],
                       min_iter=g.i + 1, max_iter=g.i + 1, **kw_n)
        assert wh_error(g.w, g.h, at.w, at.h, v) < WH_TOL
        np.testing.assert_allclose(g.obj_history, at.obj_history, rtol=4e-5)


def test_grid_runs_mur_eu_in_pairs_and_equals_the_oracle(capsys, monkeypatch):
    """nmf_amd.grid.factorize_grid for MUR-eu: combinations two at a time through mur_pair (one pass over V per pair), a
    combination left over on its own; order, results, RNG consumption and printed lines as the sequential grid -- compared with
    the f64 oracle run combination by combination (not only with the product's own separate calls)."""
    from nmf_amd.grid import factorize_grid
    monkeypatch.delenv("NMF_AMD_QUIET", raising=False)
    v = R.planted_matrix(260, 180, 6, seed=4, dtype=np.float32)
    common = dict(distance_type="eu", min_iter=12, max_iter=12, nndsvd_init=(False, "zero"))
    np.random.seed(5)
    runs = factorize_grid(v.copy(), "mur", features=(6, 9, 70), lambda_w=(0.0, 0.1), lambda_h=(0.05,), **common)
    paired_out = capsys.readouterr().out
    assert [(p["features"], p["lambda_w"]) for p, _ in runs] == [(6, 0.0), (6, 0.1), (9, 0.0), (9, 0.1), (70, 0.0), (70, 0.1)]
    np.random.seed(5)
    for params, res in runs:
        ref = R.mur(v.astype(np.float64), params["features"], lambda_w=params["lambda_w"], lambda_h=params["lambda_h"], **common)
        assert res.i == ref.i and res.w.shape == ref.w.shape
        assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
        np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=4e-5)
    monkeypatch.setenv("NMFX_GRID_PAIR", "0")
    np.random.seed(5)
    seq = factorize_grid(v.copy(), "mur", features=(6, 9, 70), lambda_w=(0.0, 0.1), lambda_h=(0.05,), **common)
    seq_out = capsys.readouterr().out
    for (pa, ra), (pb, rb) in zip(runs, seq):
        assert pa == pb and ra.i == rb.i
        assert _rel(ra.w @ ra.h, rb.w @ rb.h) < 1e-6
    strip = lambda text: [ln.split(":")[0] for ln in text.splitlines() if ln.startswith("[")]      # noqa: E731
    assert strip(paired_out) == strip(seq_out) and len(strip(seq_out)) == 6 * 12


def test_pair_mode_says_so_where_it_does_not_apply(monkeypatch):
    """The pair entry points need a k = 128 handle on the split-bf16 path and fresh factors; the grid then runs one by one."""
    from nmf_amd._lib import NmfxError
    from nmf_amd.engine import Engine
    from nmf_amd.grid import factorize_grid
    v = R.planted_matrix(260, 180, 6, seed=4, dtype=np.float32)
    rs = np.random.RandomState(0)
    with Engine(260, 180, 64) as e:                      # not a k = 128 handle
        e.upload_v(v)
        e.set_factors(np.abs(rs.randn(260, 64)), np.abs(rs.randn(64, 180)))
        with pytest.raises(NmfxError, match="k = 128"):
            e.mur_pair_run([0, 0], [0, 0], 10 ** 9, 1e-5, 1e-5, 0, 1)
    with Engine(260, 180, 128) as e:
        e.upload_v(v)
        e.set_factors(np.abs(rs.randn(260, 128)), np.abs(rs.randn(128, 180)))
        e.mur_run(0, 0.0, 0.0, 10 ** 9, 1e-5, 1e-5, 0, 2)       # started as ONE k = 128 problem ...
        with pytest.raises(NmfxError, match="set_factors"):
            e.mur_pair_run([0, 0], [0, 0], 10 ** 9, 1e-5, 1e-5, 2, 1)      # ... cannot continue as a pair
        w, h = e.get_factors()
        e.set_factors(w, h)
        e.mur_pair_run([0, 0.1], [0, 0.1], 10 ** 9, 1e-5, 1e-5, 0, 2)
        with pytest.raises(NmfxError, match="two stacked problems"):
            e.mur_run(0, 0.0, 0.0, 10 ** 9, 1e-5, 1e-5, 2, 1)
    monkeypatch.setenv("NMFX_PRECISION", "f32")               # exact-f32 engines: no pair mode, the grid runs singly and still equals the oracle
    common = dict(distance_type="eu", min_iter=6, max_iter=6, nndsvd_init=(False, "zero"))
    np.random.seed(5)
    runs = factorize_grid(v.copy(), "mur", features=(6,), lambda_w=(0.0, 0.1), lambda_h=(0.05,), **common)
    np.random.seed(5)
    for params, res in runs:
        ref = R.mur(v.astype(np.float64), params["features"], lambda_w=params["lambda_w"], lambda_h=params["lambda_h"], **common)
        assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
