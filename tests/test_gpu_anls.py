"""ANLS on the HIP engine vs the reference's golden outputs (scipy nnls and
FCNNLS paths: same minimisers)."""
import numpy as np
import pytest

from gpu_common import WH_TOL, run_fixture, slow_oracle, slow_signature, snapshot_errors, wh_error

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["anls_nnls", "anls_fcnnls", "anls_lambda_random", "anls_kl"])
def test_anls_matches_reference(name):
    from nmf_amd.anls import anls
    z, meta, v, res = run_fixture(name, anls)
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, z["w"], z["h"], v)
    snaps = snapshot_errors(name, anls) if err >= WH_TOL else {}
    assert err < WH_TOL, f"WH error {err:.3e}; per-snapshot {snaps}"
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=1e-5)      # (measured: 7.5e-7)
    assert (res.w >= 0).all() and (res.h >= 0).all()
    # exact zeros of the active set are exact zeros here too (NNLS, not a projection)
    assert ((z["h"] == 0) == (res.h == 0)).mean() > 0.97
    assert res.experiment.fcnnls == bool(meta["kwargs"].get("use_fcnnls", False))


def test_anls_larger_rank_properties():
    """k = 40 (> 32, one variable per lane, 64-wide workspace): KKT conditions of the
    returned H against the returned W, checked in float64 on the host."""
    from nmf_amd.anls import anls
    from oracle import nmf_ref as R
    v = R.planted_matrix(300, 200, 40, seed=2, dtype=np.float64)
    np.random.seed(1)
    res = anls(v, 40, min_iter=2, max_iter=2, lambda_h=0.05, nndsvd_init=(False, "zero"))
    g = res.w.T @ res.w + 2 * 0.05 * np.eye(40)
    y = g @ res.h - res.w.T @ v                     # dual variables
    scale = np.abs(res.w.T @ v).max()
    assert (res.h >= 0).all()
    assert y[res.h == 0].min() > -2e-4 * scale      # dual feasibility on the active set
    assert np.abs(y[res.h > 0]).max() < 2e-4 * scale   # stationarity on the passive set
    # (with lambda_h > 0 the plain objective need not decrease; the KKT system above is the exact test)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(300, 220, 40), (260, 400, 100)])
def test_anls_k64_k128_both_precisions_vs_oracle(precision, shape, monkeypatch):
    """k in (32, 128]: register-resident NNLS (k <= 64) / LDS NNLS (k = 128), products on the
    split-bf16 kernels or on the exact-f32 kernels, against the oracle (scipy NNLS)."""
    from oracle import nmf_ref as R
    from nmf_amd.anls import anls
    monkeypatch.setenv("NMFX_PRECISION", precision)
    m, n, k = shape
    v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
    kw = dict(lambda_w=0.05, lambda_h=0.02, min_iter=4, max_iter=4, nndsvd_init=(True, "zero"))
    # (k = 100: 12 s of scipy NNLS per run -- committed under tests/golden/slow/, see gpu_common.slow_oracle)
    ref = slow_oracle(f"anls_{m}x{n}_k{k}", slow_signature(v, k, kw), lambda: R.anls(v.astype(np.float64), k, **kw))
    res = anls(v.copy(), k, **kw)
    err = np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    assert res.i == ref.i
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=2e-4)      # (measured: 7.6e-5 split bf16, 5.0e-5 exact f32)


@pytest.mark.parametrize("k", [6, 40, 100])
@pytest.mark.parametrize("case", ["dead", "collinear"])
def test_anls_rank_deficient_passive_set_at_lambda_zero(k, case):
    """lambda_w = lambda_h = 0 with a singular Gram matrix on the warm-started passive set: a component
    whose row of H is zero while its column of W is still positive (dead), or two identical rows of H (collinear).
    The unguarded 1 / pivot turned the whole right-hand side into NaN and the solve silently returned zeros; scipy's
    Lawson-Hanson (the oracle) keeps such a variable at zero.  All three NNLS kernel shapes (k padded to 16, 64, 128).
    dead: the component stays dead in the reference, so the runs are comparable.  collinear: the NNLS minimiser is not
    unique (only the sum of the two twin variables is determined), the solvers split it differently and the iterates
    part ways -- asserted instead: everything finite and non-negative, the first W step reproduces the oracle's product
    W H0, and H is a KKT point for the returned W."""
    from nmf_amd.engine import Engine
    from oracle import nmf_ref as R
    (m, n), iters = {6: (200, 150), 40: (300, 260), 100: (500, 700)}[k], 3
    v = R.planted_matrix(m, n, k, seed=k, dtype=np.float32)
    rs = np.random.RandomState(k)
    w0, h0 = rs.rand(m, k), rs.rand(k, n)
    if case == "dead":
        h0[2] = 0.0
    else:
        h0[3] = h0[1]
    with Engine(m, n, k) as eng:
        eng.upload_v(v)
        eng.set_factors(w0, h0)
        eng.anls_set_distance(0)
        eng.anls_run(0.0, 0.0, 10 ** 9, 1e-3, 1e-3, 0, iters)
        eng.aoadmm_finish(10 ** 9, 1e-3, 1e-3, iters)
        w, h = eng.get_factors()
        _, _, n_obj = eng.state()
        obj = eng.objectives(0, n_obj)
        evicted, capped = eng.diagnostics()
    assert np.isfinite(w).all() and np.isfinite(h).all() and (w >= 0).all() and (h >= 0).all()
    assert evicted > 0 and capped == 0
    assert np.isfinite(obj).all() and obj[-1] < obj[0]
    vd = v.astype(np.float64)
    if case == "dead":
        kw = dict(lambda_w=0, lambda_h=0, min_iter=iters, max_iter=iters)
        ref = slow_oracle(f"anls_dead_{m}x{n}_k{k}", slow_signature(v, k, kw, w0, h0), lambda: R.anls(vd, k, w0=w0, h0=h0, **kw))
        assert wh_error(w, h, ref.w, ref.h, v) < WH_TOL
        np.testing.assert_allclose(obj, ref.obj_history, rtol=1e-5)      # (measured: 5e-7)
        assert not w[:, 2].any() and not h[2].any()          # the component stays dead, as in the reference
    else:
        y = (w.T @ w) @ h - w.T @ vd                         # duals of the H sub-problem for the returned W
        scale = np.abs(w.T @ vd).max()
        assert y[h == 0].min() > -5e-4 * scale and np.abs(y[h > 0]).max() < 5e-4 * scale
        with Engine(m, n, k) as eng:                         # one W step alone: W1 H0 is unique although W1 is not
            eng.upload_v(v)
            eng.set_factors(w0, h0)
            eng.anls_phase_objective(0)
            eng.anls_phase_w(0.0, 10 ** 9, 1e-3, 1e-3, 0)
            w1, _ = eng.get_factors()
        w1_ref = R.anls_w_step(vd, h0, 0)
        assert np.linalg.norm(w1 @ h0 - w1_ref @ h0) / np.linalg.norm(vd) < WH_TOL
