"""ANLS on the HIP engine vs the reference's golden outputs (scipy nnls and
FCNNLS paths: same minimisers)."""
import numpy as np
import pytest

from gpu_common import WH_TOL, run_fixture, snapshot_errors, wh_error

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["anls_nnls", "anls_fcnnls", "anls_lambda_random"])
def test_anls_matches_reference(name):
    from nmf_amd.anls import anls
    z, meta, v, res = run_fixture(name, anls)
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, z["w"], z["h"], v)
    snaps = snapshot_errors(name, anls) if err >= WH_TOL else {}
    assert err < WH_TOL, f"WH error {err:.3e}; per-snapshot {snaps}"
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=1e-3)
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
    ref = R.anls(v.astype(np.float64), k, **kw)
    res = anls(v.copy(), k, **kw)
    err = np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    assert res.i == ref.i
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=5e-4)
