"""AO-ADMM on the HIP engine vs the reference's golden outputs."""
import numpy as np
import pytest

from gpu_common import WH_TOL, run_fixture, snapshot_errors, wh_error

pytestmark = pytest.mark.gpu

OBJ_RTOL = 5e-5      # (measured on the goldens: 6.4e-6)

CASES = ["aoadmm_eu_nn_planted", "aoadmm_eu_l1n_planted", "aoadmm_eu_nn_uniform", "aoadmm_eu_l1n_uniform"]


@pytest.mark.parametrize("name", CASES)
def test_aoadmm_eu_matches_reference(name):
    from nmf_amd.ao_admm import ao_admm
    z, meta, v, res = run_fixture(name, ao_admm)
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, z["w"], z["h"], v)
    snaps = snapshot_errors(name, ao_admm) if err >= WH_TOL else {}
    assert err < WH_TOL, f"WH error {err:.3e}; per-snapshot {snaps}"
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=OBJ_RTOL)
    # inner iteration counts (device-side `terminate`) equal the reference's
    assert np.array_equal(ao_admm.last_inner_counts, z["inner"]), (ao_admm.last_inner_counts, z["inner"])
    assert res.experiment.prox_h == meta["kwargs"]["reg_h"][1]


def test_aoadmm_converges_at_reference_iteration():
    from nmf_amd.ao_admm import ao_admm
    z, meta, v, res = run_fixture("aoadmm_eu_converge", ao_admm)
    assert int(z["stop_rule"]) == 2
    assert res.i == int(z["i"]), (res.i, int(z["i"]))      # (the firing decrease clears tol2 by 34x the objective's rounding error)
    assert wh_error(res.w, res.h, z["w"], z["h"], v) < WH_TOL


def test_aoadmm_error_behaviour():
    from nmf_amd.ao_admm import ao_admm
    v = np.random.RandomState(0).rand(40, 30)
    with pytest.raises(ValueError):          # default reg_h = (0, 'l2n') raises in the reference too
        ao_admm(v, 3, max_iter=2, nndsvd_init=(False, "zero"))
    with pytest.raises(TypeError):
        ao_admm(v, 3, max_iter=2, reg_h=(0, "bogus"), nndsvd_init=(False, "zero"))
    import nmf_amd.utils as U
    # W = 0 -> Gram + rho I = 0 -> not PD (ao_admm.py:55): the scalar prepare kernel (k = 3) and the blocked
    # f64-MFMA one (k padded to 64 and to 128) must both report it
    for k, m in ((3, 40), (40, 96), (100, 160)):
        vv = np.random.RandomState(k).rand(m, 120)
        orig = U.initial_factors
        U.initial_factors = lambda x, kk, init, **kw: (np.zeros((x.shape[0], kk)), np.abs(np.random.randn(kk, x.shape[1])))
        try:
            with pytest.raises(np.linalg.LinAlgError):
                ao_admm(vv, k, max_iter=3, reg_h=(0, "nn"), nndsvd_init=(False, "zero"))
        finally:
            U.initial_factors = orig


@pytest.mark.parametrize("name", ["aoadmm_eu_w_l1inf", "aoadmm_eu_w_l1inf_t", "aoadmm_eu_h_l1inf_t",
                                  "aoadmm_kl_w_l1inf", "aoadmm_kl_w_l1inf_t", "aoadmm_kl_h_l1inf", "aoadmm_kl_h_l1inf_t"])      # (r5: the KL loss too)
def test_aoadmm_l1inf_runs_like_the_reference_until_its_cholesky_fails(name):
    """SURVEY a12 (nmf/ao_admm.py:143-195): with 'l1inf' / 'l1inf_transpose' the reference completes 0-2 outer iterations -- the
    operator wipes a factor out -- and then raises scipy's LinAlgError in the next Cholesky factorisation.  The fixtures are the
    reference's last completed run and the max_iter at which it raises; the device must return the former and raise at the latter."""
    from nmf_amd.ao_admm import ao_admm
    z, meta, v, res = run_fixture(name, ao_admm)
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    assert wh_error(res.w, res.h, z["w"], z["h"], v) < WH_TOL
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=OBJ_RTOL)
    assert np.array_equal(ao_admm.last_inner_counts, z["inner"]), (ao_admm.last_inner_counts, z["inner"])
    # a factor the reference left at exactly zero is exactly zero here too
    for got, ref in ((res.w, z["w"]), (res.h, z["h"])):
        if not np.any(ref):
            assert not np.any(got)
    bad = int(z["raises_at"])
    assert bad == int(z["i"]) + 2
    with pytest.raises(np.linalg.LinAlgError):
        run_fixture(name, ao_admm, min_iter=bad, max_iter=bad)


def test_aoadmm_l1inf_on_h_fails_in_the_first_outer_iteration_like_the_reference():
    """reg_h = 'l1inf' (oracle/make_golden.py asserts it for the reference): H = 0 after the first H sub-problem, H H^T + rho I = 0."""
    from nmf_amd.ao_admm import ao_admm
    from oracle import nmf_ref as R
    v = R.fixture_matrix(dict(kind="planted", rank=6, seed=21, m=96, n=80))
    kw = dict(distance_type="eu", reg_w=(0, "nn"), reg_h=(0.1, "l1inf"), min_iter=1, max_iter=1, admm_iter=10)
    np.random.seed(22)
    with pytest.raises(np.linalg.LinAlgError):
        with np.errstate(all="ignore"):
            R.ao_admm(v.astype(np.float64), 6, **kw)
    np.random.seed(22)
    with pytest.raises(np.linalg.LinAlgError):
        ao_admm(v.copy(), 6, **kw)


def test_aoadmm_kl_matches_reference():
    from nmf_amd.ao_admm import ao_admm
    z, meta, v, res = run_fixture("aoadmm_kl_nn", ao_admm)
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, z["w"], z["h"], v)
    snaps = snapshot_errors("aoadmm_kl_nn", ao_admm) if err >= WH_TOL else {}
    assert err < WH_TOL, f"WH error {err:.3e}; per-snapshot {snaps}"
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=2e-5)      # (measured: 1.9e-6)
    assert np.array_equal(ao_admm.last_inner_counts, z["inner"]), (ao_admm.last_inner_counts, z["inner"])


@pytest.mark.parametrize("shape,reg_w,reg_h,admm_iter,iters", [
    ((200, 160, 12), (0.05, "l1n"), (0.05, "l1n"), 8, 6),
    ((320, 256, 40), (0.05, "l1n"), (0, "nn"), 6, 5),          # k padded to 64
    ((256, 384, 100), (0, "nn"), (0.05, "l1n"), 5, 4),         # k padded to 128
])
def test_aoadmm_kl_vs_oracle(shape, reg_w, reg_h, admm_iter, iters):
    """AO-ADMM with the KL loss (ao_admm.py:71-101: the m x n auxiliaries and their duals inside the inner rounds) beyond the
    one golden fixture: both prox operators, k padded to 16 / 64 / 128, against the oracle -- WH, objective history and the
    inner counts of every sub-problem.  (Measured: WH 4e-7 .. 1.4e-6, objective 6e-7 .. 4e-6.)"""
    from oracle import nmf_ref as R
    from nmf_amd.ao_admm import ao_admm
    m, n, k = shape
    v = R.planted_matrix(m, n, min(k, 32), seed=m + n + k, dtype=np.float32)
    kw = dict(distance_type="kl", reg_w=reg_w, reg_h=reg_h, min_iter=iters, max_iter=iters, admm_iter=admm_iter,
              nndsvd_init=(True, "zero"))
    with np.errstate(all="ignore"):
        ref = R.ao_admm(v.astype(np.float64), k, **kw)
    res = ao_admm(v.copy(), k, **kw)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < 1e-5
    assert res.i == ref.i and len(res.obj_history) == len(ref.obj_history)
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=5e-5)
    assert [tuple(r) for r in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]]


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(520, 300, 40), (384, 640, 100)])
def test_aoadmm_eu_k64_k128_both_precisions_vs_oracle(precision, shape, monkeypatch):
    """k in (32, 128]: the split-bf16 products (with the objective riding on the H-side product)
    and the exact-f32 products must meet the same bars: WH error, objective history, stop index
    and inner-iteration counts of the oracle."""
    from oracle import nmf_ref as R
    from nmf_amd.ao_admm import ao_admm
    monkeypatch.setenv("NMFX_PRECISION", precision)
    m, n, k = shape
    v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
    kw = dict(distance_type="eu", reg_w=(0.1, "l1n"), reg_h=(0.05, "l1n"), min_iter=7, max_iter=7, admm_iter=10,
              nndsvd_init=(True, "zero"))
    ref = R.ao_admm(v.astype(np.float64), k, **kw)
    res = ao_admm(v.copy(), k, **kw)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
    assert res.i == ref.i and len(res.obj_history) == len(ref.obj_history)
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=1e-4)      # (measured: 1.3e-5)
    assert [tuple(r) for r in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]]


def test_aoadmm_bf16_convergence_stop_matches_oracle(monkeypatch):
    """The stop rule fires inside a batch: the lagged objective of the bf16 path must give the
    reference's stop index and leave the pair the reference returns."""
    from oracle import nmf_ref as R
    from nmf_amd.ao_admm import ao_admm
    monkeypatch.setenv("NMFX_PRECISION", "bf16")
    m, n, k = 400, 260, 36
    v = R.planted_matrix(m, n, k, seed=3, dtype=np.float32)
    kw = dict(distance_type="eu", reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=3, max_iter=200, admm_iter=10,
              tol1=1e-3, tol2=5e-2, nndsvd_init=(True, "zero"))
    ref = R.ao_admm(v.astype(np.float64), k, **kw)
    res = ao_admm(v.copy(), k, **kw)
    assert ref.trace["stop_rule"] and ref.i < 199
    assert res.i == ref.i and len(res.obj_history) == len(ref.obj_history)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
