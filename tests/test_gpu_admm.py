"""ADMM on the HIP engine vs the reference's golden outputs."""
import numpy as np
import pytest

from gpu_common import WH_TOL, run_fixture, snapshot_errors, wh_error

pytestmark = pytest.mark.gpu

CASES = ["admm_eu_nn", "admm_eu_l1n", "admm_eu_l2n", "admm_kl_nn"]


@pytest.mark.parametrize("name", CASES)
def test_admm_eu_matches_reference(name):
    from nmf_amd.admm import admm
    z, meta, v, res = run_fixture(name, admm)
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, z["w"], z["h"], v)
    snaps = snapshot_errors(name, admm) if err >= WH_TOL else {}
    assert err < WH_TOL, f"WH error {err:.3e}; per-snapshot {snaps}"
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=5e-5 if "kl" in name else 1e-4)      # (measured: 6.8e-6 / 1.2e-5)
    assert res.experiment.rho == meta["kwargs"]["rho"]


def test_admm_default_l2n_runs_like_reference():
    """reg_h defaults to (0, 'l2n'): with lambda = 0 the operator is the identity."""
    from nmf_amd.admm import admm
    from oracle import nmf_ref as R
    v = R.planted_matrix(80, 64, 4, seed=9, dtype=np.float64)
    res = admm(v, 4, min_iter=6, max_iter=6)
    ref = R.admm(v.copy(), 4, min_iter=6, max_iter=6)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("shape,reg_h", [((520, 300, 40), (0.05, "l1n")), ((384, 640, 100), (0.1, "l2n"))])
def test_admm_eu_k64_k128_both_precisions_vs_oracle(precision, shape, reg_h, monkeypatch):
    """k in (32, 128]: ADMM with the products on the split-bf16 kernels (objective by one more pass
    of the same kernel) against the exact-f32 products and the oracle."""
    from oracle import nmf_ref as R
    from nmf_amd.admm import admm
    monkeypatch.setenv("NMFX_PRECISION", precision)
    m, n, k = shape
    v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
    kw = dict(rho=1.0, distance_type="eu", reg_w=(0.02, "l1n"), reg_h=reg_h, min_iter=12, max_iter=12,
              nndsvd_init=(True, "zero"))
    ref = R.admm(v.astype(np.float64), k, **kw)
    res = admm(v.copy(), k, **kw)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
    assert res.i == ref.i and len(res.obj_history) == len(ref.obj_history)
    # The objective of a small residual is far more sensitive than WH: d(obj) / obj <= 2 ||dWH|| / ||V - WH||, and ||V - WH|| is
    # 3.3 % of ||V|| here, so the measured 2.4e-4 (split bf16) / 1.1e-4 (exact f32) IS a WH error of 7e-6 .. 1e-5 (asserted
    # above at north_star's 1e-4).  tools/lab/admm_f32_state.py: with rho fixed at 1 the systems are as ill-conditioned as the
    # Gram matrix, and ALL-f32 arithmetic in numpy (f64 everywhere else) already moves the objective by 2.3e-4 -- the f32
    # accumulation of the V-sized products (1.8e-4 alone) and of M^-1 rhs (9e-5), not a defect of a kernel.
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=5e-4)


@pytest.mark.parametrize("shape,reg_w,reg_h,rho,iters", [
    ((200, 160, 12), (0.05, "l1n"), (0.05, "l1n"), 1.0, 12),
    ((320, 256, 40), (0, "nn"), (0.05, "l1n"), 2.0, 10),           # k padded to 64
    ((256, 384, 100), (0.05, "l2n"), (0, "nn"), 1.0, 8),           # k padded to 128, the l2n operator on W
])
def test_admm_kl_vs_oracle(shape, reg_w, reg_h, rho, iters):
    """ADMM with the KL loss (admm.py: v_aux and its dual next to the factor auxiliaries) beyond the one golden fixture:
    the three prox operators of W, k padded to 16 / 64 / 128, against the oracle.  (Measured: WH 9e-7 .. 3e-6, objective
    3e-6 .. 4.3e-5.)"""
    from oracle import nmf_ref as R
    from nmf_amd.admm import admm
    m, n, k = shape
    v = R.planted_matrix(m, n, min(k, 32), seed=m + n + k, dtype=np.float32)
    kw = dict(rho=rho, distance_type="kl", reg_w=reg_w, reg_h=reg_h, min_iter=iters, max_iter=iters, nndsvd_init=(True, "zero"))
    with np.errstate(all="ignore"):
        ref = R.admm(v.astype(np.float64), k, **kw)
    res = admm(v.copy(), k, **kw)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < 2e-5
    assert res.i == ref.i and len(res.obj_history) == len(ref.obj_history)
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=2e-4)
