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
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=2e-3 if "kl" in name else 5e-4)
    assert res.experiment.rho == meta["kwargs"]["rho"]


def test_admm_default_l2n_runs_like_reference():
    """reg_h defaults to (0, 'l2n'): with lambda = 0 the operator is the identity."""
    from nmf_amd.admm import admm
    from oracle import nmf_ref as R
    v = R.planted_matrix(80, 64, 4, seed=9, dtype=np.float64)
    res = admm(v, 4, min_iter=6, max_iter=6)
    ref = R.admm(v.copy(), 4, min_iter=6, max_iter=6)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
