"""prox 'l1inf' / 'l1inf_transpose' (nmf/admm.py:158-210) on the device: function level against the reference's own
outputs (tests/golden/functions.npz) and the oracle, then through ADMM against the reference's 1- and 2-iteration
snapshots (tests/golden/admm_eu_l1inf*.npz).  Runs only on a real MI355X; everything goes through the C ABI
(nmfx_set_matrix / nmfx_prox_apply / nmfx_get_factors, nmfx_admm_run)."""
import numpy as np
import pytest

from conftest import load_golden
from gpu_common import WH_TOL, snapshot_errors
from oracle import nmf_ref as R

pytestmark = pytest.mark.gpu


def device_prox(kind, aux, dual, rho, lam, side, update_dual=False):
    """prox(kind, aux, dual) with aux, dual of shape k x cols, evaluated on the H side (h_aux = aux) or on the W side
    (w_aux^T = aux, the call of admm.py:320)."""
    from nmf_amd.engine import Engine
    k, cols = aux.shape
    m, n = (7, cols) if side == "h" else (cols, 9)
    with Engine(m, n, k) as eng:
        eng.upload_v(np.zeros((m, n), dtype=np.float32))
        eng.set_factors(np.zeros((m, k)), np.zeros((k, n)))
        if side == "h":
            eng.set_matrix("h_aux", aux); eng.set_matrix("dual_h", dual)
        else:
            eng.set_matrix("w_aux", aux.T); eng.set_matrix("dual_w", dual.T)
        eng.prox_apply(side, kind, rho, lam, update_dual=update_dual)
        w, h = eng.get_factors()
        new_dual = eng.get_matrix("dual_h") if side == "h" else eng.get_matrix("dual_w").T
    return (h if side == "h" else w.T), new_dual


@pytest.mark.parametrize("side", ["h", "w"])
@pytest.mark.parametrize("kind,key", [("l1inf", "prox_l1inf"), ("l1inf_transpose", "prox_l1inf_t")])
@pytest.mark.parametrize("tag", ["", "_mixed"])
def test_prox_l1inf_matches_the_references_vectors(kind, key, tag, side):
    f, _ = load_golden("functions")
    aux = f["prox_l1inf_aux_mixed"] if tag else f["prox_aux"]
    got, _ = device_prox(kind, aux, f["prox_dual"], 2.5, 0.4, side)
    np.testing.assert_allclose(got, f[key + tag], rtol=2e-6, atol=2e-6)      # f32 state against f64 vectors


@pytest.mark.parametrize("side", ["h", "w"])
@pytest.mark.parametrize("kind", ["l1inf", "l1inf_transpose"])
@pytest.mark.parametrize("shape", [(40, 3000), (100, 777), (128, 4096)])
def test_prox_l1inf_long_vectors_vs_oracle(kind, shape, side):
    """Vectors of thousands of entries (the LDS bitonic sort with padding to a power of two), rows on both sides of the
    sum test, plus the fused dual update dual += x - x_aux (admm.py:321-322)."""
    k, cols = shape
    rs = np.random.RandomState(k + cols)
    scale = np.where(np.arange(k) % 3 == 0, 1e-4, 1.0)[:, None]
    aux = (rs.randn(k, cols) * scale).astype(np.float32).astype(np.float64)
    dual = (0.3 * rs.randn(k, cols) * scale).astype(np.float32).astype(np.float64)
    if kind == "l1inf_transpose":          # columns: make some of them pass the sum test
        aux[:, ::4] *= 1e-3; dual[:, ::4] *= 1e-3
        aux = aux.astype(np.float32).astype(np.float64); dual = dual.astype(np.float32).astype(np.float64)
    want = R.prox(kind, aux, dual, rho=1.7, lam=0.3)
    got, new_dual = device_prox(kind, aux, dual, 1.7, 0.3, side, update_dual=True)
    assert (want == 0).any() and (want > 0).any()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(want).max()))
    np.testing.assert_allclose(new_dual, dual + want - aux, rtol=1e-4, atol=2e-5 * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("side,shape", [("h", (6, 40000)), ("w", (9, 70001))])
def test_prox_l1inf_vectors_beyond_32768_entries(side, shape):
    """r3 (missing 6): 'l1inf' vectors longer than the 32768 entries an LDS sort holds -- the W-side vectors of a tall
    matrix have m entries -- take the same kernel with its sort keys in a global work area."""
    k, cols = shape
    rs = np.random.RandomState(k + cols)
    scale = np.where(np.arange(k) % 3 == 0, 1e-5, 1.0)[:, None]
    aux = (rs.randn(k, cols) * scale).astype(np.float32).astype(np.float64)
    dual = (0.3 * rs.randn(k, cols) * scale).astype(np.float32).astype(np.float64)
    want = R.prox("l1inf", aux, dual, rho=1.7, lam=0.3)
    got, new_dual = device_prox("l1inf", aux, dual, 1.7, 0.3, side, update_dual=True)
    assert (want == 0).any() and (want > 0).any()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(want).max()))
    np.testing.assert_allclose(new_dual, dual + want - aux, rtol=1e-4, atol=2e-5 * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("name", ["admm_eu_l1inf", "admm_eu_l1inf_t"])
def test_admm_with_l1inf_follows_the_reference_over_the_first_iterations(name, capsys):
    """As written the operator makes the ADMM iteration expansive, so parity is asserted where f32 against
    f64 can still hold it: after 1 and 2 iterations (the fixtures' snapshots) and on the first objective values."""
    from nmf_amd.admm import admm
    errs = snapshot_errors(name, admm)
    assert set(errs) == {1, 2}
    assert all(e < WH_TOL for e in errs.values()), errs
    z, meta = load_golden(name)
    v = R.fixture_matrix(meta["vspec"])
    kw = {k_: (tuple(x) if isinstance(x, list) else x) for k_, x in meta["kwargs"].items()}
    kw.update(min_iter=3, max_iter=3)
    np.random.seed(meta["seed"])
    import nmf_amd.utils as U
    quiet = U.QUIET
    U.QUIET = False
    capsys.readouterr()
    try:
        res = admm(v, meta["k"], **kw)
    finally:
        U.QUIET = quiet
    np.testing.assert_allclose(res.obj_history[:3], z["obj_history"][:3], rtol=2e-5)      # (measured: 1.3e-6)
    out = capsys.readouterr().out
    if name.endswith("_t"):                # the reference's print inside the transposed branch (admm.py:190), H then W
        assert out.count("will go 96") == 3 and out.count("will go 128") == 3
        assert out.index("will go 96") < out.index("will go 128") < out.index("[0]: ")
