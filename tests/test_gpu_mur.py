"""MUR on the HIP engine vs the reference's golden outputs and the oracle.
Runs only on a real MI355X (`-m gpu`); everything goes through the C ABI."""
import numpy as np
import pytest

from gpu_common import WH_TOL, oracle_after, run_fixture, snapshot_errors, wh_error
from oracle import nmf_ref as R

pytestmark = pytest.mark.gpu

OBJ_RTOL = 4e-5   # f32 MFMA factors vs the f64 reference, objective summed in f64 (measured: 1e-7 on the goldens, 3.8e-6 at most)

EU = ["mur_eu_cfg1_random", "mur_eu_cfg1_nndsvdz", "mur_eu_lambda", "mur_eu_f32v",
      "mur_eu_signed", "mur_eu_ragged"]


@pytest.mark.parametrize("name", EU)
def test_mur_eu_matches_reference(name):
    from nmf_amd.mur import mur
    z, meta, v, res = run_fixture(name, mur)
    assert res.w.dtype == np.float64 and res.h.dtype == np.float64
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, z["w"], z["h"], v)
    snaps = snapshot_errors(name, mur) if err >= WH_TOL else {}
    assert err < WH_TOL, f"WH error {err:.3e}; per-snapshot {snaps}"
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=OBJ_RTOL)
    assert (res.w >= 0).all() and (res.h >= 0).all()
    assert res.experiment.method == "mur" and res.experiment.components == meta["k"]
    # the in-place lift of negative data (mur.py:99-101) happened on OUR array too
    assert np.isclose(np.asarray(v, dtype=np.float64).sum(), float(z["v_after_sum"]), rtol=1e-6)


def test_mur_eu_stops_at_the_reference_iteration():
    from nmf_amd.mur import mur
    z, meta, v, res = run_fixture("mur_eu_converge", mur)
    assert int(z["stop_rule"]) == 2
    assert res.i == int(z["i"]), (res.i, int(z["i"]))
    assert len(res.obj_history) == res.i + 2
    assert wh_error(res.w, res.h, z["w"], z["h"], v) < WH_TOL
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=OBJ_RTOL)


def test_class_api_sets_w_h_and_saves(tmp_path, capsys):
    from nmf_amd import NMF
    v = R.planted_matrix(96, 80, 4, seed=3, dtype=np.float64)
    np.random.seed(0)
    model = NMF(v, 4)
    model.factorize(method="mur", distance_type="eu", min_iter=5, max_iter=5)
    out = capsys.readouterr().out
    assert "[4]: " in out and "Factorization done." in out
    assert model.w is model.results.w and model.h.shape == (4, 80)
    model.save_factorization(save_dir=str(tmp_path))
    saved = np.load(tmp_path / "nmf_mur_4_eu_0.0_0.0_random.npz", allow_pickle=True)
    assert sorted(saved.files) == ["experiment", "h", "i", "obj_history", "w"]
    np.random.seed(0)
    ref = R.mur(v.copy(), 4, distance_type="eu", min_iter=5, max_iter=5)
    assert wh_error(model.w, model.h, ref.w, ref.h, v) < WH_TOL


def test_large_shape_properties():
    """At a size the oracle cannot follow: monotone objective, non-negativity,
    objective equals the directly evaluated residual."""
    from nmf_amd.mur import mur
    v = R.planted_matrix(4096, 2048, 64, seed=0, dtype=np.float32)
    np.random.seed(0)
    res = mur(v, 64, distance_type="eu", min_iter=20, max_iter=20)
    obj = np.asarray(res.obj_history)
    assert np.all(np.diff(obj) < 0), "MUR-eu objective must decrease monotonically"
    direct = 0.5 * np.sum((v.astype(np.float64) - res.w @ res.h) ** 2)
    assert abs(direct - obj[-1]) <= 1e-5 * direct
    assert (res.w >= 0).all() and (res.h >= 0).all()


KL = ["mur_kl", "mur_kl_lambda", "mur_kl_sparse"]


@pytest.mark.parametrize("name", KL)
def test_mur_kl_matches_reference(name):
    from nmf_amd.mur import mur
    z, meta, v, res = run_fixture(name, mur)
    assert res.i == int(z["i"]) and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, z["w"], z["h"], v)
    snaps = snapshot_errors(name, mur) if err >= WH_TOL else {}
    assert err < WH_TOL, f"WH error {err:.3e}; per-snapshot {snaps}"
    # KL objective: sum of v log(v/wh) - v + wh with cancellation between terms; f32 log
    np.testing.assert_allclose(res.obj_history, z["obj_history"], rtol=2e-6)      # (measured: 9.5e-8)


def test_mur_kl_default_distance_is_kl_like_reference():
    from nmf_amd.mur import mur
    v = R.planted_matrix(128, 96, 4, seed=5, dtype=np.float64)
    np.random.seed(2)
    res = mur(v, 4, min_iter=3, max_iter=3)
    assert res.experiment.distance_type == "kl"
    np.random.seed(2)
    ref = R.mur(v.copy(), 4, min_iter=3, max_iter=3)
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(512, 384, 40), (640, 1000, 64), (200, 130, 33), (520, 700, 100), (384, 256, 128)])
def test_mur_eu_k64_both_precisions_vs_oracle(precision, shape, monkeypatch):
    """k in (32, 128] pads to 64 or 128, where the split-bf16 products are available
    (NMFX_PRECISION=bf16).  Both arithmetic modes must meet the same bars."""
    from nmf_amd.mur import mur
    monkeypatch.setenv("NMFX_PRECISION", precision)
    m, n, k = shape
    v = R.planted_matrix(m, n, k, seed=m + n, dtype=np.float32)
    np.random.seed(7)
    res = mur(v, k, distance_type="eu", min_iter=40, max_iter=40, lambda_w=0.01, lambda_h=0.0)
    np.random.seed(7)
    ref = R.mur(v.astype(np.float64), k, distance_type="eu", min_iter=40, max_iter=40, lambda_w=0.01, lambda_h=0.0)
    err = wh_error(res.w, res.h, ref.w, ref.h, v)
    assert err < WH_TOL, err
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=OBJ_RTOL)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(512, 384, 40), (300, 520, 100), (200, 130, 33)])
def test_mur_kl_k64_k128_both_precisions_vs_oracle(precision, shape, monkeypatch):
    """KL divergence with k in (32, 128]: the quotient products on the split-bf16 kernels (KL mode:
    product first, quotient in registers, permuted contraction) against the exact-f32 kernels and
    the oracle."""
    from nmf_amd.mur import mur
    monkeypatch.setenv("NMFX_PRECISION", precision)
    m, n, k = shape
    v = R.planted_matrix(m, n, k, seed=m + n, dtype=np.float32)
    kw = dict(distance_type="kl", min_iter=25, max_iter=25, lambda_w=0.02, lambda_h=0.01)
    np.random.seed(11)
    res = mur(v.copy(), k, **kw)
    np.random.seed(11)
    ref = R.mur(v.astype(np.float64), k, **kw)
    err = wh_error(res.w, res.h, ref.w, ref.h, v)
    assert err < WH_TOL, err
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=4e-5)      # (measured: 3.6e-6 split-bf16, 3.6e-7 exact-f32 products)


@pytest.mark.parametrize("k", [40, 100])
def test_mur_kl_zero_and_tiny_entries_take_the_exact_objective_path(k):
    """KL on data with exact zeros (utils.py:24 zeroes their 0 log 0) and entries far below 2^-5, where W H is small too: the
    chunks concerned leave the one-transcendental fast path of the split-bf16 KL kernels (k padded to 64: pipelined form; to 128:
    the one-register-set form) and take the exact expression."""
    from nmf_amd.mur import mur
    m, n = 384, 320
    rs = np.random.RandomState(k)
    v = R.planted_matrix(m, n, 12, seed=k, dtype=np.float32)
    v[rs.rand(m, n) < 0.3] = 0.0
    v[:, : n // 4] *= 1e-3                                   # a block of tiny entries: zy < 2^-5 there
    v[: m // 8] *= 1e-2
    kw = dict(distance_type="kl", min_iter=15, max_iter=15, lambda_w=0.0, lambda_h=0.01)
    np.random.seed(3)
    res = mur(v.copy(), k, **kw)
    np.random.seed(3)
    ref = R.mur(v.astype(np.float64), k, **kw)
    assert np.isfinite(res.obj_history).all()
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=4e-5)


def test_mur_eu_bf16_stop_index_matches_f32_and_oracle(monkeypatch):
    from nmf_amd.mur import mur
    v = R.planted_matrix(300, 260, 36, seed=77, dtype=np.float32)
    kw = dict(distance_type="eu", min_iter=5, max_iter=600, tol1=1e-9, tol2=5e-3)
    out = {}
    for precision in ("f32", "bf16"):
        monkeypatch.setenv("NMFX_PRECISION", precision)
        np.random.seed(3)
        out[precision] = mur(v.copy(), 36, **kw)
    np.random.seed(3)
    ref = R.mur(v.astype(np.float64), 36, **kw)
    assert ref.trace["stop_rule"] == 2 and ref.i < 599
    assert out["f32"].i == ref.i
    # this tol2 sits 2e-6 from the firing decrease (0.004998 against 5e-3): at the edge of what an f32-grade objective
    # resolves, so the split-bf16 mode may fire one iteration off here ...
    assert abs(out["bf16"].i - ref.i) <= 1, (out["bf16"].i, ref.i)
    # ... with the iterate it returns pinned either way: the oracle's after that many iterations
    w_o, h_o, obj_o = oracle_after(R.mur, v, 36, 3, out["bf16"].i + 1, **kw)
    assert wh_error(out["bf16"].w, out["bf16"].h, w_o, h_o, v) < WH_TOL
    np.testing.assert_allclose(out["bf16"].obj_history, obj_o, rtol=4e-5)
    # ... and must hit the reference's iteration exactly where the rule has a margin (tol2 between two consecutive
    # decreases, 1e-5 from either)
    dec = -np.diff(ref.obj_history)
    j = len(dec) - 3
    kw["tol2"] = float(0.5 * (dec[j] + dec[j + 1]))
    np.random.seed(3)
    ref2 = R.mur(v.astype(np.float64), 36, **kw)
    for precision in ("f32", "bf16"):
        monkeypatch.setenv("NMFX_PRECISION", precision)
        np.random.seed(3)
        assert mur(v.copy(), 36, **kw).i == ref2.i < ref.i


def test_grid_keeps_v_resident_and_matches_separate_calls(tmp_path):
    """nmf_amd.grid: one upload per `features` value, results identical to separate solver calls
    (same RNG draws in the same order), including a KL-ADMM run after an LS run on the SAME engine
    (set_factors must reset every piece of solver state)."""
    from nmf_amd.grid import factorize_grid
    from nmf_amd.ao_admm import ao_admm
    from nmf_amd.admm import admm
    v = R.planted_matrix(260, 180, 6, seed=4, dtype=np.float32)
    common = dict(distance_type="eu", min_iter=5, max_iter=5, nndsvd_init=(False, "zero"))
    np.random.seed(5)
    runs = factorize_grid(v.copy(), "ao_admm", features=(6, 9), lambda_w=(0.0, 0.1), lambda_h=(0.05,),
                          prox_w="l1n", prox_h="l1n", save_dir=str(tmp_path), **common)
    assert [p["features"] for p, _ in runs] == [6, 6, 9, 9] and [p["lambda_w"] for p, _ in runs] == [0.0, 0.1, 0.0, 0.1]
    np.random.seed(5)
    for params, res in runs:
        ref = ao_admm(v.copy(), params["features"], reg_w=(params["lambda_w"], "l1n"), reg_h=(params["lambda_h"], "l1n"), **common)
        np.testing.assert_array_equal(res.w, ref.w)
        np.testing.assert_array_equal(res.obj_history, ref.obj_history)
    assert len(list(tmp_path.iterdir())) == 4
    # KL after LS on one engine
    from nmf_amd.engine import Engine
    kw = dict(rho=1.0, reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=4, max_iter=4, nndsvd_init=(False, "zero"))
    with Engine(260, 180, 6) as eng:
        eng.upload_v(v)
        np.random.seed(8); admm(v.copy(), 6, distance_type="eu", engine=eng, **kw)
        np.random.seed(9); again = admm(v.copy(), 6, distance_type="kl", engine=eng, **kw)
        np.random.seed(9); twice = admm(v.copy(), 6, distance_type="kl", engine=eng, **kw)
    np.random.seed(9); fresh = admm(v.copy(), 6, distance_type="kl", **kw)
    np.testing.assert_array_equal(again.w, fresh.w)
    np.testing.assert_array_equal(twice.w, fresh.w)


def test_signed_data_with_a_resident_engine_is_lifted_on_the_device_too():
    """mur(x, k, engine=eng) with negative data: the in-place lift of nmf/mur.py:99-101 happens on the host
    array; the resident engine must see the lifted matrix as well."""
    from nmf_amd.engine import Engine
    from nmf_amd.mur import mur
    rs = np.random.RandomState(3)
    x = rs.rand(96, 80) - 0.2
    with Engine(96, 80, 5) as eng:
        eng.upload_v(x)
        np.random.seed(4)
        res = mur(x.copy(), 5, distance_type="eu", min_iter=8, max_iter=8, engine=eng)
    np.random.seed(4)
    ref = R.mur(x.copy(), 5, distance_type="eu", min_iter=8, max_iter=8)
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=OBJ_RTOL)


def test_mur_kl_after_another_solver_touched_h_on_the_same_handle():
    """The fused MUR-KL epilogues leave the row sums of H (as partials) and the images of H for the NEXT iteration
    (kl_h_epilogue_kernel); they are only trusted for the iteration right behind the one that wrote them.  A Euclidean
    iteration in between changes H: the KL iterations that follow must give what a fresh handle started from the same
    factors gives, bit for bit."""
    from nmf_amd.engine import Engine
    m, n, k = 384, 320, 40
    v = R.planted_matrix(m, n, k, seed=5, dtype=np.float32)
    rs = np.random.RandomState(1)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    NEVER = 10 ** 9
    with Engine(m, n, k) as a:
        a.upload_v(v)
        a.set_factors(w0, h0)
        a.mur_run(1, 0.01, 0.02, NEVER, 1e-5, 1e-5, 0, 3)          # KL, iterations 0..2
        a.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 3, 1)            # one Euclidean iteration: H changes behind the KL path's back
        w1, h1 = a.get_factors()
        a.mur_run(1, 0.01, 0.02, NEVER, 1e-5, 1e-5, 4, 3)          # KL again, iterations 4..6
        wa, ha = a.get_factors()
    with Engine(m, n, k) as b:
        b.upload_v(v)
        b.set_factors(w1, h1)
        b.mur_run(1, 0.01, 0.02, NEVER, 1e-5, 1e-5, 0, 3)
        wb, hb = b.get_factors()
    np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(ha, hb)


def test_a_second_solver_family_on_the_same_handle_needs_fresh_factors():
    """every solver family keeps its own device state next to W and H (MUR: W ping-pong, bf16 images, the KL epilogue's
    leftovers; the ADMM family: duals and auxiliaries; ANLS: warm-start supports).  A different family that continues in the middle
    of a run -- reusing contiguous iteration indices -- used to read the other's leftovers silently; it is refused now
    (NMFX_E_STATE) until the factors have been read back and set again, and the KL leftovers are voided by every other entry."""
    from nmf_amd._lib import NmfxError
    from nmf_amd.engine import Engine
    m, n, k = 384, 320, 40
    v = R.planted_matrix(m, n, k, seed=5, dtype=np.float32)
    rs = np.random.RandomState(1)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    NEVER = 10 ** 9
    with Engine(m, n, k) as a:
        a.upload_v(v)
        a.set_factors(w0, h0)
        a.mur_run(1, 0.01, 0.02, NEVER, 1e-5, 1e-5, 0, 3)          # KL, iterations 0..2
        with pytest.raises(NmfxError, match="another solver"):
            a.anls_run(0.0, 0.0, NEVER, 1e-3, 1e-3, 3, 2)          # ANLS "continuing" at index 3
        with pytest.raises(NmfxError, match="another solver"):
            a.aoadmm_run(0, 1, 0.1, 1, 0.1, 5, NEVER, 1e-3, 1e-3, 3, 1)
        w1, h1 = a.get_factors()
        a.set_factors(w1, h1)
        a.anls_run(0.0, 0.0, NEVER, 1e-3, 1e-3, 0, 3)              # ... from fresh factors: fine
        with pytest.raises(NmfxError, match="another solver"):
            a.mur_run(1, 0.01, 0.02, NEVER, 1e-5, 1e-5, 3, 2)      # and KL may not pick up behind ANLS at contiguous indices either
        w2, h2 = a.get_factors()
        a.set_factors(w2, h2)
        a.mur_run(1, 0.01, 0.02, NEVER, 1e-5, 1e-5, 0, 3)
        wa, ha = a.get_factors()
    with Engine(m, n, k) as b:
        b.upload_v(v)
        b.set_factors(w2, h2)
        b.mur_run(1, 0.01, 0.02, NEVER, 1e-5, 1e-5, 0, 3)
        wb, hb = b.get_factors()
    np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(ha, hb)
