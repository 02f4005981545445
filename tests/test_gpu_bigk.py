"""More than 128 components (missing 1): the reference takes any `factors` (nmf/nmf.py:32-35, nmf/mur.py:52).
All four solvers compose their iterations from one tiled product kernel (kernels_generic.hip) beyond k = 128: split bf16 over operand
planes for MUR, AO-ADMM-LS and ADMM-LS (the default), exact f32 for the rest (tests/test_gpu_knobs.py runs the exact-f32 forms)."""
import numpy as np
import pytest

from gpu_common import WH_TOL, direct_objective, oracle_after, slow_oracle, slow_signature, wh_error, wh_error_blocked
from oracle import nmf_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,k", [((520, 700), 160), ((640, 520), 256), ((400, 900), 300), ((768, 512), 400)])
@pytest.mark.parametrize("distance", ["eu", "kl"])
def test_mur_beyond_128_components_vs_oracle(shape, k, distance):
    from nmf_amd.mur import mur
    m, n = shape
    v = R.planted_matrix(m, n, 24, seed=m + k, dtype=np.float32)
    kw = dict(distance_type=distance, min_iter=20, max_iter=20, lambda_w=0.02, lambda_h=0.01)
    np.random.seed(5)
    res = mur(v.copy(), k, **kw)
    np.random.seed(5)
    ref = R.mur(v.astype(np.float64), k, **kw)
    assert res.w.shape == (m, k) and res.h.shape == (k, n)
    assert res.i == ref.i and len(res.obj_history) == res.i + 2
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=4e-5)


def test_mur_kl_k160_zero_and_tiny_entries():
    """KL beyond k = 128 on data with exact zeros (utils.py:24 zeroes their 0 log 0), tiny entries and all-zero rows: the quotient
    planes and the objective of the split-bf16 product kernel's KL epilogue (x rcp(zy + 1e-9), x log(x / zy) with inf / nan -> 0)."""
    from nmf_amd.mur import mur
    m, n, k = 400, 360, 160
    rs = np.random.RandomState(k)
    v = R.planted_matrix(m, n, 12, seed=k, dtype=np.float32)
    v[rs.rand(m, n) < 0.3] = 0.0
    v[:, : n // 4] *= 1e-3
    v[: m // 8] *= 1e-2
    v[m - 3:] = 0.0
    kw = dict(distance_type="kl", min_iter=12, max_iter=12, lambda_w=0.0, lambda_h=0.01)
    np.random.seed(3)
    res = mur(v.copy(), k, **kw)
    np.random.seed(3)
    ref = R.mur(v.astype(np.float64), k, **kw)
    assert np.isfinite(res.obj_history).all() and np.isfinite(res.w).all() and np.isfinite(res.h).all()
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=4e-5)


def test_solvers_alternate_on_one_engine_beyond_128_components():
    """One handle, k = 160: MUR-eu iterations, then AO-ADMM on new factors, then MUR-eu again -- the bf16 images the MUR loop keeps
    of (W, H) must be rebuilt after another solver (or set_factors) has rewritten the factors (gxb_img_ready)."""
    from nmf_amd.engine import Engine
    from nmf_amd import _lib as L
    m, n, k = 384, 512, 160
    v = R.planted_matrix(m, n, 20, seed=9, dtype=np.float32)
    rs = np.random.RandomState(1)
    w0, h0 = 0.3 * np.abs(rs.randn(m, k)), 0.3 * np.abs(rs.randn(k, n))
    NEVER = 10 ** 12

    def mur_from(e, w, h, iters):
        e.set_factors(w, h)
        e.mur_run(L.EU, 0.0, 0.0, NEVER, 1e-9, 1e-9, 0, iters)
        e.synchronize()
        return e.get_factors(), e.objectives(0, iters)

    with Engine(m, n, k) as fresh:
        fresh.upload_v(v)
        (w_a, h_a), obj_a = mur_from(fresh, w0, h0, 6)
    with Engine(m, n, k) as e:
        e.upload_v(v)
        mur_from(e, w0 * 1.5, h0 * 0.5, 3)                        # leaves images of OTHER factors behind
        e.set_factors(w0, h0)
        e.aoadmm_run(L.EU, L.PROX['nn'], 0.0, L.PROX['nn'], 0.0, 4, NEVER, 1e-9, 1e-9, 0, 2)
        e.synchronize()
        (w_b, h_b), obj_b = mur_from(e, w0, h0, 6)
    np.testing.assert_array_equal(w_a, w_b)
    np.testing.assert_array_equal(h_a, h_b)
    np.testing.assert_array_equal(obj_a, obj_b)


def test_precision_switched_on_one_engine_beyond_128_components():
    """k = 160, one handle: split-bf16 iterations, exact-f32 iterations (set_precision), split bf16 again -- equal to the same three
    legs on fresh engines handing the factors over (the exact-f32 leg rewrites W and H without the bf16 images of the MUR loop)."""
    from nmf_amd.engine import Engine
    from nmf_amd import _lib as L
    m, n, k = 384, 512, 160
    v = R.planted_matrix(m, n, 20, seed=11, dtype=np.float32)
    rs = np.random.RandomState(2)
    w, h = 0.3 * np.abs(rs.randn(m, k)), 0.3 * np.abs(rs.randn(k, n))
    NEVER = 10 ** 12
    legs = (("bf16", 3), ("f32", 2), ("bf16", 3))
    with Engine(m, n, k) as e:
        e.upload_v(v)
        e.set_factors(w, h)
        first = 0
        for mode, count in legs:
            e.set_precision(mode)
            e.mur_run(L.EU, 0.01, 0.0, NEVER, 1e-9, 1e-9, first, count)
            first += count
        e.synchronize()
        w_a, h_a = e.get_factors()
    for mode, count in legs:
        with Engine(m, n, k) as e:
            e.upload_v(v)
            e.set_precision(mode)
            e.set_factors(w, h)
            e.mur_run(L.EU, 0.01, 0.0, NEVER, 1e-9, 1e-9, 0, count)
            e.synchronize()
            w, h = e.get_factors()
    np.testing.assert_array_equal(w_a, w)
    np.testing.assert_array_equal(h_a, h)


def test_mur_eu_k160_stop_rule_and_negative_data():
    """k = 160 with the stop rule firing (same index and rule as the oracle) on data with negative entries (lifted in place,
    nmf/mur.py:99-101)."""
    from nmf_amd.mur import mur
    v = R.planted_matrix(300, 260, 36, seed=77, dtype=np.float32) - 0.05
    kw = dict(distance_type="eu", min_iter=5, max_iter=400, tol1=1e-9, tol2=2e-2)
    a, b = v.copy(), v.astype(np.float64)
    np.random.seed(3)
    res = mur(a, 160, **kw)
    np.random.seed(3)
    ref = R.mur(b, 160, **kw)
    assert a.min() >= 0 and ref.trace["stop_rule"] == 2 and ref.i < 399
    assert abs(res.i - ref.i) <= 1 and len(res.obj_history) == res.i + 2
    # the iterate is pinned whether or not the index is the oracle's: the oracle after res.i + 1 iterations (on the lifted data)
    w_o, h_o, obj_o = oracle_after(R.mur, v.astype(np.float64) - 0.0, 160, 3, res.i + 1, **kw)
    assert wh_error(res.w, res.h, w_o, h_o, b) < WH_TOL
    np.testing.assert_allclose(res.obj_history, obj_o, rtol=4e-5)


@pytest.mark.parametrize("shape,k,regs", [((520, 700), 160, ((0.1, "l1n"), (0.05, "l1n"))), ((640, 520), 256, ((0.02, "l1n"), (0, "nn"))),
                                           ((400, 520), 300, ((0.05, "l1n"), (0.02, "l1n")))])      # k pads to 384: three factor tiles per wave of the bf16 rounds
def test_aoadmm_beyond_128_components_vs_oracle(shape, k, regs):
    """AO-ADMM (least-squares loss, nn / l1n) for k > 128: Gram systems by an f64 Gauss-Jordan inversion, the rounds of
    nmf/ao_admm.py:59-64 one by one with the stop test on the device; inner round counts equal to the oracle's."""
    from nmf_amd.ao_admm import ao_admm
    m, n = shape
    v = R.planted_matrix(m, n, 24, seed=m + k, dtype=np.float32)
    kw = dict(distance_type="eu", reg_w=regs[0], reg_h=regs[1], min_iter=6, max_iter=6, admm_iter=8, nndsvd_init=(True, "zero"))
    res = ao_admm(v.copy(), k, **kw)
    ref = R.ao_admm(v.astype(np.float64), k, **kw)
    assert res.i == ref.i and len(res.obj_history) == res.i + 2
    assert [tuple(r) for r in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]]
    assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=1e-4)


@pytest.mark.parametrize("distance,regs,rho", [
    ("eu", ((0, "nn"), (0, "l2n")), 1.0),                      # the reference's defaults (reg_h = (0, 'l2n'))
    ("eu", ((0.02, "l1n"), (0.1, "l2n")), 1.0),
    ("eu", ((0.05, "l1n"), (0.05, "l1n")), 2.0),
    ("kl", ((0, "nn"), (0.05, "l1n")), 1.0),
])
def test_admm_beyond_128_components_vs_oracle(distance, regs, rho):
    """ADMM (nmf/admm.py:292-334) for k > 128: the shifted Gram systems by an f64 Gauss-Jordan inversion, every product on the
    generic exact-f32 MFMA kernel, 'l2n' through its k x k operator, the KL auxiliaries in the epilogue of w_aux h_aux."""
    from nmf_amd.admm import admm
    m, n, k = 520, 400, 160
    v = R.planted_matrix(m, n, 24, seed=m + k, dtype=np.float32)
    kw = dict(rho=rho, distance_type=distance, reg_w=regs[0], reg_h=regs[1], min_iter=8, max_iter=8, nndsvd_init=(True, "zero"))
    res = admm(v.copy(), k, **kw)
    ref = R.admm(v.astype(np.float64), k, **kw)
    assert res.w.shape == (m, k) and res.h.shape == (k, n)
    assert res.i == ref.i and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, ref.w, ref.h, v)
    rel = np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history))
    print(f"\nADMM k={k} {distance} {regs}: WH {err:.2e}, objective max rel diff {rel:.2e}")
    assert err < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=5e-4)


def test_admm_l1inf_beyond_128_components_like_the_small_k_path():
    """'l1inf' (one vector per factor) has no limit on k; 'l1inf_transpose' (one vector per column: k entries) takes one workgroup per
    column beyond 128 entries (r4: it used to refuse) -- both against the oracle over the first two iterations, on both sides."""
    from nmf_amd.admm import admm
    m, n, k = 300, 260, 160
    v = R.planted_matrix(m, n, 12, seed=4, dtype=np.float32)
    kw = dict(rho=1.0, distance_type="eu", reg_w=(0, "nn"), reg_h=(0.1, "l1inf"), min_iter=2, max_iter=2, nndsvd_init=(True, "zero"))
    res = admm(v.copy(), k, **kw)
    ref = R.admm(v.astype(np.float64), k, **kw)
    assert res.i == ref.i
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=1e-3)
    for regs in (dict(reg_w=(0, "nn"), reg_h=(0.1, "l1inf_transpose")), dict(reg_w=(0.1, "l1inf_transpose"), reg_h=(0, "nn"))):
        kw = dict(rho=1.0, distance_type="eu", min_iter=2, max_iter=2, nndsvd_init=(True, "zero"), **regs)
        res = admm(v.copy(), k, **kw)
        with np.errstate(all="ignore"):
            ref = R.admm(v.astype(np.float64), k, **kw)
        assert res.i == ref.i
        np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=1e-3)
        assert wh_error(res.w, res.h, ref.w, ref.h, v) < WH_TOL


@pytest.mark.parametrize("regs", [((0, "nn"), (0, "nn")), ((0.05, "l1n"), (0.02, "l1n"))])
def test_aoadmm_kl_beyond_128_components_vs_oracle(regs):
    """AO-ADMM with the KL loss (nmf/ao_admm.py:71-101) for k > 128: per round W^T (v_aux + dual_v), the Gram solve, prox, and
    v_aux / dual_v in the epilogue of the product W h_aux; inner round counts equal to the oracle's."""
    from nmf_amd.ao_admm import ao_admm
    m, n, k = 520, 400, 160
    v = R.planted_matrix(m, n, 24, seed=m + k + 1, dtype=np.float32)
    kw = dict(distance_type="kl", reg_w=regs[0], reg_h=regs[1], min_iter=5, max_iter=5, admm_iter=6, nndsvd_init=(True, "zero"))
    res = ao_admm(v.copy(), k, **kw)
    ref = R.ao_admm(v.astype(np.float64), k, **kw)
    assert res.i == ref.i and len(res.obj_history) == res.i + 2
    assert [tuple(r) for r in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]]
    err = wh_error(res.w, res.h, ref.w, ref.h, v)
    rel = np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history))
    print(f"\nAO-ADMM-KL k={k} {regs}: WH {err:.2e}, objective max rel diff {rel:.2e}")
    assert err < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=2e-4)


@pytest.mark.parametrize("shape,k,distance", [((520, 400), 160, "eu"), ((300, 420), 144, "kl")])
def test_anls_beyond_128_components_vs_oracle(shape, k, distance):
    """ANLS (nmf/anls.py:18-47) for k > 128: block principal pivoting with one workgroup per right-hand side, the passive-set
    systems factorised in float64 in a global work area, against the oracle (scipy's Lawson-Hanson NNLS)."""
    from nmf_amd.anls import anls
    m, n = shape
    v = R.planted_matrix(m, n, 24, seed=m + k, dtype=np.float32)
    kw = dict(distance_type=distance, lambda_w=0.05, lambda_h=0.02, min_iter=3, max_iter=3, nndsvd_init=(True, "zero"))     # (the oracle's scipy NNLS sets the run time)
    res = anls(v.copy(), k, **kw)
    ref = slow_oracle(f"anls_{distance}_{m}x{n}_k{k}", slow_signature(v, k, kw), lambda: R.anls(v.astype(np.float64), k, **kw))
    assert res.w.shape == (m, k) and res.h.shape == (k, n)
    assert res.i == ref.i and len(res.obj_history) == res.i + 2
    err = wh_error(res.w, res.h, ref.w, ref.h, v)
    rel = np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history))
    print(f"\nANLS k={k} {distance}: WH {err:.2e}, objective max rel diff {rel:.2e}")
    assert err < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=2e-4)


def test_anls_k256_kkt_at_lambda_zero_with_a_dead_component():
    """k = 256, no regularisation, one component dead in the start (its pivot vanishes on the warm-started passive set and the
    variable is dropped, as scipy's NNLS leaves it at zero): the returned H is a KKT point of its sub-problem for the returned W,
    checked in float64 on the host."""
    from nmf_amd.engine import Engine
    m, n, k, iters = 640, 520, 256, 2
    v = R.planted_matrix(m, n, 48, seed=11, dtype=np.float32)
    rs = np.random.RandomState(2)
    w0, h0 = rs.rand(m, k), rs.rand(k, n)
    h0[5] = 0.0
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
    assert len(obj) == iters + 1 and obj[-1] < obj[0]
    assert not w[:, 5].any() and not h[5].any()
    vd = v.astype(np.float64)
    y = (w.T @ w) @ h - w.T @ vd
    scale = np.abs(w.T @ vd).max()
    assert y[h == 0].min() > -2e-4 * scale and np.abs(y[h > 0]).max() < 2e-4 * scale


def test_row_sharded_phases_run_beyond_128_components():
    """r4: the row-sharded phase entry points of AO-ADMM, ADMM and ANLS are composed from the generic kernels beyond 128 components
    (tests/test_gpu_dist.py runs them against the oracle with 1 and 2 ranks; r5: AO-ADMM's KL-loss phases too); what stays at
    k <= 128 says so: AO-ADMM's speculative (fused) W sub-problem."""
    from nmf_amd._lib import NmfxError
    from nmf_amd.engine import Engine
    v = R.planted_matrix(300, 260, 8, seed=1, dtype=np.float32)
    rs = np.random.RandomState(0)
    with Engine(300, 260, 160) as eng:
        eng.upload_v(v)
        eng.set_factors(rs.rand(300, 160), rs.rand(160, 260))
        eng.anls_phase_objective(0)
        eng.synchronize()
    with Engine(300, 260, 160) as eng:
        eng.upload_v(v)
        eng.set_factors(rs.rand(300, 160), rs.rand(160, 260))
        eng._ck(eng.lib.nmfx_aoadmm_kl_phase_h_products(eng.h, 0, 0))          # (r5: runs)
        eng.synchronize()
        with pytest.raises(NmfxError, match="more than 128 components"):
            eng.aoadmm_phase_w_fused(0, 0.0, 10)
        with pytest.raises(NmfxError, match="more than 128 components"):
            eng.aoadmm_phase_w_repair(0, 0.0, 10, 0)


def test_mur_eu_16384x8192_k256_vs_oracle():
    """The config-2 matrix with k = 256: 3 outer iterations against the f64 oracle (~ 1 s of host GEMMs per iteration at this k)."""
    from nmf_amd.mur import mur
    m, n, k, iters = 16384, 8192, 256, 3
    v = R.planted_matrix(m, n, 64, seed=0, dtype=np.float32)
    kw = dict(distance_type="eu", min_iter=iters, max_iter=iters)
    np.random.seed(0)
    res = mur(v, k, **kw)
    np.random.seed(0)
    ref = R.mur(v, k, **kw)
    assert res.i == ref.i == iters - 1
    err = wh_error_blocked(res.w, res.h, ref.w, ref.h, v)
    print(f"\nPARITY {m}x{n} k={k}: WH {err:.2e}, objective max rel diff "
          f"{np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history)):.2e}")
    assert err < WH_TOL, err
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=2e-5)
    direct = direct_objective(v, res.w, res.h, "eu")
    assert abs(direct - res.obj_history[-1]) <= 1e-5 * direct


def test_mur_kl_16384x8192_k256_vs_oracle():
    """The config-2 matrix with k = 256 and the KL divergence: 3 outer iterations against the f64 oracle.  At this size the
    quotient kernel is persistent (16 tiles of 256 x 128 per block, its epilogue inside the next tile, waits that count loads and stores
    alike) and the long contractions run on 256 x 256 tiles with the quotient planes as their operand -- none of which a small shape
    reaches (r4)."""
    from nmf_amd.mur import mur
    m, n, k, iters = 16384, 8192, 256, 3
    v = R.planted_matrix(m, n, 64, seed=0, dtype=np.float32)
    kw = dict(distance_type="kl", min_iter=iters, max_iter=iters)
    np.random.seed(0)
    res = mur(v, k, **kw)
    np.random.seed(0)
    with np.errstate(all="ignore"):
        ref = R.mur(v, k, **kw)
    assert res.i == ref.i == iters - 1
    err = wh_error_blocked(res.w, res.h, ref.w, ref.h, v)
    print(f"\nPARITY KL {m}x{n} k={k}: WH {err:.2e}, objective max rel diff "
          f"{np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history)):.2e}")
    assert err < WH_TOL, err
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=4e-5)


def test_aoadmm_4096x2048_k256_inner_rounds_on_the_bf16_matrix_cores():
    """AO-ADMM at k = 256 on a shape with whole 256 x 256 output tiles and many 64-row / 64-column blocks per sub-problem: the
    any-rank rounds' product on three bf16 images (r4) against the oracle, inner counts included."""
    from nmf_amd.ao_admm import ao_admm
    m, n, k = 4096, 2048, 256
    v = R.planted_matrix(m, n, 48, seed=5, dtype=np.float32)
    kw = dict(distance_type="eu", reg_w=(0.05, "l1n"), reg_h=(0.02, "l1n"), min_iter=4, max_iter=4, admm_iter=10, nndsvd_init=(True, "zero"))
    res = ao_admm(v.copy(), k, **kw)
    ref = R.ao_admm(v.astype(np.float64), k, **kw)
    assert res.i == ref.i
    assert [tuple(r) for r in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]]
    err = wh_error(res.w, res.h, ref.w, ref.h, v)
    print(f"\nPARITY AO-ADMM {m}x{n} k={k}: WH {err:.2e}, inner {ao_admm.last_inner_counts}")
    assert err < WH_TOL
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=1e-4)
