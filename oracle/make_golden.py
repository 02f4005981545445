#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (raleng/nmf).

Run in the build container only (the reference lives at /root/reference and
never travels to the GPU box):

    python oracle/make_golden.py

Each fixture stores the recipe of the input (seed, shape, generator), the
initial factors actually used, the reference's outputs (w, h, i, obj_history),
snapshots after a few iterations and, for AO-ADMM, the inner iteration counts.
The inputs are regenerated from seeds by the tests (`RandomState` is a frozen
legacy stream); a checksum of V guards the regeneration.

TEST INFRASTRUCTURE ONLY -- fixtures are data, no reference source is copied.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from nmf import mur as ref_mur          # noqa: E402
from nmf import ao_admm as ref_aoadmm   # noqa: E402
from nmf import admm as ref_admm        # noqa: E402
from nmf import anls as ref_anls        # noqa: E402
from nmf import utils as ref_utils      # noqa: E402
from nmf import fcnnls as ref_fcnnls    # noqa: E402

from oracle.nmf_ref import planted_matrix, fixture_matrix as make_v  # noqa: E402  (input generators only)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


ONLY = set()       # --only name[,name...]: regenerate these fixtures only ("functions" = functions.npz)


def quiet(fn, *a, **kw):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **kw)
    return out, buf.getvalue()


def initial_factors(method, v, k, nndsvd_init, seed):
    """Re-derive the init the reference will draw after np.random.seed(seed)."""
    np.random.seed(seed)
    if nndsvd_init[0]:
        return ref_utils.nndsvd(v.copy(), k, variant=nndsvd_init[1])
    if method == "anls":
        w = np.random.rand(v.shape[0], k)
        h = np.random.rand(k, v.shape[1])
    else:
        w = np.abs(np.random.randn(v.shape[0], k))
        h = np.abs(np.random.randn(k, v.shape[1]))
    return w, h


def run_solver(method, v, k, seed, kwargs):
    fn = {"mur": ref_mur.mur, "ao_admm": ref_aoadmm.ao_admm,
          "admm": ref_admm.admm, "anls": ref_anls.anls}[method]
    inner = []
    if method == "ao_admm":
        # count `terminate` calls per sub-problem (= inner iterations run)
        calls = {"n": 0}
        orig_term = ref_aoadmm.terminate
        orig_ls, orig_kl = ref_aoadmm.admm_ls_update, ref_aoadmm.admm_kl_update

        def term(*a, **kw):
            calls["n"] += 1
            return orig_term(*a, **kw)

        def wrap(f):
            def g(*a, **kw):
                calls["n"] = 0
                out = f(*a, **kw)
                inner.append(calls["n"])
                return out
            return g
        ref_aoadmm.terminate = term
        ref_aoadmm.admm_ls_update = wrap(orig_ls)
        ref_aoadmm.admm_kl_update = wrap(orig_kl)
    try:
        np.random.seed(seed)
        with np.errstate(all="ignore"):
            res, text = quiet(fn, v, k, **kwargs)
    finally:
        if method == "ao_admm":
            ref_aoadmm.terminate = orig_term
            ref_aoadmm.admm_ls_update = orig_ls
            ref_aoadmm.admm_kl_update = orig_kl
    return res, text, inner


def solver_case(name, method, vspec, k, seed, kwargs, snaps=(1, 2, 10), extra=None):
    if ONLY and name not in ONLY:
        return
    v = make_v(vspec)
    v_in = v.copy()
    init = kwargs.get("nndsvd_init", {"mur": (False, "zero")}.get(method, (True, "zero")))
    w0, h0 = initial_factors(method, v.copy() + (abs(v.min()) if (method == "mur" and v.min() < 0) else 0), k, init, seed)
    res, text, inner = run_solver(method, v, k, seed, kwargs)
    rule = 0
    if "Algorithm converged (1)." in text:
        rule = 1
    elif "Algorithm converged (2)." in text:
        rule = 2
    data = dict(
        meta=json.dumps(dict(name=name, method=method, vspec=vspec, k=k, seed=seed,
                             kwargs=kwargs, numpy=np.__version__)),
        v_sum=np.float64(v_in.astype(np.float64).sum()),
        v_after_sum=np.float64(v.astype(np.float64).sum()),   # MUR shifts in place
        w0=w0, h0=h0, w=res.w, h=res.h, i=np.int64(res.i),
        obj_history=np.asarray(res.obj_history, dtype=np.float64),
        stop_rule=np.int64(rule), w_dtype=str(res.w.dtype),
        experiment=json.dumps(list(res.experiment._asdict().items()), default=str),
    )
    if inner:
        data["inner"] = np.asarray(inner, dtype=np.int64).reshape(-1, 2)
        data["admm_breaks"] = np.int64(text.count("ADMM break after"))
    for s in snaps:
        if s > res.i + 1:
            continue
        kw = dict(kwargs)
        kw["max_iter"] = s
        kw["min_iter"] = s + 5
        r2, _, _ = run_solver(method, make_v(vspec), k, seed, kw)
        data[f"snap{s}_w"] = r2.w
        data[f"snap{s}_h"] = r2.h
    if extra:
        data.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(f"{name:34s} i={res.i:4d} obj0={res.obj_history[0]:.6g} objN={res.obj_history[-1]:.6g} "
          f"rule={rule} inner={len(inner)//2}")


def function_vectors():
    if ONLY and "functions" not in ONLY:
        return
    _function_vectors()


def _function_vectors():
    rs = np.random.RandomState(7)
    out = {}
    # distance (utils.py:18-33) incl. zeros in v and in wh
    v = rs.rand(12, 9)
    wh = rs.rand(12, 9) + 0.1
    v[0, 0] = 0.0
    v[3, 4] = 0.0
    wh2 = wh.copy()
    wh2[5, 5] = 0.0      # v>0, wh=0 -> +inf -> replaced by 0
    wh2[0, 0] = 0.0      # 0 * log(0/0) -> nan -> 0
    with np.errstate(all="ignore"):
        out["dist_v"], out["dist_wh"], out["dist_wh2"] = v, wh, wh2
        out["dist_eu"] = ref_utils.distance(v, wh, "eu")
        out["dist_kl"] = ref_utils.distance(v, wh, "kl")
        out["dist_kl_zero"] = ref_utils.distance(v, wh2, "kl")
    # convergence_check (utils.py:4-15)
    cc_in = np.array([[1e-6, 1.0, 1e-5, 1e-5], [0.5, 0.6, 1e-5, 1e-5], [0.5, 0.5, 1e-5, 1e-5],
                      [0.5, 0.500001, 1e-5, 1e-5], [0.6, 0.5, 1e-5, 1e-5], [0.4, 0.5, 1e-3, 0.2]])
    cc_out = []
    for new, old, t1, t2 in cc_in:
        r, _ = quiet(ref_utils.convergence_check, new, old, t1, t2)
        cc_out.append(r)
    out["cc_in"], out["cc_out"] = cc_in, np.array(cc_out)
    # prox nn / l1n (ao_admm.py:113-124) and l2n (admm.py:141-156)
    aux = rs.randn(6, 10)
    dual = 0.3 * rs.randn(6, 10)
    out["prox_aux"], out["prox_dual"] = aux, dual
    out["prox_nn"] = ref_aoadmm.prox("nn", aux, dual, rho=2.5, lambda_=0.4)
    out["prox_l1n"] = ref_aoadmm.prox("l1n", aux, dual, rho=2.5, lambda_=0.4)
    out["prox_l2n"] = ref_admm.prox("l2n", aux, dual, rho=2.5, lambda_=0.4)
    try:
        ref_aoadmm.prox("l2n", aux, dual, rho=2.5, lambda_=0.4)
        out["prox_l2n_aoadmm_raises"] = np.int64(0)
    except ValueError:
        out["prox_l2n_aoadmm_raises"] = np.int64(1)
    # prox l1inf / l1inf_transpose (admm.py:158-210): rows that pass the sum test and rows that do not
    aux_s = aux * np.linspace(0.02, 1.0, aux.shape[0])[:, None]
    for tag, a_ in (("", aux), ("_mixed", aux_s)):
        out["prox_l1inf" + tag] = quiet(ref_admm.prox, "l1inf", a_, dual, rho=2.5, lambda_=0.4)[0]
        out["prox_l1inf_t" + tag] = quiet(ref_admm.prox, "l1inf_transpose", a_, dual, rho=2.5, lambda_=0.4)[0]
    out["prox_l1inf_aux_mixed"] = aux_s
    # terminate (ao_admm.py:33-43), incl. zero dual -> inf
    mat, prev = np.abs(rs.randn(4, 7)), np.abs(rs.randn(4, 7))
    aux2 = mat + 1e-3 * rs.randn(4, 7)
    with np.errstate(all="ignore"):
        out["term_mat"], out["term_prev"], out["term_aux"] = mat, prev, aux2
        out["term_dual"] = dual[:4, :7]
        out["term_a"] = ref_aoadmm.terminate(mat, prev, aux2, dual[:4, :7])
        out["term_zero_dual"] = ref_aoadmm.terminate(mat, mat + 1e-6, aux2, np.zeros((4, 7)))
        out["term_true"] = ref_aoadmm.terminate(mat, mat + 1e-7, mat + 1e-7, 10 + np.zeros((4, 7)))
    # nndsvd (utils.py:36-93)
    x = planted_matrix(40, 30, 5, seed=3, dtype=np.float64)
    out["nndsvd_x"] = x
    for var in ("zero", "mean"):
        w, h = ref_utils.nndsvd(x, 5, variant=var)
        out[f"nndsvd_{var}_w"], out[f"nndsvd_{var}_h"] = w, h
    np.random.seed(11)
    w, h = ref_utils.nndsvd(x, 5, variant="random")
    out["nndsvd_random_w"], out["nndsvd_random_h"] = w, h
    # fcnnls (fcnnls.py:55-136) against per-column NNLS inputs
    c = rs.rand(20, 5)
    a = c @ np.maximum(rs.rand(5, 8) - 0.15, 0) + 0.01 * rs.randn(20, 8)
    out["fc_c"], out["fc_a"] = c, a
    kfc, txt = quiet(ref_fcnnls.fcnnls, c, a)
    assert "Not converged" not in txt and kfc.min() >= 0
    out["fc_k"] = kfc
    # MUR single steps (mur.py:20-49)
    xv = rs.rand(15, 11)
    w, h = np.abs(rs.randn(15, 3)), np.abs(rs.randn(3, 11))
    out["step_v"], out["step_w"], out["step_h"] = xv, w, h
    for kind in ("eu", "kl"):
        for lam in (0.0, 0.3):
            wn = ref_mur.w_update(kind, xv, w, h, w @ h, lam)
            hn = ref_mur.h_update(kind, xv, wn, h, wn @ h, lam)
            tag = f"{kind}_{str(lam).replace('.', 'p')}"
            out[f"step_w_{tag}"], out[f"step_h_{tag}"] = wn, hn
    # one AO-ADMM sub-problem (ao_admm.py:46-68)
    y = rs.rand(14, 9)
    w = np.abs(rs.randn(14, 4))
    h = np.abs(rs.randn(4, 9))
    du = 0.1 * rs.randn(4, 9)
    (hh, dd), txt = quiet(ref_aoadmm.admm_ls_update, y, w, h, du, 4, "l1n", admm_iter=10, lambda_=0.2)
    out["ls_y"], out["ls_w"], out["ls_h"], out["ls_dual"] = y, w, h, du
    out["ls_h_out"], out["ls_dual_out"] = hh, dd
    # ADMM aux_update (admm.py:216-230)
    out["aux_eu"] = ref_admm.aux_update(h, du, w, y, None, 1.7, "eu")
    # cssls (fcnnls.py:14-52): the unconstrained solve, and passive sets with repeated and unique columns
    ct_c, ct_a = c.T @ c, c.T @ a
    out["cssls_full"] = ref_fcnnls.cssls(ct_c, ct_a)
    p_set = np.random.RandomState(77).rand(5, 8) > 0.35
    p_set[:, 1] = p_set[:, 0]; p_set[:, 5] = p_set[:, 0]; p_set[:, 6] = True
    out["cssls_pset"] = p_set
    out["cssls_k"] = ref_fcnnls.cssls(ct_c, ct_a, p_set=p_set)
    np.savez_compressed(os.path.join(OUT, "functions.npz"), **out)
    print("functions.npz", len(out), "arrays")


def main():
    P = dict(kind="planted", rank=8, seed=0)
    # ---- MUR (BASELINE config 1 = 512x256 f64 k=8) ----
    solver_case("mur_eu_cfg1_random", "mur", dict(P, m=512, n=256), 8, 1,
                dict(distance_type="eu", min_iter=40, max_iter=40))
    solver_case("mur_eu_cfg1_nndsvdz", "mur", dict(P, m=512, n=256), 8, 1,
                dict(distance_type="eu", min_iter=40, max_iter=40, nndsvd_init=(True, "zero")))
    solver_case("mur_eu_lambda", "mur", dict(P, m=200, n=120), 8, 2,
                dict(distance_type="eu", min_iter=30, max_iter=30, lambda_w=0.1, lambda_h=0.1))
    solver_case("mur_eu_f32v", "mur", dict(P, m=192, n=160, dtype="float32"), 8, 3,
                dict(distance_type="eu", min_iter=30, max_iter=30))
    solver_case("mur_eu_signed", "mur", dict(kind="signed", seed=4, m=96, n=80), 6, 4,
                dict(distance_type="eu", min_iter=20, max_iter=20))
    solver_case("mur_eu_converge", "mur", dict(kind="planted", rank=4, seed=5, m=120, n=90), 4, 5,
                dict(distance_type="eu", min_iter=5, max_iter=500, tol1=1e-9, tol2=2e-4))
    solver_case("mur_eu_ragged", "mur", dict(kind="uniform", seed=6, m=77, n=53), 5, 6,
                dict(distance_type="eu", min_iter=25, max_iter=25))
    solver_case("mur_kl", "mur", dict(P, m=256, n=192), 8, 7,
                dict(distance_type="kl", min_iter=30, max_iter=30))
    solver_case("mur_kl_lambda", "mur", dict(kind="uniform", seed=8, m=150, n=110), 6, 8,
                dict(distance_type="kl", min_iter=30, max_iter=30, lambda_w=0.05, lambda_h=0.05))
    solver_case("mur_kl_sparse", "mur", dict(kind="sparse", seed=9, m=100, n=70), 5, 9,
                dict(distance_type="kl", min_iter=20, max_iter=20))
    # ---- AO-ADMM ----
    A = dict(kind="planted", rank=16, seed=10, m=256, n=192)
    U = dict(kind="uniform", seed=11, m=256, n=192)
    for tag, vs in (("planted", A), ("uniform", U)):
        solver_case(f"aoadmm_eu_nn_{tag}", "ao_admm", vs, 16, 12,
                    dict(distance_type="eu", reg_w=(0, "nn"), reg_h=(0, "nn"),
                         min_iter=12, max_iter=12, admm_iter=10), snaps=(1, 2))
        solver_case(f"aoadmm_eu_l1n_{tag}", "ao_admm", vs, 16, 12,
                    dict(distance_type="eu", reg_w=(0.1, "l1n"), reg_h=(0.1, "l1n"),
                         min_iter=12, max_iter=12, admm_iter=10), snaps=(1, 2))
    solver_case("aoadmm_kl_nn", "ao_admm", dict(kind="planted", rank=8, seed=13, m=96, n=80), 8, 13,
                dict(distance_type="kl", reg_w=(0, "nn"), reg_h=(0, "nn"),
                     min_iter=8, max_iter=8, admm_iter=10), snaps=(1, 2))
    solver_case("aoadmm_eu_converge", "ao_admm", dict(kind="planted", rank=6, seed=14, m=90, n=70), 6, 14,
                dict(distance_type="eu", reg_w=(0, "nn"), reg_h=(0, "nn"),
                     min_iter=3, max_iter=200, admm_iter=10, tol1=1e-9, tol2=1e-4), snaps=(1,))
    # prox 'l1inf' / 'l1inf_transpose' in ao_admm (ao_admm.py:143-195): the operator wipes a factor out and the next Cholesky
    # factorisation raises LinAlgError -- after 0, 1 or 2 completed outer iterations, depending on the placement.  Per placement:
    # the LARGEST max_iter (<= 6) the reference completes (fixture = that run; none when the first iteration already raises)
    # and `raises_at` = the max_iter at which it raises (-1: it does not within 6).
    L = dict(kind="planted", rank=6, seed=21, m=96, n=80)
    raises = {}
    for loss in ("eu", "kl"):                            # (r5: the KL loss as well -- admm_kl_update applies the same operator, ao_admm.py:71-101)
        for tag, reg_w, reg_h in (("w_l1inf", (0.1, "l1inf"), (0, "nn")), ("w_l1inf_t", (0.1, "l1inf_transpose"), (0, "nn")),
                                  ("h_l1inf", (0, "nn"), (0.1, "l1inf")), ("h_l1inf_t", (0, "nn"), (0.1, "l1inf_transpose"))):
            name = f"aoadmm_{loss}_" + tag
            if ONLY and name not in ONLY:
                continue
            last_ok, raises_at = 0, -1
            for mx in range(1, 7):
                kw = dict(distance_type=loss, reg_w=reg_w, reg_h=reg_h, min_iter=mx, max_iter=mx, admm_iter=10)
                try:
                    with np.errstate(all="ignore"):
                        run_solver("ao_admm", make_v(L), 6, 22, kw)
                    last_ok = mx
                except np.linalg.LinAlgError:
                    raises_at = mx
                    break
            raises[(loss, tag)] = raises_at
            if last_ok:
                with np.errstate(all="ignore"):
                    solver_case(name, "ao_admm", L, 6, 22,
                                dict(distance_type=loss, reg_w=reg_w, reg_h=reg_h, min_iter=last_ok, max_iter=last_ok, admm_iter=10),
                                snaps=(), extra=dict(raises_at=np.int64(raises_at)))
            else:
                print(f"{name:34s} raises LinAlgError in its first outer iteration (no fixture)")
    if raises and not ONLY:
        assert raises[("eu", "h_l1inf")] == 1, raises        # (tests/test_gpu_aoadmm.py relies on it: no fixture for this placement)
    print("l1inf raises_at:", raises)
    # ---- ADMM ----
    D = dict(kind="planted", rank=8, seed=15, m=128, n=96)
    solver_case("admm_eu_nn", "admm", D, 8, 16,
                dict(rho=1, distance_type="eu", reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=25, max_iter=25))
    solver_case("admm_eu_l1n", "admm", D, 8, 16,
                dict(rho=2, distance_type="eu", reg_w=(0.05, "l1n"), reg_h=(0.05, "l1n"), min_iter=25, max_iter=25))
    solver_case("admm_eu_l2n", "admm", D, 8, 16,
                dict(rho=1, distance_type="eu", reg_w=(0, "nn"), reg_h=(0.5, "l2n"), min_iter=25, max_iter=25))
    # l1inf*: 6 iterations only -- as written the operator makes the iteration diverge (objective 4 -> 1e31
    # within 25 iterations on this matrix) and amplify rounding differences ~10x per iteration, so longer
    # runs pin nothing ("out of scope")
    solver_case("admm_eu_l1inf", "admm", D, 8, 16,
                dict(rho=1, distance_type="eu", reg_w=(0.05, "l1inf"), reg_h=(0.05, "l1inf"), min_iter=6, max_iter=6),
                snaps=(1, 2))
    solver_case("admm_eu_l1inf_t", "admm", D, 8, 16,
                dict(rho=2, distance_type="eu", reg_w=(0.05, "l1inf_transpose"), reg_h=(0.05, "l1inf_transpose"),
                     min_iter=6, max_iter=6), snaps=(1, 2))
    solver_case("admm_kl_nn", "admm", D, 8, 16,
                dict(rho=1, distance_type="kl", reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=25, max_iter=25))
    # ---- ANLS ----
    N = dict(kind="planted", rank=4, seed=17, m=64, n=48)
    solver_case("anls_nnls", "anls", N, 4, 18, dict(min_iter=6, max_iter=6), snaps=(1, 2))
    solver_case("anls_fcnnls", "anls", N, 4, 18, dict(min_iter=6, max_iter=6, use_fcnnls=True), snaps=(1, 2))
    # distance_type='kl': the least-squares iterates with the KL objective reported (anls.py:108,118)
    solver_case("anls_kl", "anls", N, 4, 18, dict(distance_type="kl", min_iter=6, max_iter=6), snaps=(1, 2))
    solver_case("anls_lambda_random", "anls", dict(kind="uniform", seed=19, m=50, n=40), 5, 19,
                dict(min_iter=6, max_iter=6, lambda_w=0.1, lambda_h=0.2, nndsvd_init=(False, "zero")),
                snaps=(1, 2))
    function_vectors()


if __name__ == "__main__":
    if "--only" in sys.argv:
        ONLY.update(sys.argv[sys.argv.index("--only") + 1].split(","))
    main()
