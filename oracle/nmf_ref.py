"""numpy restatement of the reference NMF hot path (TEST INFRASTRUCTURE ONLY).

This module restates, in plain numpy/scipy and in the reference's literal
evaluation order, the arithmetic of the four solvers of raleng/nmf so that the
HIP engine can be checked on a box where the reference itself is absent.  Every
function cites the reference lines it follows (paths relative to the reference
checkout, e.g. ``nmf/mur.py:29``).  It is pinned against outputs of the real
reference by ``tests/test_oracle_golden.py`` (fixtures written by
``oracle/make_golden.py``).

It is deliberately slow and simple: six m*n*k GEMMs per MUR iteration and a
materialised ``w @ h``, exactly what the reference spends its time on; that is
also what ``bench.py`` times as the ``cpu_baseline`` ("kind": "port").

Not restated (unreachable or broken in the reference):
``mur.normalize``, ``admm.admm_ls_update``/``admm_kl_update`` (dead code),
``bpp.py``, the two legacy drivers, and the ``l1inf*`` prox operators.
"""
from __future__ import annotations

import math
from collections import namedtuple

import numpy as np
import scipy.linalg as sla
import scipy.optimize as sopt
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = 1e-9  # nmf/mur.py:25,29,41,45

Outcome = namedtuple("Outcome", "w h i obj_history trace")


# --------------------------------------------------------------------------
# L0 utilities
# --------------------------------------------------------------------------
def objective(v, wh, kind="eu"):
    """nmf/utils.py:18-33 (`distance`).  KL: inf and nan terms of
    v*log(v/wh) are replaced by 0 before `- v + wh` is added."""
    if kind == "eu":
        return 0.5 * np.sum((v - wh) ** 2)
    if kind == "kl":
        with np.errstate(divide="ignore", invalid="ignore"):
            t = v * np.log(v / wh)
        t = np.where(t == np.inf, 0, t)
        t = np.where(np.isnan(t), 0, t)
        return np.sum(t - v + wh)
    raise KeyError('Distance type unknown: use "kl" or "eu"')


def stop_rule(new, old, tol1, tol2):
    """nmf/utils.py:4-15 (`convergence_check`).  Returns 0 (continue),
    1 (new < tol1) or 2 (new >= old - tol2); rule 1 is tested first."""
    if new < tol1:
        return 1
    if new >= old - tol2:
        return 2
    return 0


def svd_init(v, rank, variant="zero", rng=np.random):
    """nmf/utils.py:36-93 (`nndsvd`, Boutsidis & Gallopoulos)."""
    left, sing, right_t = np.linalg.svd(v, full_matrices=False)
    right = right_t.T
    m, n = v.shape
    w = np.zeros((m, rank))
    h = np.zeros((rank, n))
    w[:, 0] = np.sqrt(sing[0]) * np.abs(left[:, 0])
    h[0, :] = np.sqrt(sing[0]) * np.abs(right[:, 0].T)
    for c in range(1, rank):
        a, b = left[:, c], right[:, c]
        a_p, a_n = (a >= 0) * a, (a < 0) * -a
        b_p, b_n = (b >= 0) * b, (b < 0) * -b
        na_p, na_n = np.linalg.norm(a_p, 2), np.linalg.norm(a_n, 2)
        nb_p, nb_n = np.linalg.norm(b_p, 2), np.linalg.norm(b_n, 2)
        mass_p, mass_n = na_p * nb_p, na_n * nb_n
        if mass_p >= mass_n:
            w[:, c] = np.sqrt(sing[c] * mass_p) / na_p * a_p
            h[c, :] = np.sqrt(sing[c] * mass_p) / nb_p * b_p.T
        else:
            w[:, c] = np.sqrt(sing[c] * mass_n) / na_n * a_n
            h[c, :] = np.sqrt(sing[c] * mass_n) / nb_n * b_n.T
    if variant == "mean":
        w = np.where(w == 0, np.mean(v), w)
        h = np.where(h == 0, np.mean(v), h)
    elif variant == "random":
        fill = np.mean(v) * rng.random_sample(w.shape) / 100
        w = np.where(w == 0, fill, w)
        fill = np.mean(v) * rng.random_sample(h.shape) / 100
        h = np.where(h == 0, fill, h)
    return w, h


def start_factors(v, k, nndsvd_init, rng=np.random, uniform=False):
    """Initial W then H, consuming the RNG in the reference's order:
    nmf/mur.py:105-109, nmf/admm.py:20-24, nmf/ao_admm.py:19-23 (|randn|),
    nmf/anls.py:101-105 (rand)."""
    if nndsvd_init[0]:
        return svd_init(v, k, variant=nndsvd_init[1], rng=rng)
    m, n = v.shape
    if uniform:
        w = rng.rand(m, k)
        h = rng.rand(k, n)
    else:
        w = np.abs(rng.randn(m, k))
        h = np.abs(rng.randn(k, n))
    return w, h


# --------------------------------------------------------------------------
# MUR  (nmf/mur.py)
# --------------------------------------------------------------------------
def mur_w_step(kind, v, w, h, wh, lam=0.0):
    """nmf/mur.py:20-33."""
    if kind == "eu":
        return w * (v @ h.T) / (wh @ h.T + lam * w + EPS)
    if kind == "kl":
        num = w * ((v / (wh + EPS)) @ h.T)
        den = np.ones_like(v) @ h.T
        return 2 * num / (den + np.sqrt(den ** 2 + 4 * lam * num))
    raise KeyError("Unknown distance type.")


def mur_h_step(kind, v, w, h, wh, lam=0.0):
    """nmf/mur.py:36-49 (called with the NEW w and wh = w_new @ h)."""
    if kind == "eu":
        return h * (w.T @ v) / (w.T @ wh + lam * h + EPS)
    if kind == "kl":
        num = h * (w.T @ (v / (wh + EPS)))
        den = 0 * np.ones(h.shape) + w.T @ np.ones_like(v)
        return 2 * num / (den + np.sqrt(den ** 2 + 4 * lam * num))
    raise KeyError("Unknown distance type.")


def mur(v, k, *, distance_type="kl", min_iter=100, max_iter=100000, tol1=1e-5,
        tol2=1e-5, lambda_w=0.0, lambda_h=0.0, nndsvd_init=(False, "zero"),
        w0=None, h0=None, rng=np.random, snapshots=()):
    """nmf/mur.py:52-146.  `w0`/`h0` override the init (test hook);
    `snapshots` = iteration counts after which (w, h) copies are kept in
    `trace['snap']`.  The negative-data shift mutates `v` in place like the
    reference (mur.py:99-101)."""
    if np.min(v) < 0:
        v += abs(np.min(v))
    if w0 is None:
        w, h = start_factors(v, k, nndsvd_init, rng)
    else:
        w, h = w0.copy(), h0.copy()
    wh = w @ h
    hist = [objective(v, wh, distance_type)]
    trace = {"snap": {}, "stop_rule": 0}
    i = -1
    for i in range(max_iter):
        w = mur_w_step(distance_type, v, w, h, wh, lambda_w)
        h = mur_h_step(distance_type, v, w, h, w @ h, lambda_h)
        wh = w @ h
        hist.append(objective(v, wh, distance_type))
        if (i + 1) in snapshots:
            trace["snap"][i + 1] = (w.copy(), h.copy())
        if i > min_iter:
            rule = stop_rule(hist[-1], hist[-2], tol1, tol2)
            if rule:
                trace["stop_rule"] = rule
                break
    return Outcome(w, h, i, hist, trace)


# --------------------------------------------------------------------------
# prox operators and the inner stop test (nmf/ao_admm.py, nmf/admm.py)
# --------------------------------------------------------------------------
def prox(kind, aux, dual, *, rho=None, lam=None, ragged_raises=False):
    """nmf/ao_admm.py:104-141 == nmf/admm.py:117-156 for nn / l1n / l2n.

    `ragged_raises=True` models ao_admm.py:128, whose dtype-less ragged
    `np.array` raises ValueError on numpy >= 1.24."""
    if kind == "nn":
        d = aux - dual
        return np.where(d < 0, 0, d)
    if kind == "l1n":
        d = aux - dual
        d = d - lam / rho
        return np.where(d < 0, 0, d)
    if kind == "l2n":
        if ragged_raises:
            raise ValueError("setting an array element with a sequence "
                             "(nmf/ao_admm.py:128 on numpy >= 1.24)")
        n = aux.shape[0]
        # second-difference operator T = tridiag(-1, 2, -1), admm.py:143-145
        t = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
        a = 1 / rho * (lam * t.T @ t + rho * sp.eye(n))
        out = spla.spsolve(a, aux - dual)
        return np.where(out < 0, 0, out)
    if kind in ("l1inf", "l1inf_transpose"):
        return prox_l1inf(aux, dual, rho, lam, by_columns=(kind == "l1inf_transpose"))
    raise TypeError("Unknown prox_type.")


def prox_l1inf(aux, dual, rho, lam, by_columns=False, upper_bound=1):
    """nmf/admm.py:158-183 ('l1inf', one vector per row) and :185-210
    ('l1inf_transpose', one vector per column), restated as written -- including
    the `aux + dual` sign of the shifted vector (the other prox types use
    `aux - dual`), the sorted vector taken from `aux - dual`, the count that is one
    less than the first failing index, and, by columns, the sorted vector built
    with column 1 of the dual for every column (admm.py:196) and the clamp of
    theta at zero (admm.py:206).  Only ADMM reaches these: in ao_admm the same
    text divides by a zero count in its first call and the next Cholesky raises."""
    a = aux.T if by_columns else aux
    u = dual.T if by_columns else dual
    out = np.zeros_like(a)
    shift = lam / rho
    pos = a + u - shift
    pos = np.where(pos < 0, 0, pos)
    length = a.shape[1]
    for i in range(a.shape[0]):
        if np.sum(pos[i]) <= upper_bound:
            out[i] = pos[i]
            continue
        val = -np.sort(-(a[i] - (u[1] if by_columns else u[i])))
        run = np.cumsum(val)
        j = np.arange(1, length + 1)
        test = rho * val + lam - rho / j * (run + shift - upper_bound)
        bad = np.nonzero(test < 0)[0]
        if bad.size:
            count = int(bad[0])                 # first failing 1-based index, minus one
            total = run[count]                  # sum of the first count + 1 entries
        else:
            count = length + 1
            total = run[-1]
        theta = rho / count * (total + shift - upper_bound)
        if by_columns and not theta > 0:
            theta = 0
        z = a[i] + u[i] - shift - theta / rho
        out[i] = np.where(z < 0, 0, z)
    return out.T if by_columns else out


def inner_residuals(mat, mat_prev, aux, dual):
    """The two ratios of nmf/ao_admm.py:33-43 (`terminate`); x/0 -> inf or nan
    exactly as numpy gives, and both compare False against the tolerance."""
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.linalg.norm(mat - aux) / np.linalg.norm(mat)
        s = np.linalg.norm(mat - mat_prev) / np.linalg.norm(dual)
    return r, s


def inner_stop(mat, mat_prev, aux, dual, tol=1e-2):
    r, s = inner_residuals(mat, mat_prev, aux, dual)
    return bool(r < tol and s < tol)


# --------------------------------------------------------------------------
# AO-ADMM  (nmf/ao_admm.py)
# --------------------------------------------------------------------------
def aoadmm_ls_block(y, w, h, dual, k, prox_kind="nn", *, admm_iter=10, lam=0,
                    ragged_raises=True):
    """nmf/ao_admm.py:46-68.  Returns (h, dual, inner_iterations_run)."""
    g = w.T @ w
    rho = np.trace(g) / k
    chol = sla.cholesky(g + rho * np.eye(g.shape[0]), lower=True)
    wty = w.T @ y
    ran = 0
    for j in range(admm_iter):
        aux = sla.cho_solve((chol, True), wty + rho * (h + dual))
        prev = h.copy()
        h = prox(prox_kind, aux, dual, rho=rho, lam=lam, ragged_raises=ragged_raises)
        dual = dual + h - aux
        ran = j + 1
        if inner_stop(h, prev, aux, dual):
            break
    return h, dual, ran


def aoadmm_kl_block(v, v_aux, dual_v, w, h, dual_h, k, prox_kind="nn", *,
                    admm_iter=10, lam=0, ragged_raises=True):
    """nmf/ao_admm.py:71-101."""
    g = w.T @ w
    rho = np.trace(g) / k
    chol = sla.cholesky(g + rho * np.eye(g.shape[0]), lower=True)
    ran = 0
    for j in range(admm_iter):
        aux = sla.cho_solve((chol, True), w.T @ (v_aux + dual_v) + rho * (h + dual_h))
        prev = h.copy()
        h = prox(prox_kind, aux, dual_h, rho=rho, lam=lam, ragged_raises=ragged_raises)
        v_bar = w @ aux - dual_v
        v_aux = 1 / 2 * ((v_bar - 1) + np.sqrt((v_bar - 1) ** 2 + 4 * v))
        dual_h = dual_h + h - aux
        dual_v = dual_v + v_aux - w @ aux
        ran = j + 1
        if inner_stop(h, prev, aux, dual_h):
            break
    return h, dual_h, v_aux, dual_v, ran


def ao_admm(v, k, *, distance_type="eu", reg_w=(0, "nn"), reg_h=(0, "l2n"),
            min_iter=10, max_iter=100000, admm_iter=10, tol1=1e-3, tol2=1e-3,
            nndsvd_init=(True, "zero"), w0=None, h0=None, rng=np.random,
            snapshots=()):
    """nmf/ao_admm.py:201-311.  `trace['inner']` lists, per outer iteration,
    the inner iteration counts (h-block, w-block)."""
    if distance_type not in ("eu", "kl"):
        # reference raises inside the loop (ao_admm.py:288) after the initial
        # objective, which itself raises KeyError first (utils.py:31)
        raise KeyError('Distance type unknown: use "kl" or "eu"')
    if w0 is None:
        w, h = start_factors(v, k, nndsvd_init, rng)
    else:
        w, h = w0.copy(), h0.copy()
    dual_w = np.zeros_like(w)
    dual_h = np.zeros_like(h)
    v_aux = np.zeros_like(v)   # ao_admm.py:28-30: v_aux and dual_v start as
    dual_v = v_aux             # the same zero array; only ever rebound
    hist = [objective(v, w @ h, distance_type)]
    trace = {"snap": {}, "stop_rule": 0, "inner": []}
    i = -1
    for i in range(max_iter):
        if distance_type == "eu":
            h, dual_h, nh = aoadmm_ls_block(v, w, h, dual_h, k, reg_h[1],
                                            admm_iter=admm_iter, lam=reg_h[0])
            wt, dwt, nw = aoadmm_ls_block(v.T, h.T, w.T, dual_w.T, k, reg_w[1],
                                          admm_iter=admm_iter, lam=reg_w[0])
            w, dual_w = wt.T, dwt.T
        else:
            h, dual_h, v_aux, dual_v, nh = aoadmm_kl_block(
                v, v_aux, dual_v, w, h, dual_h, k, reg_h[1],
                admm_iter=admm_iter, lam=reg_h[0])
            wt, dwt, vat, dvt, nw = aoadmm_kl_block(
                v.T, v_aux.T, dual_v.T, h.T, w.T, dual_w.T, k, reg_w[1],
                admm_iter=admm_iter, lam=reg_w[0])
            w, dual_w, v_aux, dual_v = wt.T, dwt.T, vat.T, dvt.T
        trace["inner"].append((nh, nw))
        hist.append(objective(v, w @ h, distance_type))
        if (i + 1) in snapshots:
            trace["snap"][i + 1] = (w.copy(), h.copy())
        if i > min_iter:
            rule = stop_rule(hist[-1], hist[-2], tol1, tol2)
            if rule:
                trace["stop_rule"] = rule
                break
    return Outcome(w, h, i, hist, trace)


# --------------------------------------------------------------------------
# ADMM  (nmf/admm.py, live part only)
# --------------------------------------------------------------------------
def admm_aux_step(mat, dual, other_aux, data_aux, data_dual, rho, kind):
    """nmf/admm.py:216-230 (`aux_update`): LU solve of the shifted Gram."""
    a = other_aux.T @ other_aux + rho * np.eye(other_aux.shape[1])
    if kind == "eu":
        b = other_aux.T @ data_aux + rho * (mat + dual)
    elif kind == "kl":
        b = other_aux.T @ (data_aux + data_dual) + rho * (mat + dual)
    else:
        raise TypeError("Unknown loss type.")
    return np.linalg.solve(a, b)


def admm(v, k, *, rho=1, distance_type="eu", reg_w=(0, "nn"), reg_h=(0, "l2n"),
         min_iter=10, max_iter=100000, tol1=1e-3, tol2=1e-3,
         nndsvd_init=(True, "zero"), w0=None, h0=None, rng=np.random,
         snapshots=()):
    """nmf/admm.py:233-345."""
    if distance_type not in ("eu", "kl"):
        raise KeyError('Distance type unknown: use "kl" or "eu"')
    if w0 is None:
        w, h = start_factors(v, k, nndsvd_init, rng)
    else:
        w, h = w0.copy(), h0.copy()
    w_aux, h_aux = w.copy(), h.copy()
    dual_w, dual_h = np.zeros_like(w), np.zeros_like(h)
    v_aux = np.zeros_like(v)
    dual_v = v_aux
    hist = [objective(v, w @ h, distance_type)]
    trace = {"snap": {}, "stop_rule": 0}
    i = -1
    for i in range(max_iter):
        if distance_type == "eu":
            h_aux = admm_aux_step(h, dual_h, w_aux, v, None, rho, "eu")
            w_aux = admm_aux_step(w.T, dual_w.T, h_aux.T, v.T, None, rho, "eu").T
            h = prox(reg_h[1], h_aux, dual_h, rho=rho, lam=reg_h[0])
            w = prox(reg_w[1], w_aux.T, dual_w.T, rho=rho, lam=reg_w[0]).T
        else:
            h_aux = admm_aux_step(h, dual_h, w_aux, v_aux, dual_v, rho, "kl")
            w_aux = admm_aux_step(w.T, dual_w.T, h_aux.T, v_aux.T, dual_v.T, rho, "kl").T
            h = prox(reg_h[1], h_aux, dual_h, rho=rho, lam=reg_h[0])
            w = prox(reg_w[1], w_aux.T, dual_w.T, rho=rho, lam=reg_w[0]).T
            v_bar = w_aux @ h_aux - dual_v
            v_aux = 1 / 2 * ((v_bar - 1) + np.sqrt((v_bar - 1) ** 2 + 4 * v))
            dual_v = dual_v + v_aux - w_aux @ h_aux
        dual_h = dual_h + h - h_aux
        dual_w = dual_w + w - w_aux
        hist.append(objective(v, w @ h, distance_type))
        if (i + 1) in snapshots:
            trace["snap"][i + 1] = (w.copy(), h.copy())
        if i > min_iter:
            rule = stop_rule(hist[-1], hist[-2], tol1, tol2)
            if rule:
                trace["stop_rule"] = rule
                break
    return Outcome(w, h, i, hist, trace)


# --------------------------------------------------------------------------
# ANLS  (nmf/anls.py; NNLS itself is scipy.optimize.nnls as in anls.py:28-29)
# --------------------------------------------------------------------------
def nnls_columns(a, b):
    """min ||a x - b_j||, x >= 0 for every column j (nmf/anls.py:27-29).
    The reference's FCNNLS path (nmf/fcnnls.py:55-136) solves the same
    strictly convex problems and agrees to rounding (pinned by a fixture)."""
    out = np.zeros((a.shape[1], b.shape[1]))
    for j in range(b.shape[1]):
        out[:, j], _ = sopt.nnls(a, b[:, j])
    return out


def passive_set_solve(gram, rhs, passive=None):
    """nmf/fcnnls.py:14-52 (`cssls`): solve gram @ k = rhs for the variables of the passive set of every column (all
    variables when `passive` is None or all True), zeros elsewhere.  The reference groups columns with equal passive
    sets and solves each group with one np.linalg.solve; column by column gives the same numbers up to the BLAS's
    grouping of right-hand sides (pinned by tests/golden/functions.npz: cssls_*)."""
    out = np.zeros_like(rhs, dtype=np.float64)
    if passive is None or np.all(passive):
        return np.linalg.solve(gram, rhs)
    for j in range(rhs.shape[1]):
        idx = np.nonzero(passive[:, j])[0]
        if idx.size:
            out[idx, j] = np.linalg.solve(gram[np.ix_(idx, idx)], rhs[idx, j])
    return out


def anls_w_step(v, h, lam):
    """nmf/anls.py:18-31: rows of W from the stacked system [h.T; sqrt(2 lam) I]."""
    a = np.concatenate((h.T, math.sqrt(2 * lam) * np.eye(h.shape[0])))
    b = np.concatenate((v.T, np.zeros((h.shape[0], v.shape[0]))))
    return nnls_columns(a, b).T


def anls_h_step(v, w, lam):
    """nmf/anls.py:34-47."""
    a = np.concatenate((w, math.sqrt(2 * lam) * np.eye(w.shape[1])))
    b = np.concatenate((v, np.zeros((w.shape[1], v.shape[1]))))
    return nnls_columns(a, b)


def anls(v, k, *, distance_type="eu", use_fcnnls=False, lambda_w=0, lambda_h=0,
         min_iter=10, max_iter=1000, tol1=1e-3, tol2=1e-3,
         nndsvd_init=(True, "zero"), w0=None, h0=None, rng=np.random,
         snapshots=()):
    """nmf/anls.py:50-135 (`use_fcnnls` only selects the reference's solver;
    the minimiser is the same)."""
    if w0 is None:
        w, h = start_factors(v, k, nndsvd_init, rng, uniform=True)
    else:
        w, h = w0.copy(), h0.copy()
    hist = [objective(v, w @ h, distance_type)]
    trace = {"snap": {}, "stop_rule": 0}
    i = -1
    for i in range(max_iter):
        w = anls_w_step(v, h, lambda_w)
        h = anls_h_step(v, w, lambda_h)
        hist.append(objective(v, w @ h, distance_type))
        if (i + 1) in snapshots:
            trace["snap"][i + 1] = (w.copy(), h.copy())
        if i > min_iter:
            rule = stop_rule(hist[-1], hist[-2], tol1, tol2)
            if rule:
                trace["stop_rule"] = rule
                break
    return Outcome(w, h, i, hist, trace)


SOLVERS = {"mur": mur, "ao_admm": ao_admm, "admm": admm, "anls": anls}


# --------------------------------------------------------------------------
# synthetic inputs shared by tests and bench
# --------------------------------------------------------------------------
def planted_matrix(m, n, k, seed=0, dtype=np.float32, noise=0.01, rows=None):
    """V = (U1 @ U2)/k + noise*U3 with U* ~ U(0,1) from RandomState(seed).
    `rows=(r0, r1)` returns only that row block (same values as the full
    matrix) -- the stream is drawn as U1 (m*k), U2 (k*n) then U3 row by row."""
    rs = np.random.RandomState(seed)
    left = rs.rand(m, k)
    right = rs.rand(k, n)
    r0, r1 = (0, m) if rows is None else rows
    out = np.empty((r1 - r0, n), dtype=dtype)
    step = 2048
    for a in range(0, m, step):
        b = min(m, a + step)
        nz = rs.rand(b - a, n)
        lo, hi = max(a, r0), min(b, r1)
        if lo < hi:
            blk = (left[lo:hi] @ right) / k + noise * nz[lo - a:hi - a]
            out[lo - r0:hi - r0] = blk.astype(dtype)
    return out


def fixture_matrix(spec):
    """Input matrix of a golden fixture from its recipe (`meta['vspec']`)."""
    kind = spec["kind"]
    m, n = spec["m"], spec["n"]
    dtype = np.dtype(spec.get("dtype", "float64"))
    if kind == "planted":
        return planted_matrix(m, n, spec["rank"], seed=spec["seed"], dtype=dtype)
    rs = np.random.RandomState(spec["seed"])
    if kind == "uniform":
        return rs.rand(m, n).astype(dtype)
    if kind == "signed":    # has negatives -> exercises MUR's in-place shift
        return (rs.rand(m, n) - 0.05).astype(dtype)
    if kind == "sparse":    # exact zeros -> KL inf/nan branches
        v = rs.rand(m, n)
        v[rs.rand(m, n) < 0.2] = 0.0
        return v.astype(dtype)
    raise ValueError(kind)
