#!/usr/bin/env python3
"""Round-3 paths on awkward shapes (GPU box): pair mode, more than 128 components, the float64 referee of the stop rule."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.mur import mur, mur_pair
from nmf_amd.ao_admm import ao_admm
bad = 0


def err(res, ref, v):
    return np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))


print("-- pair mode")
for (m, n, ka, kb) in [(70, 90, 7, 64), (257, 255, 33, 1), (1000, 129, 64, 40), (129, 1000, 12, 50)]:
    v = R.planted_matrix(m, n, 8, seed=m + n, dtype=np.float32)
    par = [dict(k=ka, lambda_w=0.0, lambda_h=0.03), dict(k=kb, lambda_w=0.2, lambda_h=0.0)]
    np.random.seed(3); got = mur_pair(v.copy(), max(ka, kb), par, min_iter=15, max_iter=15)
    np.random.seed(3); ref = [R.mur(v.astype(np.float64), p["k"], distance_type="eu", lambda_w=p["lambda_w"], lambda_h=p["lambda_h"], min_iter=15, max_iter=15) for p in par]
    es = [err(g, r, v) for g, r in zip(got, ref)]
    os_ = [np.max(np.abs(np.asarray(g.obj_history) / np.asarray(r.obj_history) - 1)) for g, r in zip(got, ref)]
    ok = max(es) < 1e-4 and max(os_) < 1e-4
    bad += not ok
    print(f"{m}x{n} k=({ka},{kb}): WH {es[0]:.1e} {es[1]:.1e} obj {os_[0]:.1e} {os_[1]:.1e} {'ok' if ok else 'FAIL'}", flush=True)

print("-- more than 128 components")
for (m, n, k, dist) in [(200, 300, 129, "eu"), (333, 257, 200, "kl"), (130, 1100, 130, "eu"), (900, 140, 513, "eu")]:
    v = R.planted_matrix(m, n, 8, seed=m + k, dtype=np.float32)
    kw = dict(distance_type=dist, min_iter=12, max_iter=12, lambda_w=0.01)
    np.random.seed(4); res = mur(v.copy(), k, **kw)
    np.random.seed(4); ref = R.mur(v.astype(np.float64), k, **kw)
    e, o = err(res, ref, v), np.max(np.abs(np.asarray(res.obj_history) / np.asarray(ref.obj_history) - 1))
    ok = e < 1e-4 and o < 1e-4 and res.i == ref.i
    bad += not ok
    print(f"MUR-{dist} {m}x{n} k={k}: WH {e:.1e} obj {o:.1e} {'ok' if ok else 'FAIL'}", flush=True)
for (m, n, k) in [(300, 260, 129), (260, 520, 200)]:
    v = R.planted_matrix(m, n, 8, seed=m + k, dtype=np.float32)
    kw = dict(distance_type="eu", reg_w=(0.05, "l1n"), reg_h=(0, "nn"), min_iter=4, max_iter=4, admm_iter=7, nndsvd_init=(True, "zero"))
    res = ao_admm(v.copy(), k, **kw); ref = R.ao_admm(v.astype(np.float64), k, **kw)
    same = [tuple(t) for t in ao_admm.last_inner_counts] == [tuple(t) for t in ref.trace["inner"]]
    e = err(res, ref, v)
    ok = e < 1e-4 and same
    bad += not ok
    print(f"AO-ADMM {m}x{n} k={k}: WH {e:.1e} inner counts equal {same} {'ok' if ok else 'FAIL'}", flush=True)

print("-- the float64 referee, forced (NMFX_VERIFY_STOP=1), against the oracle's stop index")
os.environ["NMFX_VERIFY_STOP"] = "1"
for (m, n, k, tol2) in [(300, 260, 36, 5e-3), (512, 384, 40, 1e-3), (640, 1000, 64, 2e-3), (200, 130, 8, 1e-4)]:
    v = R.planted_matrix(m, n, k, seed=77, dtype=np.float32)
    kw = dict(distance_type="eu", min_iter=5, max_iter=3000, tol1=1e-9, tol2=tol2)
    np.random.seed(3); res = mur(v.copy(), k, **kw)
    np.random.seed(3); ref = R.mur(v.astype(np.float64), k, **kw)
    rf = mur.last_referee
    ok = abs(res.i - ref.i) <= 1 and len(res.obj_history) == res.i + 2
    bad += not ok
    print(f"{m}x{n} k={k} tol2={tol2}: stop {res.i} (oracle {ref.i}), guard {rf.guard:.1e}, {rf.walked} iterations refereed {'ok' if ok else 'FAIL'}", flush=True)
from nmf_amd.admm import admm
from nmf_amd.anls import anls
for name, solver, oracle, kw in [
        ("ao_admm", ao_admm, R.ao_admm, dict(reg_w=(0.05, "l1n"), reg_h=(0.02, "l1n"), min_iter=2, max_iter=400, tol1=1e-9, tol2=2e-3, admm_iter=6)),
        ("admm", admm, R.admm, dict(rho=1.0, reg_w=(0.02, "l1n"), reg_h=(0.02, "l1n"), min_iter=2, max_iter=800, tol1=1e-9, tol2=2e-3)),
        ("anls", anls, R.anls, dict(lambda_w=0.05, lambda_h=0.02, min_iter=2, max_iter=300, tol1=1e-9, tol2=1e-4))]:
    for (m, n, k) in [(300, 260, 12), (384, 520, 40)]:
        v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
        res = solver(v.copy(), k, nndsvd_init=(True, "zero"), **kw)
        ref = oracle(v.astype(np.float64), k, nndsvd_init=(True, "zero"), **kw)
        rf = solver.last_referee
        ok = abs(res.i - ref.i) <= 1 and len(res.obj_history) == res.i + 2 and err(res, ref, v) < 1e-4
        bad += not ok
        print(f"{name} {m}x{n} k={k}: stop {res.i} (oracle {ref.i}), WH {err(res, ref, v):.1e}, guard {rf.guard:.1e}, {rf.walked} refereed {'ok' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
