#!/usr/bin/env python3
"""CPU model of ADMM (oracle/nmf_ref.py: admm, Euclidean) with single pieces of state / arithmetic rounded to f32: which of
them sets the distance to the f64 oracle.  No GPU.   python tools/lab/admm_f32_state.py [m n k rho iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nmf_ref as R  # noqa: E402
from tools.lab.ao_f32_state import f32, split_prod  # noqa: E402

S4 = [(0, 0), (0, 1), (1, 0), (1, 1)]


def aux_step(mat, dual, other, data, rho, rnd):
    k = other.shape[1]
    if "gb32" in rnd:
        g = (other.T.astype(np.float32) @ other.astype(np.float32)).astype(np.float64)
        b = (other.T.astype(np.float32) @ data.astype(np.float32)).astype(np.float64)
    elif "gb4" in rnd:
        g = split_prod(other.T, other, 2, 2, S4)
        b = split_prod(other.T, data, 2, 2, S4)
    else:
        g, b = other.T @ other, other.T @ data
    if "gbr" in rnd:
        g, b = f32(g), f32(b)
    minv = np.linalg.inv(g + rho * np.eye(k))
    if "minv" in rnd:
        minv = f32(minv)
    rhs = b + rho * (mat + dual)
    if "rhs" in rnd:
        rhs = f32(rhs)
    if "mm32" in rnd:
        return (minv.astype(np.float32) @ rhs.astype(np.float32)).astype(np.float64)
    out = minv @ rhs
    return f32(out) if "st" in rnd else out


def run(v, k, rho, reg_w, reg_h, iters, rnd):
    w, h = R.start_factors(v, k, (True, "zero"))
    w_aux, h_aux = w.copy(), h.copy()
    dw, dh = np.zeros_like(w), np.zeros_like(h)
    hist = [R.objective(v, w @ h, "eu")]
    r = f32 if "st" in rnd else (lambda a: a)
    for _ in range(iters):
        h_aux = aux_step(h, dh, w_aux, v, rho, rnd)
        w_aux = aux_step(w.T, dw.T, h_aux.T, v.T, rho, rnd).T
        h = r(R.prox(reg_h[1], h_aux, dh, rho=rho, lam=reg_h[0]))
        w = r(R.prox(reg_w[1], w_aux.T, dw.T, rho=rho, lam=reg_w[0]).T)
        dh = r(dh + h - h_aux)
        dw = r(dw + w - w_aux)
        hist.append(R.objective(v, w @ h, "eu"))
    return w, h, np.array(hist)


if __name__ == "__main__":
    m, n, k, rho, it = 384, 640, 100, 1.0, 12
    if len(sys.argv) > 1:
        m, n, k = (int(a) for a in sys.argv[1:4]); rho = float(sys.argv[4]); it = int(sys.argv[5])
    v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32).astype(np.float64)
    nv = np.linalg.norm(v)
    regs = ((0.02, "l1n"), (0.1, "l2n"))
    w0, h0, o0 = run(v, k, rho, *regs, it, ())
    g = w0.T @ w0
    print("cond(G + rho I) at the end:", np.linalg.cond(g + rho * np.eye(k)), " residual/|V|:", np.sqrt(2 * o0[-1]) / nv)
    for rnd in [("st",), ("minv",), ("gbr",), ("rhs",), ("mm32",), ("gb32",), ("gb4",), ("st", "minv", "gbr", "rhs"), ("st", "minv", "gb32", "rhs", "mm32"),
                ("st", "gb32", "rhs"), ("st", "gb4", "gbr", "rhs", "minv", "mm32")]:
        w, h, o = run(v, k, rho, *regs, it, rnd)
        print(f"{'+'.join(rnd):28s} WH err {np.linalg.norm(w @ h - w0 @ h0) / nv:.3e}  objective max rel {np.max(np.abs(o - o0) / o0):.3e}", flush=True)
