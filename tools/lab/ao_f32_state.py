#!/usr/bin/env python3
"""CPU model of where AO-ADMM loses digits when its state is kept in f32.

The oracle's loop (oracle/nmf_ref.py: ao_admm / aoadmm_ls_block) with selectable roundings to f32 of
single pieces of state; prints the WH error against the all-f64 run for each subset.  No GPU.
"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nmf_ref as R  # noqa: E402

f32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731


def split16(a):
    """hi + lo bf16 images of an f32 value (16 significant bits, round to nearest even each)."""
    a32 = a.astype(np.float32)
    def bf(x):
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    hi = bf(a32)
    lo = bf((a32 - hi).astype(np.float32))
    return hi.astype(np.float64) + lo.astype(np.float64)


def bf(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def parts(a, n):
    """n bf16 images of an f32 array: a ~ p0 + p1 + ..."""
    rest = a.astype(np.float32)
    out = []
    for _ in range(n):
        p_ = bf(rest)
        out.append(p_.astype(np.float64))
        rest = (rest - p_).astype(np.float32)
    return out


def split_prod(a, b, na, nb, terms):
    pa, pb = parts(a, na), parts(b, nb)
    acc = 0
    for (i, j) in terms:
        acc = acc + pa[i] @ pb[j]
    return acc


VARIANTS = {
    "s22": (2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
    "s32": (3, 2, [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1)]),
    "s23": (2, 3, [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)]),
    "s33": (3, 3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]),
}


def block(y, w, h, dual, k, kind, lam, admm_iter, rnd):
    g = w.T @ w
    if "gb32" in rnd:
        g = (w.T.astype(np.float32) @ w.astype(np.float32)).astype(np.float64)
    if "g" in rnd:
        g = f32(g)
    rho = np.trace(g) / k
    minv = np.linalg.inv(g + rho * np.eye(k))
    if "minv" in rnd:
        minv = f32(minv)
    wty = w.T @ y
    if "gb32" in rnd:
        wty = (w.T.astype(np.float32) @ y.astype(np.float32)).astype(np.float64)
    if "b3" in rnd:
        wty = split_prod(w.T, y, 2, 2, [(0, 0), (0, 1), (1, 0)])
        g = split_prod(w.T, w, 2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)])
        rho = np.trace(g) / k
        minv = f32(np.linalg.inv(g + rho * np.eye(k)))
    if "b4" in rnd:
        wty = split_prod(w.T, y, 2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)])
        g = split_prod(w.T, w, 2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)])
        rho = np.trace(g) / k
        minv = f32(np.linalg.inv(g + rho * np.eye(k)))
    if "b" in rnd:
        wty = f32(wty)
    ran = 0
    for j in range(admm_iter):
        rhs = wty + rho * (h + dual)
        if "rhs" in rnd:
            rhs = f32(rhs)
        if "mm32" in rnd:
            aux = (minv.astype(np.float32) @ rhs.astype(np.float32)).astype(np.float64)
        elif any(t in VARIANTS for t in rnd):
            na, nb, terms = VARIANTS[[t for t in rnd if t in VARIANTS][0]]
            aux = split_prod(minv, rhs, na, nb, terms)
        elif "split" in rnd:
            aux = split16(minv) @ split16(rhs)
        else:
            aux = minv @ rhs
        if "aux" in rnd:
            aux = f32(aux)
        prev = h.copy()
        h = R.prox(kind, aux, dual, rho=rho, lam=lam)
        if "x" in rnd:
            h = f32(h)
        dual = dual + h - aux
        if "u" in rnd:
            dual = f32(dual)
        ran = j + 1
        if R.inner_stop(h, prev, aux, dual):
            break
    return h, dual, ran


def run(v, k, reg_w, reg_h, iters, admm_iter, rnd):
    w, h = R.start_factors(v, k, (True, "zero"))
    if "x" in rnd:
        w, h = f32(w), f32(h)
    dw, dh = np.zeros_like(w), np.zeros_like(h)
    inner = []
    for i in range(iters):
        h, dh, nh = block(v, w, h, dh, k, reg_h[1], reg_h[0], admm_iter, rnd)
        wt, dwt, nw = block(v.T, h.T, w.T, dw.T, k, reg_w[1], reg_w[0], admm_iter, rnd)
        w, dw = wt.T, dwt.T
        inner.append((nh, nw))
    return w, h, inner


if __name__ == "__main__":
    m, n, k, T, it = 320, 448, 100, 16, 14
    if len(sys.argv) > 1:
        m, n, k, T, it = (int(a) for a in sys.argv[1:6])
    v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32).astype(np.float64)
    nv = np.linalg.norm(v)
    w0, h0, in0 = run(v, k, (0.02, "l1n"), (0, "nn"), it, T, ())
    base = ("x", "u", "aux", "b", "g", "minv", "rhs", "gb32")
    sets = [base + ("s33", "b3"), base + ("s33", "b4"), base + ("s22", "b3")]
    for rnd in sets:
        w, h, inn = run(v, k, (0.02, "l1n"), (0, "nn"), it, T, rnd)
        print(f"{'+'.join(rnd) or 'f64':28s} WH err {np.linalg.norm(w @ h - w0 @ h0) / nv:.3e}  inner same: {inn == in0}", flush=True)
