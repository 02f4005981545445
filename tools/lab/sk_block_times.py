"""Experiment (build with NMFX_EXTRA_DEFS=-DNMFX_EXP_BLOCKTIME): per-block timeline of the two stream-K products of AO-ADMM config 3
(r4): durations of the workers and of the side job (block `workers`: Gram slab sum + the f64 inversion)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix

m, n, k, T = 16384, 8192, 128, 10
if len(sys.argv) > 3:
    m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01
eng = Engine(m, n, k)
eng.upload_v(v); eng.set_factors(w0, h0)
eng.aoadmm_run(0, 1, 0.1, 1, 0.1, T, 10 ** 12, 1e-3, 1e-3, 0, 40)
eng.synchronize()
buf = (C.c_ulonglong * (2 * 2 * 1024))()
assert eng.lib.nmfx_debug_block_times(buf) == 0
t = np.array(buf, dtype=np.float64).reshape(2, 2, 1024) / 100.0      # wall clock: 100 MHz -> us
def plan(units, ngroups, rb, workers, cross):
    def deal(T, ends=None):
        u = 0; nw = 0
        while u < units:
            cost = 0; v = u; first = True
            while v < units:
                b = v // ngroups; room = (b + 1) * ngroups - v
                extra = 0 if first else cross
                if cost + extra + 1 > T: break
                take = min(room, T - cost - extra)
                cost += extra + take; v += take; first = False
                if take < room: break
            if v == u: v = u + 1
            u = v; nw += 1
            if ends is not None: ends.append(u)
        return nw
    lo, hi = 1, units + cross * rb
    while lo < hi:
        mid = (lo + hi) // 2
        if deal(mid) <= workers: hi = mid
        else: lo = mid + 1
    ends = []; deal(lo, ends)
    return lo, ends

cross = int(os.environ.get("NMFX_SK_CROSS", "5"))
for side, (name, a) in enumerate((("wphase_noobj (W side)", t[0]), ("hphase (H side, with objective)", t[1]))):
    R, G = (m, n // 64) if side == 0 else (n, m // 64)
    mp = (R + 127) // 128 * 128
    T_, ends = plan(mp // 128 * G, G, mp // 128, 255, cross)
    nsegs = [((e - 1) // G) - ((ends[i - 1] if i else 0) // G) + 1 for i, e in enumerate(ends)]
    ngr = [e - (ends[i - 1] if i else 0) for i, e in enumerate(ends)]
    nb = int((a[0] > 0).sum())
    a = a[:, :nb]
    s, e = a[0] - a[0].min(), a[1] - a[0].min()
    d = e - s
    side = os.environ.get("NMFX_AO_OVERLAP", "1") != "0"
    wk = slice(0, nb - 1) if side else slice(0, nb)
    print(name, "blocks", nb, "start spread %.1f us; worker durations min %.1f med %.1f max %.1f; worker end max %.1f; SIDE JOB start %.1f end %.1f" % (
        s.max(), d[wk].min(), np.median(d[wk]), d[wk].max(), e[wk].max(), s[nb - 1], e[nb - 1]))
    dd = np.sort(d[wk])
    print("   worker duration deciles:", [round(float(dd[int(i * (len(dd) - 1) / 10)]), 1) for i in range(11)])
    if os.environ.get("NMFX_AO_OVERLAP", "1") != "0" and len(ends) == nb - 1:
        for ns in (1, 2):
            sel = [i for i in range(len(ends)) if nsegs[i] == ns]
            if sel:
                print("   %d-segment workers: %d, groups %.1f, duration mean %.1f us (%.2f us per group)" % (
                    ns, len(sel), np.mean([ngr[i] for i in sel]), d[sel].mean(), d[sel].mean() / np.mean([ngr[i] for i in sel])))
    order = np.argsort(e[wk])
    print("   latest workers:", [(int(b), round(float(e[b]), 1)) for b in order[-6:]])
