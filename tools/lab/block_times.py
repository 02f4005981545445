"""Experiment (needs a build with NMFX_EXTRA_DEFS=-DNMFX_EXP_BLOCKTIME): spread of the per-block
durations of the two product kernels of MUR-eu config 2 -- how much a dynamic work split could gain."""
import ctypes as C
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix

m, n, k = 16384, 8192, 64
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
eng = Engine(m, n, k)
eng.upload_v(v); eng.set_factors(w0, h0)
eng.mur_run(0, 0.0, 0.0, 10 ** 12, 1e-5, 1e-5, 0, 600)
eng.synchronize()
buf = (C.c_ulonglong * (2 * 2 * 1024))()
assert eng.lib.nmfx_debug_block_times(buf) == 0
t = np.array(buf, dtype=np.float64).reshape(2, 2, 1024)[:, :, :256] / 100.0      # wall clock: 100 MHz -> us
for name, a in (("hphase", t[0]), ("wphase", t[1])):
    s, e = a[0] - a[0].min(), a[1] - a[0].min()
    d = e - s
    print(name, "start spread %.1f us; durations min %.1f med %.1f max %.1f; end min %.1f med %.1f max %.1f" % (
        s.max(), d.min(), np.median(d), d.max(), e.min(), np.median(e), e.max()))
    order = np.argsort(e)
    print("   latest blocks:", [(int(b), round(float(e[b]), 1)) for b in order[-8:]])
    print("   per-XCD median end:", [round(float(np.median(e[x::8])), 1) for x in range(8)])
