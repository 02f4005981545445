"""Timeline of the blocked f64 Gauss-Jordan (`ao_prepare_mfma_kernel`, k padded to 128) inside a config-3-like AO-ADMM run.

    NMFX_EXTRA_DEFS=-DNMFX_EXP_STAMPS python -m nmf_amd.build && python tools/lab/prep_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

m, n, k = 4096, 2048, 128
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
eng = Engine(m, n, k)
eng.upload_v(v)
eng.set_factors(rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01)
eng.aoadmm_run(0, 1, 0.1, 1, 0.1, 10, 10 ** 12, 1e-3, 1e-3, 0, 12)
eng.synchronize()
out = np.zeros((9, 40), dtype=np.uint64)
assert eng.lib.nmfx_debug_prep_stamps(out.ctypes.data_as(C.c_void_p)) == 0
t = (out.astype(np.int64) - int(out[:, 0].min())) / 100.0          # us; row 8 = the helper wave
print("start (per wave)    :", np.round(t[:, 0], 2))
print("loaded              :", np.round(t[:, 1], 2))
print("first tile inverted :", round(t[8, 2], 2))
for kb in range(8):
    print("step %d: barrier %.2f | mid-step barrier %.2f | row waves done %.2f .. %.2f | helper done %.2f"
          % (kb, t[:, 3 + 3 * kb].max(), t[:, 4 + 3 * kb].max() if kb < 7 else float("nan"), t[:8, 5 + 3 * kb].min(), t[:8, 5 + 3 * kb].max(),
             t[8, 5 + 3 * kb]))
print("end                 :", np.round(t[:8, 30], 2))
