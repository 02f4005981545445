"""Where the cycles of the 32-row product kernel go (build with NMFX_EXTRA_DEFS=-DNMFX_EXP_STAMPS):
per wave, the shader cycles spent in the segments of the group loop, summed over the groups of the last launch
of each kind (W phase = with objective, H phase = without), and the clock the chip held.

    NMFX_EXTRA_DEFS=-DNMFX_EXP_STAMPS python -m nmf_amd.build && python tools/lab/stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

kl = "--kl" in sys.argv          # MUR-KL: segments = wait + barrier | first product (+ objective terms of the group before) | second product (+ quotient, split)
argv = [a for a in sys.argv if a != "--kl"]
m, n, k = (int(x) for x in argv[1:4]) if len(argv) > 3 else (16384, 8192, 64)
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
eng = Engine(m, n, k)
eng.upload_v(v)
eng.set_factors(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n)))
eng.mur_run(1 if kl else 0, 0, 0, 10 ** 12, 1e-5, 1e-5, 0, (100 if kl else 600) if k <= 64 else 150)
eng.synchronize()
out = np.zeros((2, 256, 8, 6), dtype=np.uint64)
assert eng.lib.nmfx_debug_stamps(out.ctypes.data_as(C.c_void_p)) == 0
for kind, name in ((1, "W phase (objective)"), (0, "H phase")):
    a = out[kind].astype(np.float64)
    tot, rt = a[..., 4], a[..., 5]
    print(name, "median block: %.0f cycles, %.1f us, clock %.2f GHz" % (np.median(tot), np.median(rt) / 100.0, np.median(tot / rt) / 10.0))
    blk = rt.max(axis=1) / 100.0
    print("   block durations (us): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f; the 8 slowest blocks: %s"
          % (blk.min(), np.percentile(blk, 10), np.median(blk), np.percentile(blk, 90), blk.max(), np.argsort(blk)[-8:].tolist()))
    for role, ws in (("V loaders (waves 0-3)", slice(0, 4)), ("Y loaders (waves 4-7)", slice(4, 8))):
        seg = np.median(a[:, ws, :4].reshape(-1, 4), axis=0)
        groups = (n // 64) / max(1, round(256 / (m // 128)))     # groups per block: the contraction over the splits that fill 256 CUs
        labels = ("wait+barrier", "first product (+ objective)", "-", "second product (+ quotient, split)") if kl else \
            ("wait+barrier", "head (LDS reads + split)", "early barrier", "MFMA stages")
        print("   %-22s %s %6.0f  %s %6.0f  %s %6.0f  %s %6.0f  | per group (%d groups): %s"
              % (role, labels[0], seg[0], labels[1], seg[1], labels[2], seg[2], labels[3], seg[3], groups, np.round(seg / groups, 0)))
