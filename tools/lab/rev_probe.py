"""Does the Infinity Cache serve the turn-around of a V-sized stream?  Library built with -DNMFX_EXP_REVERSE (tools/lab/ab/
libnmfx_rev.so): the product launch of profile_repeat walks its groups forwards every time (NMFX_EXP_ALTREV=0) or backwards every
other time (=1), i.e. starts with what the previous launch streamed last.
    python tools/lab/rev_probe.py [cfg2|cfg5shard|small]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CFG = {"cfg2": (16384, 8192, 64), "cfg5shard": (16384, 16384, 128), "small": (8192, 4096, 64), "tiny": (4096, 4096, 64)}
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix
m, n, k = %(shape)r
v = planted_matrix(m, n, 32, seed=0, dtype=np.float32)
rs = np.random.RandomState(1)
with Engine(m, n, k) as e:
    e.upload_v(v)
    e.set_factors(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n)))
    e.mur_run(0, 0, 0, 10 ** 12, 1e-5, 1e-5, 0, 300)
    out = {}
    for which in ("wphase", "hphase"):
        out[which] = round(min(e.profile_repeat(which, reps=100, dist=0) for _ in range(3)) * 1e3, 2)
print("AB " + json.dumps(out))
'''
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
for rnd in range(2):
    for alt in ("0", "1"):
        env = dict(os.environ, NMFX_LIB=os.path.join(ROOT, "tools/lab/ab/libnmfx_rev.so"), NMFX_LIB_LAX="1", NMFX_EXP_ALTREV=alt)
        p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, shape=CFG[cfg])], env=env, capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("AB ")]
        print(cfg, "alternate directions" if alt == "1" else "always forwards    ", line[0][3:] if line else "FAILED " + p.stderr[-500:], flush=True)
