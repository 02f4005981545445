"""Same-box A/B of whole MUR iterations between ENVIRONMENT settings of one library build (children interleaved), with a
checksum of the factors after a fixed number of iterations so that a knob that must not change results is seen not to:
    python tools/lab/ab_env.py cfg2 base: nw4w:NMFX_NW4=1 nw4h:NMFX_NW4=2 nw4:NMFX_NW4=3 [--rounds 3]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CFG = {"cfg2": (16384, 8192, 64, 0), "cfg4": (32768, 16384, 64, 1), "cfg3shape": (16384, 8192, 128, 0), "small": (640, 384, 24, 0),
       "shard8": (2048, 8192, 64, 0), "shard4": (4096, 8192, 64, 0), "shard2": (8192, 8192, 64, 0), "c5shard8": (16384, 16384, 128, 0)}
CHILD = r'''
import os, sys, json, time, hashlib
sys.path.insert(0, %(root)r)
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix
m, n, k, dist = %(shape)r
v = planted_matrix(m, n, min(k, 64), seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
NEVER = 10 ** 12
with Engine(m, n, k) as e:
    e.upload_v(v)
    e.set_factors(w0, h0)
    e.mur_run(dist, 0.1, 0.2, NEVER, 1e-5, 1e-5, 0, 12)
    e.synchronize()
    w, h = e.get_factors()
    obj = e.objectives(0, 12)
    out = {"wh": hashlib.md5(np.ascontiguousarray(w).tobytes() + np.ascontiguousarray(h).tobytes()).hexdigest()[:10],
           "obj11": float(obj[11])}
    e.mur_run(dist, 0, 0, NEVER, 1e-5, 1e-5, 12, 400)
    e.synchronize()
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        e.mur_run(dist, 0, 0, NEVER, 1e-5, 1e-5, 412 + 200 * r, 200)
        e.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200)
    e.profile_enable(True); e.profile_reset()
    e.mur_run(dist, 0, 0, NEVER, 1e-5, 1e-5, 1012, 40)
    e.synchronize()
    out["iter_us"] = round(best * 1e6, 2)
    for name in ("wphase", "hphase", "w_update", "h_update", "row_sums", "small"):
        ms, cnt = e.profile_get(name)
        if cnt: out[name] = round(ms / cnt * 1e3, 2)
print("AB " + json.dumps(out))
'''
args = sys.argv[1:]
rounds = 3
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
cfg, variants = args[0], args[1:]
for r in range(rounds):
    for var in variants:
        name, _, kv = var.partition(":")
        env = dict(os.environ)
        for item in filter(None, kv.split(",")):
            k_, _, v_ = item.partition("=")
            env[k_] = v_
        p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, shape=CFG[cfg])], env=env, capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("AB ")]
        print(f"{name:10s}", line[0][3:] if line else "FAILED " + p.stderr[-600:], flush=True)
