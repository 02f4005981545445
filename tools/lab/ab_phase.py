"""Same-box A/B of the two product kernels: per-launch time over back-to-back launches (nmfx_profile_repeat), one child
process per library build, the builds interleaved and repeated so that clock drift shows up as spread, not as a difference.

    python tools/lab/ab_phase.py cfg4 nmf_amd/lib/libnmfx.so tools/lab/libnmfx_base.so [--rounds 3] [--env NMFX_ABLATE=1]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CFG = {"cfg2": (16384, 8192, 64, 0), "cfg4": (32768, 16384, 64, 1), "cfg5": (131072, 16384, 128, 0),
       "cfg3": (16384, 8192, 128, 0), "shard8": (2048, 8192, 64, 0)}

CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
os.environ.setdefault("NMF_AMD_QUIET", "1")
import torch
from nmf_amd.engine import Engine
m, n, k, dist = %(shape)r
g = torch.Generator(device="cuda"); g.manual_seed(0)
v = (torch.rand(m, k, device="cuda", generator=g) @ torch.rand(k, n, device="cuda", generator=g)) / k
v += 0.01 * torch.rand(m, n, device="cuda", generator=g)
eng = Engine(m, n, k)
torch.cuda.synchronize()
eng.upload_v_device(v.data_ptr(), m); del v
w0 = torch.rand(m, k, generator=torch.Generator().manual_seed(1)).abs().numpy().astype("float64")
h0 = torch.rand(k, n, generator=torch.Generator().manual_seed(2)).abs().numpy().astype("float64")
eng.set_factors(w0, h0)
out = {}
for which in ("wphase", "hphase"):
    ts = [eng.profile_repeat(which, reps=%(reps)d, dist=dist) * 1e3 for _ in range(3)]
    out[which] = min(ts)
print("AB " + json.dumps(out))
'''


def main():
    args = sys.argv[1:]
    rounds, envs, reps = 3, {}, 60
    while "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    while "--reps" in args:
        i = args.index("--reps"); reps = int(args[i + 1]); del args[i:i + 2]
    while "--env" in args:
        i = args.index("--env"); kk, vv = args[i + 1].split("=", 1); envs[kk] = vv; del args[i:i + 2]
    cfg, libs = args[0], args[1:]
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            env = dict(os.environ, NMFX_LIB=os.path.abspath(l), **envs)
            p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, shape=CFG[cfg], reps=reps)], env=env,
                               capture_output=True, text=True)
            line = [x for x in p.stdout.splitlines() if x.startswith("AB ")]
            if not line:
                print(l, "FAILED", p.stderr[-800:]); continue
            res[l].append(json.loads(line[0][3:]))
    for l in libs:
        w = [round(x["wphase"], 1) for x in res[l]]; h = [round(x["hphase"], 1) for x in res[l]]
        print(f"{l:40s} W {w}  H {h}")


if __name__ == "__main__":
    main()
