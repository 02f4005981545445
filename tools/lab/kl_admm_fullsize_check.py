"""r5: the KL-loss ADMM variants at the config-3 shape against the oracle (a lab check, not in the suite: the oracle needs about a minute per
solver here).  python tools/lab/kl_admm_fullsize_check.py [m n k outer [ao_admm|admm]]  ->  one JSON line per solver: WH error, objective history difference,
inner counts equal (AO-ADMM).  Fused auxiliaries + gathered products are the default path (NMFX_KL_FUSE / NMFX_KL_GATHER = 0 for the others)."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.ao_admm import ao_admm
from nmf_amd.admm import admm

m, n, k, outer = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (16384, 8192, 128, 2)
v = R.planted_matrix(m, n, min(k, 64), seed=0, dtype=np.float32)
nv = float(np.linalg.norm(v.astype(np.float64)))
from nmf_amd.engine import Engine
from nmf_amd import utils
for name, fn, ref_fn, kw in (
        ("ao_admm kl", ao_admm, R.ao_admm, dict(reg_w=(0, "nn"), reg_h=(0.02, "l1n"), admm_iter=10)),
        ("admm kl", admm, R.admm, dict(rho=1.0, reg_w=(0, "nn"), reg_h=(0, "nn")))):
    if len(sys.argv) > 5 and sys.argv[5] not in name.split():
        continue
    kw = dict(kw, distance_type="kl", min_iter=outer, max_iter=outer)
    t0 = time.time()
    with Engine(m, n, k) as eng:                        # (NNDSVD start from the device's singular triplets, handed to both sides: tests/test_gpu_fullsize.py)
        eng.upload_v(v)
        w0, h0 = utils.nndsvd_device(eng, v, k, "zero")
        res = fn(v, k, nndsvd_init=(True, "zero"), engine=eng, **kw)
    t_dev = time.time() - t0
    inner = [list(map(int, t)) for t in ao_admm.last_inner_counts] if fn is ao_admm else None
    t0 = time.time()
    with np.errstate(all="ignore"):
        ref = ref_fn(v, k, w0=w0, h0=h0, **kw)
    t_ref = time.time() - t0
    err = 0.0
    for r0 in range(0, m, 2048):
        d = res.w[r0:r0 + 2048] @ res.h - ref.w[r0:r0 + 2048] @ ref.h
        err += float(np.sum(d * d))
    out = {"solver": name, "shape": [m, n, k], "outer": outer, "wh_rel_err": float(np.sqrt(err) / nv),
           "obj_max_rel_diff": float(np.max(np.abs(np.asarray(res.obj_history) - np.asarray(ref.obj_history)) / np.abs(ref.obj_history))),
           "obj_history": [float(x) for x in res.obj_history], "i": [int(res.i), int(ref.i)],
           "seconds_device_incl_svd": round(t_dev, 2), "seconds_oracle": round(t_ref, 1)}
    if inner is not None:
        out["inner_counts_equal"] = [tuple(t) for t in inner] == [tuple(t) for t in ref.trace["inner"]]
        out["inner_counts"] = inner
    print(json.dumps(out), flush=True)
