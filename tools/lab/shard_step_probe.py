"""r4: one rank's step of the row-sharded MUR loop with a world of one on RCCL (nmfx_mur_run_sharded) against the single-GPU loop on
the same shard -- where do the extra microseconds go?  python tools/lab/shard_step_probe.py [rows n k]   (under rocprofv3 for the trace)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
from nmf_amd import dist as nd
from nmf_amd.engine import Engine
from nmf_amd.synth import planted_matrix

rows, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 16384, 128)
steps = 40
v = planted_matrix(rows, n, min(k, 64), seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(rows, k)), np.abs(rs.randn(k, n))
NEVER = 10 ** 12
shard = nd.NativeShard(v, k, w0, h0, 0)
comm = nd.NativeComm(shard, 0, 1, shard.eng.comm_unique_id())
shard.negotiate(comm)
e = shard.eng
e.mur_run_sharded(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 0, 100)
e.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    e.mur_run_sharded(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 100 + rep * steps, steps)
    e.synchronize()
    print("sharded loop (world of one): %.1f us per step" % ((time.perf_counter() - t0) / steps * 1e6), flush=True)
e.profile_enable(True); e.profile_reset()
e.mur_run_sharded(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 400, 10)
e.synchronize()
print({nm: round(e.profile_get(nm)[0] / max(1, e.profile_get(nm)[1]) * 1e3, 1) for nm in
       ("wphase", "hphase", "gram_tn", "gram_nt", "sum_hht", "w_update", "pack", "h_update", "small") if e.profile_get(nm)[1]})
comm.close(); shard.close()
with Engine(rows, n, k) as e:
    e.upload_v(v); e.set_factors(w0, h0)
    e.mur_run(0, 0, 0, NEVER, 1e-5, 1e-5, 0, 100); e.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        e.mur_run(0, 0, 0, NEVER, 1e-5, 1e-5, 100 + rep * steps, steps); e.synchronize()
        print("single-GPU loop: %.1f us per step" % ((time.perf_counter() - t0) / steps * 1e6), flush=True)
