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

rsag = "--rsag" in sys.argv       # r5: reduce-scatter . sliced H update . all-gather inside nmfx_mur_run_sharded; --world N: the slice of one of N ranks
argv = [a for a in sys.argv if a != "--rsag"]
world = 1
if "--world" in argv:
    i = argv.index("--world"); world = int(argv[i + 1]); del argv[i:i + 2]
rows, n, k = (int(x) for x in argv[1:4]) if len(argv) > 3 else (16384, 16384, 128)
steps = 40
v = planted_matrix(rows, n, min(k, 64), seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(rows, k)), np.abs(rs.randn(k, n))
NEVER = 10 ** 12
shard = nd.NativeShard(v, k, w0, h0, 0)
comm = nd.NativeComm(shard, 0, 1, shard.eng.comm_unique_id())
shard.negotiate(comm)
e = shard.eng
if rsag:
    e.comm_set_exchange(1)
    print("exchange: reduce-scatter + all-gather, slice info", e.mur_slice_info(0, 1), flush=True)
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
       ("wphase", "hphase", "gram_tn", "gram_nt", "sum_hht", "w_update", "pack", "h_update", "h_unpack", "small") if e.profile_get(nm)[1]})
if world > 1:
    # what ONE of `world` ranks would launch per step with the reduce-scatter / all-gather exchange: phase A, the H update of n / world
    # columns, the other ranks' columns unpacked (TIMING ONLY: with a world of one the gathered columns are not those of any rank)
    cols = e.mur_slice_info(0, 1)[0] // world // 64 * 64
    e.set_factors(w0, h0)
    e.profile_reset()
    for j in range(6):
        e.mur_phase_a(0, 0.0, j)
        e.mur_phase_b_slice(0, 0.0, NEVER, 1e-5, 1e-5, j, 0, cols)
        e.mur_phase_b_rest(0, 0, cols)
    e.synchronize()
    print("one of %d ranks, sliced phase B (%d columns):" % (world, cols),
          {nm: round(e.profile_get(nm)[0] / max(1, e.profile_get(nm)[1]) * 1e3, 1) for nm in
           ("wphase", "hphase", "gram_tn", "gram_nt", "sum_hht", "w_update", "pack", "h_update", "h_unpack", "small") if e.profile_get(nm)[1]}, flush=True)
comm.close(); shard.close()
if world > 1:
    sys.exit(0)
with Engine(rows, n, k) as e:
    e.upload_v(v); e.set_factors(w0, h0)
    e.mur_run(0, 0, 0, NEVER, 1e-5, 1e-5, 0, 100); e.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        e.mur_run(0, 0, 0, NEVER, 1e-5, 1e-5, 100 + rep * steps, steps); e.synchronize()
        print("single-GPU loop: %.1f us per step" % ((time.perf_counter() - t0) / steps * 1e6), flush=True)
