#!/usr/bin/env python3
"""Headline benchmark: NMF outer iterations/sec, MUR-Euclidean,
V = 16384 x 8192 float32, k = 64 (BASELINE.json configs[1]) on N MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one outer iteration of the reference's loop (nmf/mur.py:122-127):
W update, H update with the new W, objective of the result.  V is resident in
HBM before the timed region.  For N > 1 the SAME matrix is row-sharded over the
ranks (strong scaling) with one RCCL all-reduce of [W^T V | W^T W | objective]
per iteration.  Rank 0 prints one JSON line.

`roofline`: the dominant kernel (the W phase: A = V H^T with the fused residual
objective), mean launch time from HIP events on the engine's stream in a
separate pass of back-to-back launches.  In the default arithmetic (split bf16: each f32 operand
as bf16 hi + lo, three bf16 MFMA terms, f32 accumulation) the kernel is HBM
bound: achieved = algorithmic bytes (V read once: m*n*4, plus the A slabs) over
time against 8 TB/s.  With NMFX_PRECISION=f32 (exact f32-input MFMA) it is MFMA
bound: algorithmic flops 2*m*n*k (the objective's second product is executed but
not counted) against the 157.3 TFLOP/s f32 matrix peak.
`cpu_baseline`: the numpy oracle (the reference's literal evaluation order:
six m*n*k GEMMs per iteration, float64 factors) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("NMF_AMD_QUIET", "1")

import numpy as np  # noqa: E402

from nmf_amd.synth import limit_blas_threads  # noqa: E402

HOST_THREADS = limit_blas_threads()      # BLAS pool = the CPUs the container may use (see nmf_amd.synth.usable_cpus)

M, N, K = 16384, 8192, 64
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md, dense bf16 matrix peak (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    # (NMFX_BENCH_SHAPE=MxNxK: the same through the environment -- torch.distributed.run's own parser chokes on "--m")
    dm, dn, dk = (int(x) for x in os.environ.get("NMFX_BENCH_SHAPE", f"{M}x{N}x{K}").split("x"))
    ap.add_argument("--m", type=int, default=dm)
    ap.add_argument("--n", type=int, default=dn)
    ap.add_argument("--k", type=int, default=dk)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-iters", type=int, default=10)     # ~12 s of host work at 0.9 iterations/s
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--tol-max-iter", type=int, default=20000,
                    help="time-to-tol leg: iteration cap of the run to the reference's default tolerances (0 = skip)")
    ap.add_argument("--preheat", type=int, default=400,
                    help="untimed iterations run BEFORE the W warm-up steps to bring the device to its sustained "
                         "clocks (the first ~50 iterations after idle run 15 %% slower); the factors are reset "
                         "afterwards, so the W + K steps start from the same state as without it (0 = off)")
    ap.add_argument("--no-others", action="store_true", help="skip the other_configs legs (configs 3, 4, 5-on-1-GPU)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def hbm_traffic(kernel_prefix, m, n, k):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
    passes (--pmc with --kernel-trace only), FETCH_SIZE (KB) doubled because gfx950
    reports half of the bytes of wide coalesced streaming reads.  Each pass runs this
    script again as a child (`-- python3 bench.py --pmc-child`).  None when rocprofv3 is
    unavailable or fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="nmfx_pmc_")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "3", "--warmup", "1",
                   "--m", str(m), "--n", str(n), "--k", str(k)]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=240, cwd="/tmp")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            got = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                   if row["Counter_Name"] == ctr and kernel_prefix in row["Kernel_Name"]]
            if not got:
                return None
            vals[ctr] = sum(got) / len(got)
        except Exception:  # noqa: BLE001
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return vals["FETCH_SIZE"] * 1024.0 * 2.0 + vals["WRITE_SIZE"] * 1024.0


def cpu_baseline(v, k, iters):
    """The oracle's MUR-eu loop body (reference evaluation order) on the host: 1 warm-up + `iters` timed
    iterations from the SAME start as the GPU run.  Returns (iterations/s, W, H, objective history of the
    1 + iters iterations incl. the initial value) -- the factors feed the `parity` block of the bench line."""
    from oracle import nmf_ref as R
    rs = np.random.RandomState(0)
    w = np.abs(rs.randn(v.shape[0], k))
    h = np.abs(rs.randn(k, v.shape[1]))
    wh = w @ h
    hist = [R.objective(v, wh, "eu")]

    def one(w, h, wh):
        w = R.mur_w_step("eu", v, w, h, wh, 0.0)
        h = R.mur_h_step("eu", v, w, h, w @ h, 0.0)
        wh = w @ h
        return w, h, wh, R.objective(v, wh, "eu")

    w, h, wh, o = one(w, h, wh)          # warm-up
    hist.append(o)
    t0 = time.perf_counter()
    for _ in range(iters):
        w, h, wh, o = one(w, h, wh)
        hist.append(o)
    dt = time.perf_counter() - t0
    return iters / dt, w, h, np.asarray(hist)


def converge_on_device(eng, w0, h0, tol, max_iter, min_iter=100):
    """MUR-eu from (w0, h0) until the reference's stop rule (nmf/utils.py:4-15, tol1 = tol2 = tol, nmf/mur.py:131 `i > min_iter`)
    fires, or max_iter: the product's own loop (nmf_amd._driver.drive, 64 queued iterations per host round trip) with its float64
    referee of the stop rule (nmf_amd._driver.Referee).  Returns (rule, stop index i, iterations run, seconds, referee)."""
    from nmf_amd._driver import Referee, drive
    NEVER = 10 ** 15
    eng.set_factors(w0, h0)
    eng.synchronize()
    referee = Referee(eng, lambda i: eng.mur_run(0, 0.0, 0.0, NEVER, tol, tol, i, 1), min_iter, tol, tol)
    t1 = time.perf_counter()
    i, history = drive(eng, lambda first, count: eng.mur_run(0, 0.0, 0.0, min_iter, tol, tol, first, count),
                       lambda done: eng.mur_finish(0, NEVER if referee.walked else min_iter, tol, tol, done), max_iter, tol, tol, referee=referee)
    secs = time.perf_counter() - t1
    referee.history = history
    return int(referee.final_rule), int(i if referee.final_rule else -1), int(i + 1 if referee.final_rule else max_iter), secs, referee


def oracle_stop_check(eng, v, w0, h0, tol, device_rule, device_i, lead=15, span=30, min_iter=100):
    """The time-to-tol half of the metric, pinned at the full size (part of the cpu_baseline leg: the oracle is the CHECKER here).
    The device says the reference's stop rule fires at outer iteration `device_i` with rule `device_rule`.  Its run is bit-stable,
    so running it again for s = device_i - lead iterations gives the iterate (W_s, H_s) it passed through; the float64 oracle
    (reference evaluation order, nmf/mur.py:119-136) continues from there for `span` iterations with the reference's own
    convergence_check on ITS float64 objective values.  Both must stop at the same index by the same rule."""
    from oracle import nmf_ref as R
    s = max(0, device_i - lead)
    eng.set_factors(w0, h0)
    eng.mur_run(0, 0.0, 0.0, min_iter, tol, tol, 0, s)
    w, h = eng.get_factors()
    dev_obj = None
    wh = w @ h
    hist = [R.objective(v, wh, "eu")]
    oracle_i, oracle_rule = -1, 0
    for t in range(span):
        i = s + t
        w = R.mur_w_step("eu", v, w, h, wh, 0.0)
        h = R.mur_h_step("eu", v, w, h, w @ h, 0.0)
        wh = w @ h
        hist.append(R.objective(v, wh, "eu"))
        if i > min_iter:
            oracle_rule = R.stop_rule(hist[-1], hist[-2], tol, tol)
            if oracle_rule:
                oracle_i = i
                break
    dec = -np.diff(np.asarray(hist))
    return {"tol": tol, "device_i": int(device_i), "device_rule": int(device_rule), "oracle_i": int(oracle_i), "oracle_rule": int(oracle_rule),
            "agree": bool(oracle_i == device_i and oracle_rule == device_rule), "snapshot_iteration": int(s),
            "oracle_iterations_run": int(len(hist) - 1),
            "oracle_decrease_last_two": [float(x) for x in dec[-2:]],
            "note": "f64 oracle continued from the device's iterate `lead` iterations before its stop; convergence_check on the "
                    "oracle's own objective values (nmf/utils.py:4-15)"}


def parity_block(v, w_g, h_g, obj_g, w_r, h_r, obj_r, block=2048):
    """||W_g H_g - W_r H_r||_F / ||V||_F (north_star's bar: < 1e-4) by row blocks + objective histories."""
    num = den = 0.0
    for a in range(0, v.shape[0], block):
        b = min(v.shape[0], a + block)
        d = w_g[a:b] @ h_g - w_r[a:b] @ h_r
        num += float(np.sum(d * d))
        vb = v[a:b].astype(np.float64)
        den += float(np.sum(vb * vb))
    return {"wh_rel_err": float(np.sqrt(num / den)),
            "obj_max_rel_diff": float(np.max(np.abs(obj_g - obj_r) / np.abs(obj_r))),
            "iterations": int(len(obj_r) - 1), "tolerance": 1e-4,
            "against": "numpy oracle (reference evaluation order, float64 factors), same V, same |randn| start"}


ALL_KERNELS = ("wphase", "wphase_noobj", "objective", "hphase", "gram_nt", "gram_tn", "sum_hht", "w_update", "pack",
               "h_update", "images", "row_sums", "prepare", "inner_h", "inner_w", "sums", "kl_vaux", "kl_vaux_fused", "kl_round_h", "kl_round_w", "transpose",
               "nnls", "small")
V_SIZED = ("wphase", "wphase_noobj", "objective", "hphase", "kl_vaux", "kl_vaux_fused")      # launches that stream V (or V^T) once
KL_STREAMS = {"kl_vaux": 4, "kl_vaux_fused": 3}      # V-sized streams of the KL auxiliaries' launches: V and dual_v read, dual_v (and S) written


def device_planted(eng, torch, m, n, k, seed, dev, chunk=8192, rows=None):
    """The synthetic matrix of SURVEY 8d -- (U1 U2) / k + 0.01 U3, U* ~ U(0, 1) -- drawn ON the device from
    torch's generator and handed to the engine device-to-device (nmfx_upload_v_device): the 2 and 8 GiB
    matrices of configs 4 and 5 would take tens of seconds to draw on the host.  Same distribution as
    nmf_amd.synth.planted_matrix, a different random stream: U1, U2 from `seed`, the noise of rows
    [c * chunk, (c + 1) * chunk) from seed' = seed * 1000003 + 1 + c -- so a rank of a row-sharded run draws
    exactly its rows (`rows = (r0, r1)`) of the SAME global matrix whatever the number of ranks."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    left = torch.rand(m, k, generator=g, device=dev, dtype=torch.float32)
    right = torch.rand(k, n, generator=g, device=dev, dtype=torch.float32)
    r0, r1 = (0, m) if rows is None else rows
    for c in range(r0 // chunk, (r1 + chunk - 1) // chunk):
        a, b = c * chunk, min(m, (c + 1) * chunk)
        g.manual_seed(seed * 1000003 + 1 + c)
        blk = (left[a:b] @ right) / k + 0.01 * torch.rand(b - a, n, generator=g, device=dev, dtype=torch.float32)
        lo, hi = max(a, r0), min(b, r1)
        part = blk[lo - a:hi - a]
        torch.cuda.synchronize()
        eng.upload_v_device(part.data_ptr(), hi - lo, row0=lo - r0)
        del blk, part
    del left, right
    torch.cuda.empty_cache()


def make_sharded(torch, dist, nd, rank, world, local_rank, build):
    """This rank's shard and its exchange: nmf_amd.dist.make_sharded.  The default is torch.distributed's collectives between the
    phase calls; NMFX_DIST_NATIVE=1 runs the exchange BEHIND the C ABI (nd.NativeShard / nd.NativeComm: nmfx_comm_init_rank +
    nmfx_mur_run_sharded, no torch on the data path) -- opt-in until it has run with more than one rank on RCCL; with a
    world of one (NMFX_BENCH_FORCE_SHARDED=1: the one-GPU rehearsal) the native loop is the default, as before.  Every rank takes the
    same branch (the choice is all-reduced before any data-path collective).  Returns (shard, comm, name of the loop)."""
    on_gpu = dist.get_backend() == "nccl"
    env = os.environ.get("NMFX_DIST_NATIVE")
    want = (env == "1") if env is not None else world == 1
    return nd.make_sharded(build, rank, torch.device(f"cuda:{local_rank}"), on_gpu, want_native=want,
                           break_native=os.environ.get("NMFX_BENCH_BREAK_NATIVE") == "1")


def cfg5_sharded(torch, dist, nd, rank, world, local_rank, steps=5, warmup=2, progress=None):
    """BASELINE.json's config 5 as it is meant: MUR Euclidean, V = 131072 x 16384 f32, k = 128, rows of V and W sharded over
    the `world` GPUs, one RCCL all-reduce of [W^T V | W^T W | objective] per iteration (nmf_amd.dist.run_iterations).  The same
    global matrix as other_configs' cfg5_on_1_gpu (device_planted draws any row range of it), so that
    iter/s(N) / iter/s(1) is the strong scaling SURVEY 8d asks for.  Every rank runs this; returns the slot on rank 0.
    NMFX_BENCH_CFG5_SHAPE=MxNxK shrinks it (rehearsals of the code path with several ranks on one GPU)."""
    m, n, k = (int(x) for x in os.environ.get("NMFX_BENCH_CFG5_SHAPE", "131072x16384x128").split("x"))
    NEVER = 10 ** 12
    r0, r1 = nd.row_range(m, rank, world)
    rs = np.random.RandomState(0)                       # nmf/mur.py:108-109
    w0 = np.abs(rs.randn(m, k))[r0:r1]
    h0 = np.abs(rs.randn(k, n))
    dev = torch.device(f"cuda:{local_rank}")
    on_gpu = dist.get_backend() == "nccl"
    made = []

    def build(cls):
        try:
            made.append(cls(None, k, w0, h0, local_rank, shape=(r1 - r0, n),
                            fill=lambda e: device_planted(e, torch, m, n, k, 0, dev, rows=(r0, r1))))
        except Exception as e:  # noqa: BLE001
            made.append(e)
        # every rank learns whether EVERY rank is set up before the first data-path collective (the communicator's own start-up
        # included): a rank that failed alone (out of memory, ...) must not leave the others waiting
        flag = torch.tensor([0 if isinstance(made[-1], Exception) else 1], dtype=torch.int32, device=dev if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if not isinstance(made[-1], Exception):
                made[-1].eng.close()
            raise RuntimeError(f"config-5 shard could not be set up on every rank (rank {rank}: {made[-1]!r})")
        return made[-1]
    # Which loop and which exchange is fastest can only be settled on a multi-GPU node, so the same run times them all (VERDICT r3
    # item 4d, r4 item 5): {torch.distributed collectives between the phase calls, RCCL behind the C ABI} x {one all-reduce per
    # iteration, two column chunks with the all-reduce of chunk 0 behind the product of chunk 1, reduce-scatter . sliced H update .
    # all-gather}.  The slot's iter_per_s is the FASTEST form that ran (named in `exchange`); every form's time rides along.  With
    # NMFX_DIST_NATIVE / NMFX_DIST_CHUNKS / NMFX_DIST_EXCHANGE set, only the form they ask for runs.
    forms = [("allreduce", 1), ("allreduce", 2), ("rsag", 1)]
    loops = ["torch", "native"] if on_gpu else ["torch"]
    if os.environ.get("NMFX_DIST_NATIVE") is not None:
        loops = ["native" if os.environ["NMFX_DIST_NATIVE"] == "1" else "torch"]
    if os.environ.get("NMFX_DIST_CHUNKS") or os.environ.get("NMFX_DIST_EXCHANGE"):
        forms = [(nd.exchange_mode(), max(1, int(os.environ.get("NMFX_DIST_CHUNKS", "1") or 1)))]
    saved_env = {key: os.environ.get(key) for key in ("NMFX_DIST_CHUNKS", "NMFX_DIST_EXCHANGE", "NMFX_DIST_NATIVE")}
    times, slot = {}, None
    flops = 4.0 * m * n * k + 4.0 * k * k * (m + n)
    nbytes = 2.0 * m * n * 4 + 3.0 * (m + n) * k * 4

    def result():
        dt, obj = slot["dt"], slot["obj"]
        return {"config": "cfg5", "workload": f"MUR Euclidean, V={m}x{n} f32 row-sharded over {world} GPU(s), k={k}, |randn| start, objective every "
                                              "iteration; exchange per iteration = the fastest of the forms timed (see `exchange`)",
                "n_gpus": world, "rows_per_gpu": r1 - r0, "iter_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "steps": steps,
                "warmup": warmup, "scaling": "strong (base: other_configs.cfg5_on_1_gpu of the --gpus 1 line, the same matrix)",
                "precision": slot["precision"], "loop": slot["loop"], "objective_first_last": [float(obj[0]), float(obj[-1])],
                "exchange": slot["key"],
                "ms_per_step_by_exchange": {key: (v * 1e3 if isinstance(v, float) else v) for key, v in times.items()},
                "objective_decreasing": True, "algorithmic_gflop_per_iter": flops / 1e9, "algorithmic_gbytes_per_iter": nbytes / 1e9,
                "tflops": flops / dt / 1e12, "hbm_gbs_per_gpu": nbytes / dt / 1e9 / world,
                "frac_of_hbm_peak": nbytes / dt / 1e9 / PEAK_HBM_GBS / world,
                "all_reduce_bytes": slot["ar_bytes"], "collectives_per_iteration": slot["collectives"],
                "data": "synthetic, drawn on the device (torch generator, seed 0; each rank its own rows of the same matrix)"}
    try:
        for kind in loops:
            os.environ["NMFX_DIST_NATIVE"] = "1" if kind == "native" else "0"
            try:
                shard, comm, loop = make_sharded(torch, dist, nd, rank, world, local_rank, build)
            except Exception as e:  # noqa: BLE001  (build() has agreed the failure over the ranks)
                times[f"{kind}/setup"] = f"{type(e).__name__}: {e}"
                continue
            took = "native" if isinstance(comm, nd.NativeComm) else "torch"
            if took != kind:                                    # (the native communicator did not come up: the torch loop was timed already, or is next)
                times[f"{kind}/setup"] = "fell back to torch.distributed's collectives"
                shard.eng.close()
                del shard
                torch.cuda.empty_cache()
                continue
            try:
                def fence():
                    shard.eng.synchronize()
                    torch.cuda.synchronize()
                    if isinstance(comm, nd.NativeComm):         # (the barrier through the communicator the data path uses, see main())
                        comm.barrier()
                    else:
                        dist.barrier()
                    torch.cuda.synchronize()

                def timed():
                    # a fresh Runner per form: it reads the exchange mode when it is made
                    run = nd.Runner(shard, comm, 0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 2 * (warmup + steps) + 8)
                    # untimed rehearsal (lazy allocations, pools) -- and long enough for the sustained clocks: a rank's step is 0.7 ms at N = 8, and
                    # after idle the device needs some 15 ms of work to get there (protocol note; r4: 7 rehearsal steps read 751 us per
                    # step where 60 read 628) -- then from the start again
                    t_w = time.perf_counter()
                    while True:
                        shard.eng.set_factors(w0, h0)
                        run(0, warmup + steps)
                        fence()
                        # (every rank must leave this loop in the same round: the decision is rank 0's)
                        go = torch.tensor([1 if time.perf_counter() - t_w > 0.15 else 0], dtype=torch.int32, device=dev if on_gpu else "cpu")
                        dist.broadcast(go, src=0)
                        if int(go.item()):
                            break
                    shard.eng.set_factors(w0, h0)
                    run(0, warmup)
                    fence()
                    t0 = time.perf_counter()
                    run(warmup, steps)
                    fence()
                    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if on_gpu else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    _, _, n_obj = shard.eng.state()
                    obj = shard.eng.objectives(0, n_obj)
                    ok = bool(np.all(np.isfinite(obj)) and obj[-1] < obj[0])
                    return float(t.item()) / steps, obj, ok, run.mode

                for mode, nch in forms:
                    os.environ["NMFX_DIST_EXCHANGE"] = mode
                    os.environ["NMFX_DIST_CHUNKS"] = str(nch)
                    key = f"{kind}/{mode}" + (f"/{nch} chunks" if nch > 1 else "")
                    if progress is not None:
                        progress["running"] = key          # (what a deadline that expires would name)
                    try:
                        dt, obj, ok, rmode = timed()
                    except Exception as e:  # noqa: BLE001  (every rank runs the same sequence: a failure here is the same on all of them)
                        times[key] = f"{type(e).__name__}: {e}"
                        continue
                    times[key] = dt
                    if ok and (slot is None or dt < slot["dt"]):
                        slot = {"dt": dt, "key": key, "obj": obj, "loop": loop + " / " + rmode, "precision": shard.eng.precision(),
                                "ar_bytes": float(shard.xf32.numel() * 4 if shard.merge_objective() else shard.xf32.numel() * 4 + 64),
                                "collectives": (1 if shard.merge_objective() else 2) if mode == "allreduce" else 2}
                    if progress is not None and slot is not None:
                        progress["result"] = result()      # (the best of the forms that have run so far)
            finally:
                if isinstance(comm, nd.NativeComm):
                    try:
                        comm.close()
                    except Exception:  # noqa: BLE001
                        pass
                shard.eng.close()
                del shard
                torch.cuda.empty_cache()
    finally:
        for key, val in saved_env.items():
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
    if slot is None:
        raise RuntimeError(f"no form of the sharded loop ran: {times}")
    return result()


def scaling_model(torch, dev, base_ms):
    """What CAN be measured about N = 2 / 4 / 8 on one GPU: one rank's step of the row-sharded loop -- phase A,
    pack, exchange, phase B through nmfx_mur_run_sharded, RCCL behind the C ABI with a world of ONE (the collective is a local copy)
    -- at the per-rank shapes of configs 2 and 5, and a stated model of the all-reduce that is missing from it, so that the first
    SCALE record has something to be checked against.  base_ms: measured single-GPU ms per iteration of the two configs."""
    from nmf_amd import dist as nd
    NEVER = 10 ** 12
    # All-reduce model (an ASSUMPTION until a multi-GPU node has run it; MI355X_MICROARCH: 7 xGMI links x ~153 GB/s per GPU, a ring
    # is bound by one link): t = alpha(N) + 2 (N - 1) / N * bytes / beta, alpha = launch + 2 (N - 1) hops of ~2.5 us, beta = 0.8 x 153 GB/s
    beta = 0.8 * 153e9
    alpha = {2: 12e-6, 4: 22e-6, 8: 42e-6}
    out = {"all_reduce_model": {"formula": "t = alpha(N) + 2 (N - 1) / N * bytes / beta", "alpha_us": {str(q): a * 1e6 for q, a in alpha.items()},
                                "beta_gbs": beta / 1e9, "status": "assumed (ring over one xGMI link per direction); not hidden behind compute in the prediction"},
           "per_rank_step": "measured: nmfx_mur_run_sharded with a world of one on RCCL (the N > 1 code path; its collective moves nothing)",
           "configs": {}}
    for cfg, (m, n, k, steps) in (("cfg2", (16384, 8192, 64, 40)), ("cfg5", (131072, 16384, 128, 6))):
        rows_out = []
        for world in (2, 4, 8):
            rows = m // world
            rs = np.random.RandomState(0)
            w0, h0 = np.abs(rs.randn(rows, k)), np.abs(rs.randn(k, n))
            try:
                shard = nd.NativeShard(None, k, w0, h0, dev.index or 0, shape=(rows, n),
                                       fill=lambda e: device_planted(e, torch, m, n, k, 0, dev, rows=(0, rows)))
            except Exception as e:  # noqa: BLE001
                rows_out.append({"n_gpus": world, "error": f"{type(e).__name__}: {e}"})
                continue
            try:
                comm = nd.NativeComm(shard, 0, 1, shard.eng.comm_unique_id())
                shard.negotiate(comm)
                eng = shard.eng
                t_w = time.perf_counter()                  # clocks, pools, lazy allocations: at least 0.15 s of work (protocol note)
                while True:
                    eng.mur_run_sharded(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 0, 2 * steps)
                    eng.synchronize()
                    if time.perf_counter() - t_w > 0.15:
                        break
                eng.set_factors(w0, h0)
                eng.mur_run_sharded(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 0, 3)
                eng.synchronize()
                # (as many steps as keep the timed region at the N = 2 shard's length: with 6 steps of a 0.65 ms N = 8 step the one call's fixed
                #  cost and the ramp of the first launches read 700 us where tools/lab/shard_step_probe.py reads 623, r5)
                nsteps = steps * world // 2
                t0 = time.perf_counter()
                eng.mur_run_sharded(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 3, nsteps)
                eng.synchronize()
                step = (time.perf_counter() - t0) / nsteps
                _, _, n_obj = eng.state()
                obj = eng.objectives(0, n_obj)
                assert np.all(np.isfinite(obj)) and obj[-1] < obj[0], f"scaling_model {cfg} N={world}: bad objective history"
                n32 = shard.xf32.count
                nbytes = 4.0 * n32 if shard.merge_objective() else 4.0 * n32 + 64
                t_ar = alpha[world] + 2.0 * (world - 1) / world * nbytes / beta
                pred = step + t_ar
                rows_out.append({"n_gpus": world, "rows_per_rank": rows, "measured_step_us_world_of_one": step * 1e6,
                                 "all_reduce_bytes": nbytes, "collectives_per_iteration": 1 if shard.merge_objective() else 2,
                                 "model_all_reduce_us": t_ar * 1e6, "predicted_iter_per_s": 1.0 / pred,
                                 "predicted_speedup_vs_1_gpu": (base_ms[cfg] * 1e-3 / pred) if base_ms.get(cfg) else None,
                                 "upper_bound_speedup_without_exchange": (base_ms[cfg] * 1e-3 / step) if base_ms.get(cfg) else None})
                comm.close()
            except Exception as e:  # noqa: BLE001
                rows_out.append({"n_gpus": world, "rows_per_rank": rows, "error": f"{type(e).__name__}: {e}"})
            finally:
                shard.eng.close()
                del shard
                torch.cuda.empty_cache()
        out["configs"][cfg] = {"shape": [m, n, k], "ms_per_iteration_on_1_gpu": base_ms.get(cfg), "ranks": rows_out}
    return out


def other_config(torch, dev, name, workload, m, n, k, queue, steps, warmup, init, flops, nbytes, admm_iter=0, repeat_dist=None,
                 precision=None, bound="hbm", check_f64=False, kl_streams=None):
    """One of BASELINE.json's non-headline single-GPU configs: iterations/s over `steps` steps after `warmup`,
    per-kernel device times (HIP events on the engine's stream, separate pass), the algorithmic work per
    iteration and the dominant kernel against the HBM roofline."""
    from nmf_amd.engine import Engine
    from nmf_amd import utils
    t_all = time.perf_counter()
    eng = Engine(m, n, k, device=dev.index or 0)
    try:
        if precision:
            eng.set_precision(precision)
        device_planted(eng, torch, m, n, min(k, 128), 0, dev)
        rs = np.random.RandomState(0)
        if init == "randn":                      # nmf/mur.py:108-109
            w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
        elif init == "randn_small":              # |randn| scaled so that W H starts at the data's scale (from |randn| itself AO-ADMM's first
            w0, h0 = 0.05 * np.abs(rs.randn(m, k)), 0.05 * np.abs(rs.randn(k, n))      # Gram system at k = 256 is not positive definite: LinAlgError)
        elif init == "rand":                     # nmf/anls.py:104-105
            w0, h0 = rs.rand(m, k), rs.rand(k, n)
        elif init == "rand_kl":                  # a start of the data's scale for the KL-loss ADMM variants (W H = O(1) everywhere: no 0 / 0 quotients)
            w0, h0 = rs.rand(m, k) + 0.01, rs.rand(k, n) / k + 0.01
        else:                                    # NNDSVD 'zero' (nmf/utils.py:36-93) from the device's singular triplets
            class _Shape:
                shape = (m, n)
            u, sv, vt, _, resid = eng.topk_svd(k)
            assert resid <= 1e-9, resid
            w0, h0 = utils._nndsvd_from_triplets(_Shape, u, sv, vt, k, "zero")
        # untimed rehearsal of the whole sequence first: the first deep launch queue of a process makes the HIP runtime grow
        # its signal / kernel-argument pools, a one-time host stall of tens of milliseconds that would otherwise land in the
        # timed region of whichever config runs first; then back to the initial factors
        eng.set_factors(w0, h0)
        queue(eng, 0, warmup + steps)
        eng.synchronize()
        eng.set_factors(w0, h0)
        queue(eng, 0, warmup)
        eng.synchronize()
        t0 = time.perf_counter()
        queue(eng, warmup, steps)
        t_queued = time.perf_counter() - t0
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        _, _, n_obj = eng.state()
        obj = eng.objectives(0, n_obj)
        assert np.all(np.isfinite(obj)) and obj[-1] < obj[0], f"{name}: bad objective history"
        psteps = 4
        eng.profile_enable(True)
        eng.profile_reset()
        queue(eng, warmup + steps, psteps)
        eng.synchronize()
        prof = {}
        for kn in ALL_KERNELS:
            ms, cnt = eng.profile_get(kn)
            if cnt:
                prof[kn] = {"us_per_launch": round(ms / cnt * 1e3, 2), "launches_per_iter": cnt / psteps}
        eng.profile_enable(False)
        check = None
        if check_f64:
            # the matrix in its stated size CHECKED, not only timed (MUR-eu only): the objective the device
            # recorded for the final pair against nmfx_objective_f64 (product and sum in float64 on the device) of that pair, the
            # whole history decreasing, the factors non-negative and finite (nmf/mur.py:119-128 keeps all three)
            done = warmup + steps + psteps
            eng.mur_finish(0, 10 ** 12, 1e-5, 1e-5, done)
            _, _, n_all = eng.state()
            hist = eng.objectives(0, n_all)
            f64 = eng.objective_f64()
            w_f, h_f = eng.get_factors()
            check = {"iterations": int(done), "objective_recorded": float(hist[-1]), "objective_f64": float(f64),
                     "objective_f64_rel_diff": float(abs(hist[-1] - f64) / abs(f64)),
                     "history_decreasing": bool(len(hist) == done + 1 and np.all(np.diff(hist) < 0)),
                     "factors_nonnegative_finite": bool(np.isfinite(w_f).all() and np.isfinite(h_f).all() and w_f.min() >= 0 and h_f.min() >= 0)}
            del w_f, h_f
            if not os.environ.get("NMFX_BENCH_NOASSERT"):
                assert check["objective_f64_rel_diff"] < 1e-5 and check["history_decreasing"] and check["factors_nonnegative_finite"], f"{name}: {check}"
        inner = paths = None
        if admm_iter:
            inner = (eng.inner_counts(0, warmup + steps) & 0xFFFF).mean(axis=0).tolist()
            flops, nbytes = flops(inner), nbytes(inner)
            paths = eng.inner_paths()
        dom = max((kn for kn in prof if kn in V_SIZED), key=lambda kn: prof[kn]["us_per_launch"])

        def kl_streams_of(kn):          # (ADMM's fused auxiliaries launch always stores S: kl_streams = 4)
            return (kl_streams if kl_streams and kn == "kl_vaux_fused" else KL_STREAMS.get(kn, 1))
        if repeat_dist is not None and dom in ("wphase", "hphase"):      # MUR: the same launch back to back (see main())
            prof[dom]["us_per_launch_with_event_per_launch"] = prof[dom]["us_per_launch"]
            prof[dom]["us_per_launch"] = round(eng.profile_repeat(dom, 20, repeat_dist) * 1e3, 2)
        dsec = prof[dom]["us_per_launch"] * 1e-6
        return {"config": name, "workload": workload, "iter_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "steps": steps,
                "warmup": warmup,
                # (nmfx_get_precision speaks of the tuned k <= 128 kernels; beyond 128 components the composed path runs its V-sized products
                #  in split bf16 unless NMFX_PRECISION=f32 / set_precision('f32'), kernels_generic.hip gxb_on)
                "precision": (eng.precision() if k <= 128 else
                              "f32" if (precision == "f32" or os.environ.get("NMFX_PRECISION", "") in ("f32", "fp32")) else "bf16"),
                "host_ms_to_queue_all_steps": t_queued * 1e3,
                "algorithmic_gflop_per_iter": flops / 1e9, "algorithmic_gbytes_per_iter": nbytes / 1e9,
                "tflops": flops / dt / 1e12, "hbm_gbs": nbytes / dt / 1e9, "frac_of_hbm_peak": nbytes / dt / 1e9 / PEAK_HBM_GBS,
                "dominant_kernel": ({"name": dom, "us_per_launch": prof[dom]["us_per_launch"],
                                     # (the KL auxiliaries' update reads V and dual_v and writes dual_v -- and S, unless the launch forms the next product itself)
                                     "algorithmic_bytes_per_launch": m * n * 4.0 * kl_streams_of(dom), "bound": "hbm",
                                     "achieved_gbs": m * n * 4.0 * kl_streams_of(dom) / dsec / 1e9,
                                     "frac": m * n * 4.0 * kl_streams_of(dom) / dsec / 1e9 / PEAK_HBM_GBS}
                                    if bound == "hbm" else
                                    # split bf16: every algorithmic product is three bf16 MFMA terms (hi hi + lo hi + hi lo)
                                    {"name": dom, "us_per_launch": prof[dom]["us_per_launch"], "bound": "mfma (split bf16: 3 executed terms per product)",
                                     "algorithmic_tflops": 2.0 * m * n * (-(-k // 128) * 128) / dsec / 1e12,
                                     "executed_tflops": 6.0 * m * n * (-(-k // 128) * 128) / dsec / 1e12,
                                     "peak_tflops": PEAK_BF16_MFMA_TFLOPS,
                                     "frac": 6.0 * m * n * (-(-k // 128) * 128) / dsec / 1e12 / PEAK_BF16_MFMA_TFLOPS}
                                    if bound == "mfma_bf16x3" else
                                    {"name": dom, "us_per_launch": prof[dom]["us_per_launch"], "bound": "mfma (f32 inputs)",
                                     "tflops": 2.0 * m * n * (-(-k // 128) * 128 if k > 128 else k) / dsec / 1e12,
                                     "peak_tflops": PEAK_F32_MFMA_TFLOPS,
                                     "frac": 2.0 * m * n * (-(-k // 128) * 128 if k > 128 else k) / dsec / 1e12 / PEAK_F32_MFMA_TFLOPS}),
                "mean_inner_rounds_h_w": inner, "inner_first_leg_stood_cut_continued_both": paths, "objective_first_last": [float(obj[0]), float(obj[-1])],
                "kernels": prof, "data": "synthetic, drawn on the device (torch generator, seed 0)",
                **({"full_matrix_check": check, "objective_f64_rel_diff": check["objective_f64_rel_diff"]} if check else {}),
                "wall_s_incl_setup": round(time.perf_counter() - t_all, 1)}
    finally:
        eng.close()


def other_configs(torch, dev, only=None):
    """Configs 3, 4 and 5-on-one-GPU of BASELINE.json, each a few seconds (the driver-visible numbers the
    round-1 review asked for).  A failure in one of them is reported in its slot, never hidden."""
    NEVER = 10 ** 12
    T = 10
    out = []
    specs = [
        dict(name="cfg3", workload="AO-ADMM Euclidean, reg_w = reg_h = (0.1, 'l1n'), V=16384x8192 f32, k=128, admm_iter=10, "
                                   "NNDSVD-zero start from the device SVD",
             m=16384, n=8192, k=128, steps=20, warmup=3, init="nndsvd", admm_iter=T,
             queue=lambda e, f, c: e.aoadmm_run(0, 1, 0.1, 1, 0.1, T, NEVER, 1e-3, 1e-3, f, c),
             flops=lambda t: 4.0 * 16384 * 8192 * 128 + 2.0 * 128 * 128 * (16384 + 8192) + 2.0 * 128 * 128 * (t[0] * 8192 + t[1] * 16384)
             + 2.0 * 128 ** 3 / 3,
             nbytes=lambda t: 2.0 * 16384 * 8192 * 4 + 8.0 * 128 * 4 * (t[0] * 8192 + t[1] * 16384)),
        dict(name="cfg4", workload="MUR KL-divergence, V=32768x16384 f32, k=64, |randn| start, objective every iteration",
             m=32768, n=16384, k=64, steps=10, warmup=2, init="randn", repeat_dist=1,
             queue=lambda e, f, c: e.mur_run(1, 0.0, 0.0, NEVER, 1e-5, 1e-5, f, c),
             flops=8.0 * 32768 * 16384 * 64, nbytes=2.0 * 32768 * 16384 * 4),
        dict(name="cfg5_on_1_gpu", workload="MUR Euclidean, V=131072x16384 f32 (8 GiB), k=128 on ONE GPU: the strong-scaling base of "
                                            "the 8-GPU config",
             m=131072, n=16384, k=128, steps=5, warmup=2, init="randn", repeat_dist=0, check_f64=True,
             queue=lambda e, f, c: e.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, f, c),
             flops=4.0 * 131072 * 16384 * 128 + 4.0 * 128 * 128 * (131072 + 16384),
             nbytes=2.0 * 131072 * 16384 * 4 + 3.0 * (131072 + 16384) * 128 * 4),
        # not one of BASELINE.json's configs: the fourth solver of the API on the headline shape (anls.py:112-126; exact NNLS per
        # row of W / column of H).  Algorithmic work: the two V-sized products + the objective pass over V
        dict(name="anls_on_cfg2_shape", workload="ANLS Euclidean (exact NNLS, lambda = 0), V=16384x8192 f32, k=64, uniform random start, "
                                                "objective every iteration",
             m=16384, n=8192, k=64, steps=10, warmup=4, init="rand",
             queue=lambda e, f, c: e.anls_run(0.0, 0.0, NEVER, 1e-3, 1e-3, f, c),
             flops=6.0 * 16384 * 8192 * 64, nbytes=3.0 * 16384 * 8192 * 4),
        # the headline config in the exact-f32 arithmetic (NMFX_PRECISION=f32: f32-input MFMA, bit-exact FMA chains): MFMA-bound
        dict(name="cfg2_exact_f32", workload="MUR Euclidean, V=16384x8192 f32, k=64 with the exact-f32 products (NMFX_PRECISION=f32)",
             m=16384, n=8192, k=64, steps=20, warmup=3, init="randn", precision="f32", bound="mfma",
             queue=lambda e, f, c: e.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, f, c),
             flops=4.0 * 16384 * 8192 * 64 + 4.0 * 64 * 64 * (16384 + 8192), nbytes=2.0 * 16384 * 8192 * 4 + 3.0 * (16384 + 8192) * 64 * 4),
        # beyond 128 components (nmf/nmf.py:32-35 takes any `factors`): the iteration composed from split-bf16 NT product kernels over
        # operand planes (kernels_generic.hip: gxt_* on tiled planes filled by LDS-DMA for the long contractions, gxb_* for the short
        # ones; r3 -- the exact-f32 form of the same composition is the leg after it)
        dict(name="mur_k256_on_cfg2_shape", workload="MUR Euclidean, V=16384x8192 f32, k=256 (composed path for k > 128, split-bf16 products)",
             m=16384, n=8192, k=256, steps=10, warmup=2, init="randn", bound="mfma_bf16x3",
             queue=lambda e, f, c: e.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, f, c),
             flops=4.0 * 16384 * 8192 * 256 + 4.0 * 256 * 256 * (16384 + 8192), nbytes=2.0 * 16384 * 8192 * 4 + 3.0 * (16384 + 8192) * 256 * 4),
        dict(name="mur_k256_exact_f32", workload="MUR Euclidean, V=16384x8192 f32, k=256 with the exact-f32 product kernel (NMFX_PRECISION=f32)",
             m=16384, n=8192, k=256, steps=5, warmup=1, init="randn", bound="mfma", precision="f32",
             queue=lambda e, f, c: e.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, f, c),
             flops=4.0 * 16384 * 8192 * 256 + 4.0 * 256 * 256 * (16384 + 8192), nbytes=2.0 * 16384 * 8192 * 4 + 3.0 * (16384 + 8192) * 256 * 4),
        dict(name="mur_kl_k256_on_cfg2_shape", workload="MUR KL-divergence, V=16384x8192 f32, k=256 (composed path, split-bf16 products, quotient as bf16 planes)",
             m=16384, n=8192, k=256, steps=8, warmup=2, init="randn", bound="mfma_bf16x3",
             queue=lambda e, f, c: e.mur_run(1, 0.0, 0.0, NEVER, 1e-5, 1e-5, f, c),
             flops=8.0 * 16384 * 8192 * 256, nbytes=4.0 * 16384 * 8192 * 4),
        dict(name="aoadmm_k256_on_cfg2_shape", workload="AO-ADMM Euclidean, reg_w = reg_h = (0.1, 'l1n'), V=16384x8192 f32, k=256, admm_iter=10, "
                                                        "0.05 |randn| start (composed path: split-bf16 V-sized products, Gram systems by blocks of 128)",
             m=16384, n=8192, k=256, steps=6, warmup=2, init="randn_small", admm_iter=T, bound="mfma_bf16x3",
             queue=lambda e, f, c: e.aoadmm_run(0, 1, 0.1, 1, 0.1, T, NEVER, 1e-3, 1e-3, f, c),
             flops=lambda t: 4.0 * 16384 * 8192 * 256 + 2.0 * 256 * 256 * (16384 + 8192) + 2.0 * 256 * 256 * (t[0] * 8192 + t[1] * 16384)
             + 2.0 * 256 ** 3 / 3,
             nbytes=lambda t: 2.0 * 16384 * 8192 * 4 + 8.0 * 256 * 4 * (t[0] * 8192 + t[1] * 16384)),
        # the KL-loss variants (nmf/ao_admm.py:71-101, nmf/admm.py:303-315) on the config-3 shape: two V-sized auxiliaries (v_aux, dual_v)
        # updated in EVERY inner round; split bf16 (r4: the auxiliaries as a mode of the product kernel) and the exact-f32 path beside it.
        # Algorithmic work per inner round: the product of the right-hand side and the product inside the auxiliaries' update;
        # V read, dual_v read + written (r5: S = v_aux + dual_v stays in registers between the rounds of a sub-problem -- per sub-problem it is
        # read once by the first round's product and written once by the last round: 3 T + 2 V-sized streams; the exact-f32 leg moves 5 T)
        dict(name="aoadmm_kl_on_cfg3_shape", workload="AO-ADMM KL loss, reg_w = reg_h = (0, 'nn'), V=16384x8192 f32, k=128, admm_iter=10, uniform "
                                                      "random start of the data's scale (split bf16: auxiliaries as a mode of the product kernel)",
             m=16384, n=8192, k=128, steps=3, warmup=1, init="rand_kl", admm_iter=T,
             queue=lambda e, f, c: e.aoadmm_run(1, 0, 0.0, 0, 0.0, T, NEVER, 1e-3, 1e-3, f, c),
             flops=lambda t: 4.0 * 16384 * 8192 * 128 * (t[0] + t[1]), nbytes=lambda t: (3.0 * (t[0] + t[1]) + 4.0) * 16384 * 8192 * 4),
        dict(name="aoadmm_kl_exact_f32", workload="AO-ADMM KL loss on the config-3 shape with the exact-f32 kernels (NMFX_PRECISION=f32)",
             m=16384, n=8192, k=128, steps=2, warmup=1, init="rand_kl", admm_iter=T, precision="f32",
             queue=lambda e, f, c: e.aoadmm_run(1, 0, 0.0, 0, 0.0, T, NEVER, 1e-3, 1e-3, f, c),
             flops=lambda t: 4.0 * 16384 * 8192 * 128 * (t[0] + t[1]), nbytes=lambda t: 5.0 * 16384 * 8192 * 4 * (t[0] + t[1])),
        dict(name="admm_kl_on_cfg3_shape", workload="ADMM KL loss, rho = 1, reg_w = reg_h = (0, 'nn'), V=16384x8192 f32, k=128 (split bf16)",
             m=16384, n=8192, k=128, steps=8, warmup=2, init="rand_kl",
             queue=lambda e, f, c: e.admm_run(1, 1.0, 0, 0.0, 0, 0.0, NEVER, 1e-3, 1e-3, f, c),
             # (V-sized streams per iteration: the auxiliaries read V and dual_v and write dual_v and S -- their launch also forms the next
             #  iteration's first product from S in registers (r5) --, the second product reads S, the objective pass reads V: 6; r4's sequence: 7 + a transpose)
             flops=8.0 * 16384 * 8192 * 128, nbytes=6.0 * 16384 * 8192 * 4, kl_streams=4),
        # ADMM (nmf/admm.py:292-334) beyond 128 components: V-sized products with FOUR split-bf16 terms (they are fed back through the
        # unshifted-rho Gram systems), the objective of (w, h) by one more product; algorithmic work: three V-sized products
        dict(name="admm_k256_on_cfg2_shape", workload="ADMM Euclidean, rho = 1, reg_w = (0, 'nn'), reg_h = (0.1, 'l1n'), V=16384x8192 f32, k=256, "
                                                      "0.05 |randn| start (composed path, split-bf16 products with four terms)",
             m=16384, n=8192, k=256, steps=8, warmup=3, init="randn_small", bound="mfma_bf16x3",
             queue=lambda e, f, c: e.admm_run(0, 1.0, 0, 0.0, 1, 0.1, NEVER, 1e-5, 1e-5, f, c),
             flops=6.0 * 16384 * 8192 * 256 + 4.0 * 256 * 256 * (16384 + 8192) + 4.0 * 256 ** 3 / 3,
             nbytes=3.0 * 16384 * 8192 * 4 + 10.0 * (16384 + 8192) * 256 * 4),
    ]
    for sp in specs:
        if only and sp["name"] not in only:
            continue
        try:
            out.append(other_config(torch, dev, **sp))
        except Exception as e:  # noqa: BLE001
            out.append({"config": sp["name"], "error": f"{type(e).__name__}: {e}"})
        torch.cuda.empty_cache()
    if not only or "pair_on_cfg2_shape" in only:
        try:
            out.append(pair_config(torch, dev))
        except Exception as e:  # noqa: BLE001
            out.append({"config": "pair_on_cfg2_shape", "error": f"{type(e).__name__}: {e}"})
        torch.cuda.empty_cache()
    return out


def pair_config(torch, dev, steps=40, warmup=5):
    """SURVEY 8 f4, the parameter grid of the reference's author (nmf/nmf_old.py:52-66): two MUR-Euclidean problems (k = 64 each,
    different lambda and start) on the config-2 matrix, as ONE pass over V per half-iteration (nmfx_mur_pair_run) against the same
    two problems one after the other on a k = 64 engine."""
    from nmf_amd.engine import Engine
    m, n, k = M, N, K
    NEVER = 10 ** 12
    rs = np.random.RandomState(0)
    starts = [(np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))) for _ in range(2)]
    lws, lhs = [0.0, 0.1], [0.0, 0.05]
    single = []
    with Engine(m, n, k, device=dev.index or 0) as e:
        device_planted(e, torch, m, n, k, 0, dev)
        for (w0, h0), lw, lh in zip(starts, lws, lhs):
            e.set_factors(w0, h0)
            e.mur_run(0, lw, lh, NEVER, 1e-5, 1e-5, 0, 200)          # clocks + pools
            e.synchronize()
            e.set_factors(w0, h0)
            e.mur_run(0, lw, lh, NEVER, 1e-5, 1e-5, 0, warmup)
            e.synchronize()
            t0 = time.perf_counter()
            e.mur_run(0, lw, lh, NEVER, 1e-5, 1e-5, warmup, steps)
            e.synchronize()
            single.append((time.perf_counter() - t0) / steps)
    with Engine(m, n, 128, device=dev.index or 0) as e:
        device_planted(e, torch, m, n, k, 0, dev)
        w0 = np.concatenate([starts[0][0], starts[1][0]], axis=1)
        h0 = np.concatenate([starts[0][1], starts[1][1]], axis=0)
        e.set_factors(w0, h0)
        e.mur_pair_run(lws, lhs, NEVER, 1e-5, 1e-5, 0, 200)
        e.synchronize()
        e.set_factors(w0, h0)
        e.mur_pair_run(lws, lhs, NEVER, 1e-5, 1e-5, 0, warmup)
        e.synchronize()
        t0 = time.perf_counter()
        e.mur_pair_run(lws, lhs, NEVER, 1e-5, 1e-5, warmup, steps)
        e.synchronize()
        pair = (time.perf_counter() - t0) / steps
        objs = [e.pair_objectives(p, 0, warmup + steps) for p in (0, 1)]
        assert all(np.all(np.isfinite(o)) and o[-1] < o[0] for o in objs), "pair: bad objective history"
        e.profile_enable(True)
        e.profile_reset()
        e.mur_pair_run(lws, lhs, NEVER, 1e-5, 1e-5, warmup + steps, 4)
        e.synchronize()
        prof = {}
        for kn in ALL_KERNELS:
            ms, cnt = e.profile_get(kn)
            if cnt:
                prof[kn] = {"us_per_launch": round(ms / cnt * 1e3, 2), "launches_per_iter": cnt / 4}
    return {"config": "pair_on_cfg2_shape",
            "workload": f"two MUR-Euclidean problems (k = {k} each, lambda_w = {lws}, lambda_h = {lhs}, separate |randn| starts) on V={m}x{n}: "
                        "one pass over V per half-iteration for both (k = 128 layouts) vs one after the other",
            "ms_per_iteration_of_the_pair": pair * 1e3, "ms_per_iteration_single": [s_ * 1e3 for s_ in single],
            "problem_iterations_per_s_paired": 2.0 / pair, "problem_iterations_per_s_sequential": 2.0 / sum(single),
            "speedup_per_problem": sum(single) / pair, "kernels": prof,
            "data": "synthetic, drawn on the device (torch generator, seed 0)"}


HEADLINE_LIMIT = 4096             # bytes: the ONE stdout line must fit, whole, in the driver's stdout tail


def _r(x, sig=5):
    """A float rounded to `sig` significant digits (None and non-floats pass through)."""
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")) and x != 0.0:
        from math import floor, log10
        return round(x, sig - 1 - int(floor(log10(abs(x)))))
    return x


def headline(detail):
    """The compact stdout line made from the full record `detail` (which goes to bench_detail.json and stderr): the contract's keys,
    `roofline`, `cpu_baseline`, the full-size parity figures, the exact-f32 leg, ONE pair of numbers per other BASELINE config and the
    scaling model's N = 8 predictions as bare numbers.  No prose beyond the workload names."""
    line = {k_: detail.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                         "scaling", "vs_baseline", "dtype", "data", "config")}
    roof = detail.get("roofline")
    if roof:
        line["roofline"] = {k_: _r(roof.get(k_)) for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                           "algorithmic_bytes_per_launch", "us_per_launch") if k_ in roof}
    else:
        line["roofline"] = None
    cpu = detail.get("cpu_baseline")
    line["cpu_baseline"] = ({k_: _r(cpu.get(k_)) for k_ in ("value", "unit", "cores", "kind", "sample")} if cpu else None)
    par = detail.get("parity")
    line["parity"] = ({k_: _r(par.get(k_), 3) for k_ in ("wh_rel_err", "obj_max_rel_diff", "iterations")} if par else None)
    it = detail.get("iteration")
    if it:
        line["iteration_frac_of_hbm_peak"] = _r(it.get("frac_of_hbm_peak"), 3)
    ttt = detail.get("time_to_tol")
    if ttt:
        line["time_to_tol"] = [{"tol": t["tol1"], "converged": t["converged"], "iterations": t["iterations"], "seconds": _r(t["seconds"], 4),
                                "oracle_agrees": (t.get("oracle_stop_check") or {}).get("agree")} for t in ttt]
    for o in detail.get("other_configs") or []:
        name = o.get("config")
        if name in ("cfg3", "cfg4", "cfg5_on_1_gpu", "cfg5"):
            if "error" in o:
                line[name] = {"error": str(o["error"])[:120]}
                continue
            slot = {"iter_per_s": _r(o.get("iter_per_s")), "frac_of_hbm_peak": _r(o.get("frac_of_hbm_peak"), 3)}
            dom = o.get("dominant_kernel")
            if dom:
                slot["dominant_kernel"] = dom.get("name")
                slot["dominant_frac"] = _r(dom.get("frac"), 3)
            for k_ in ("n_gpus", "loop", "objective_f64_rel_diff", "exchange"):
                if k_ in o:
                    slot[k_] = _r(o[k_], 3)
            line[name] = slot
        elif name == "cfg2_exact_f32" and "iter_per_s" in o:
            line["exact_f32_iter_per_s"] = _r(o["iter_per_s"])
            line["exact_f32_frac_of_f32_mfma_peak"] = _r((o.get("dominant_kernel") or {}).get("frac"), 3)
    sm = detail.get("scaling_model")
    if sm and "configs" in sm:
        pred = {}
        for cfg, body in sm["configs"].items():
            for row in body.get("ranks", []):
                if "predicted_speedup_vs_1_gpu" in row and row["predicted_speedup_vs_1_gpu"] is not None:
                    pred[f"{cfg}_n{row['n_gpus']}"] = _r(row["predicted_speedup_vs_1_gpu"], 3)
        line["scaling_model_predicted_speedup"] = pred or None
    line["detail"] = "bench_detail.json (cwd; also on stderr)"
    # the limit is a promise: shed the optional slots, last first, rather than break it
    for drop in ("time_to_tol", "scaling_model_predicted_speedup", "iteration_frac_of_hbm_peak", "cfg5", "cfg3", "cfg4", "cfg5_on_1_gpu"):
        if len(json.dumps(line)) < HEADLINE_LIMIT:
            break
        line.pop(drop, None)
    return line


def emit(detail, json_fd):
    """Full record -> bench_detail.json (cwd, and gpurun_out/ when that exists, so that it is pulled back) and stderr; compact line -> stdout."""
    text = json.dumps(detail)
    # (NMFX_BENCH_DETAIL=<path>: that file only -- tests)
    targets = [os.environ["NMFX_BENCH_DETAIL"]] if os.environ.get("NMFX_BENCH_DETAIL") else \
        [os.path.join(d, "bench_detail.json") for d in (os.getcwd(), os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    for path in targets:
        try:
            with open(path, "w") as f:
                f.write(text + "\n")
        except OSError as e:
            sys.stderr.write(f"bench.py: could not write {path}: {e}\n")
    sys.stderr.write("bench_detail: " + text + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(headline(detail)) + "\n").encode())


class Deadline:
    """The config-5 leg of an N > 1 run times forms of the exchange that no multi-GPU box has run before this line is printed (RCCL
    behind the C ABI with more than one rank, reduce-scatter + all-gather).  A collective that never completes on some rank would take
    the WHOLE line with it -- the strong-scaling figure of config 2 included -- so the leg runs under a deadline: when it expires,
    rank 0 prints the line with the forms that HAVE run (naming the one that did not come back) and every rank leaves with exit code
    0.  A watchdog thread, because a rank stuck inside a C call never returns to the interpreter for a signal handler."""

    def __init__(self, seconds, rank, on_expiry):
        import threading
        self.done = threading.Event()
        self.seconds, self.rank, self.on_expiry = seconds, rank, on_expiry
        self.thread = threading.Thread(target=self._watch, daemon=True)

    def start(self):
        self.thread.start()
        return self

    def cancel(self):
        self.done.set()

    def _watch(self):
        if self.done.wait(self.seconds + (0.0 if self.rank == 0 else 5.0)):      # (rank 0 prints first)
            return
        try:
            if self.rank == 0:
                self.on_expiry()
        finally:
            sys.stderr.write(f"bench.py: rank {self.rank}: the config-5 leg did not come back within {self.seconds:.0f} s -- leaving\n")
            sys.stderr.flush()
            os._exit(0)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT torchrun: become the launcher.  Nothing has touched the GPU yet (torch is
    not even imported), so starting `python -m torch.distributed.run ... bench.py <same arguments>` as a child process is safe; its
    rank 0 writes the one JSON line to the stdout it inherits from us, and we leave with its exit code.  The rendezvous port is
    probed here and bound by the child a moment later: if that one step loses the port (EADDRINUSE before any rank has started),
    the launch -- not the benchmark -- is repeated with another port."""
    import socket
    import subprocess
    import threading
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = 1
    for attempt in range(3):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.stderr.write(f"bench.py: --gpus {args.gpus} without WORLD_SIZE: launching torch.distributed.run (port {port})\n")
        sys.stderr.flush()
        child = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True, errors="replace")
        seen = []

        def relay():
            for ln in child.stderr:
                seen.append(ln)
                sys.stderr.write(ln)
                sys.stderr.flush()
        th = threading.Thread(target=relay, daemon=True)
        th.start()
        rc = child.wait()
        th.join(timeout=10)
        if rc == 0 or not any("EADDRINUSE" in ln or "address already in use" in ln.lower() for ln in seen):
            break
    raise SystemExit(rc)


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (before anything initialises HIP: the host driver only supports dmabuf IPC)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # the contract is ONE JSON line on stdout: RCCL prints a version banner to stdout when the
    # communicator comes up, so everything but the final line is sent to stderr at the fd level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: running with the world the launcher made\n")
    import torch
    import torch.distributed as dist
    limit_blas_threads(HOST_THREADS)              # (torch's OpenMP pool too, now that it is loaded)
    torch.set_num_threads(HOST_THREADS)
    from nmf_amd.synth import planted_matrix      # (the oracle is imported by the cpu_baseline leg only)
    from nmf_amd import dist as nd

    m, n, k = args.m, args.n, args.k
    if os.environ.get("NMFX_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = 0            # rehearsal: all ranks share GPU 0
    torch.cuda.set_device(local_rank)
    # NMFX_BENCH_FORCE_SHARDED=1: take the N > 1 code path (phase A -> all-reduce -> phase B driven
    # from Python) with a world of one, to measure its host-side cost on a single GPU
    sharded = world > 1 or bool(os.environ.get("NMFX_BENCH_FORCE_SHARDED"))
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" (= RCCL over xGMI) is the real thing; NMFX_BENCH_BACKEND=gloo stages the
        # exchange through the host so that the N > 1 code path can be rehearsed with several
        # ranks on ONE GPU (RCCL refuses duplicate devices)
        backend = os.environ.get("NMFX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    comm, loop = None, "library"

    r0, r1 = nd.row_range(m, rank, world)
    v_local = planted_matrix(m, n, k, seed=0, dtype=np.float32, rows=(r0, r1))
    rs = np.random.RandomState(0)
    w0 = np.abs(rs.randn(m, k))[r0:r1]
    h0 = np.abs(rs.randn(k, n))

    NEVER = 10 ** 12          # min_iter: the stop rule is evaluated but cannot fire
    if sharded:
        shard, comm, loop = make_sharded(torch, dist, nd, rank, world, local_rank, lambda cls: cls(v_local, k, w0, h0, local_rank))

        run = nd.Runner(shard, comm, 0, 0.0, 0.0, NEVER, 1e-5, 1e-5,
                        args.warmup + args.steps + args.profile_steps + 8)
        eng = shard.eng
    else:
        from nmf_amd.engine import Engine
        eng = Engine(m, n, k, device=local_rank)
        eng.upload_v(v_local)
        eng.set_factors(w0, h0)

        def run(first, count):
            eng.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, first, count)

    precision = eng.precision()

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if sharded:
            # the barrier of the protocol, through the communicator the data path uses: torch.distributed's own one has been idle
            # for the whole timed region of the native loop and takes ~0.5 ms to answer (tools/lab/native_batch_probe.py)
            if isinstance(comm, nd.NativeComm):
                comm.barrier()
            else:
                dist.barrier()
            torch.cuda.synchronize()

    if args.preheat > 0 and not args.pmc_child:
        # device warm-up, not part of the measured protocol: same kernels, then back to the initial factors
        (run.eager if sharded else run)(0, args.preheat)
        fence()
        eng.set_factors(w0, h0)
    run(0, args.warmup)
    if sharded:
        run.ensure_graph(args.warmup)      # a capture still pending must not land in the timed region
    fence()
    if sharded and run.mode in ("hipgraph", "native-hipgraph") and args.warmup >= 4:
        # the replayed loop must have produced a sane history; if not, start over with the eager loop
        _, _, n_obj = eng.state()
        hist = eng.objectives(0, n_obj)
        if n_obj != args.warmup or not np.all(np.isfinite(hist)) or not hist[-1] < hist[0]:
            sys.stderr.write(f"rank {rank}: graphed loop gave a bad objective history, falling back to eager\n")
            if run.mode == "native-hipgraph":
                eng.comm_set_graph(False)
                run.mode = "native"
            else:
                run.graph, run.mode, run.want_graph = None, "eager", False
            eng.set_factors(w0, h0)
            run(0, args.warmup)
            fence()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    if sharded:
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    done = args.warmup + args.steps

    # sanity: objective history is finite and decreasing
    _, _, n_obj = eng.state()
    obj = eng.objectives(0, n_obj)
    if not os.environ.get("NMFX_BENCH_NOASSERT"):      # (timing experiments with deliberately wrong kernels)
        assert np.all(np.isfinite(obj)) and obj[-1] < obj[0], "bench run produced a bad objective"

    # profiled pass: per-kernel device time from HIP events on the engine's stream
    roof = None
    prof = {}
    if args.profile_steps > 0 and not args.pmc_child:
        eng.profile_enable(True)
        eng.profile_reset()
        (run.eager if sharded else run)(done, args.profile_steps)   # per-kernel events need eager launches
        fence()
        for name in ("wphase", "hphase", "gram_tn", "gram_nt", "sum_hht", "w_update", "pack", "h_update", "small"):
            ms, cnt = eng.profile_get(name)
            if cnt:
                prof[name] = {"ms_per_launch": ms / cnt, "launches": cnt}
        eng.profile_enable(False)
        # the dominant kernels once more, `reps` launches back to back between ONE pair of HIP events: an event in front of
        # every launch is a command-processor barrier that costs the W phase ~10 us which no real iteration pays (its
        # rocprofv3 kernel-trace average agrees with THIS number, profiles/)
        for name in ("wphase", "hphase"):
            if name in prof:
                prof[name]["ms_per_launch_with_event_per_launch"] = prof[name]["ms_per_launch"]
                prof[name]["ms_per_launch"] = eng.profile_repeat(name, 200)
        fence()
        ml = r1 - r0
        if "wphase" in prof:
            sec = prof["wphase"]["ms_per_launch"] * 1e-3
            flops = 2.0 * ml * n * k
            nbytes = ml * n * 4.0 + 2.0 * ml * k * 4
            if precision == "bf16":
                ach = nbytes / sec / 1e9
                roof = {"kernel": "xyt32_bf16_kernel<true, 3> (W phase: A = V H^T + residual objective)", "bound": "hbm", "achieved": ach,
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                        "algorithmic_bytes_per_launch": nbytes,
                        "algorithmic_tflops": flops / sec / 1e12}
            else:
                ach = flops / sec / 1e12
                roof = {"kernel": "wphase_kernel<64,true,true,false>", "bound": "mfma", "achieved": ach,
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                        "executed_flops_per_launch": 4.0 * ml * n * k,
                        "hbm_gbs": nbytes / sec / 1e9}
        if "hphase" in prof:
            sec = prof["hphase"]["ms_per_launch"] * 1e-3
            prof["hphase"]["tflops"] = 2.0 * ml * n * k / sec / 1e12
            prof["hphase"]["hbm_gbs"] = ml * n * 4.0 / sec / 1e9

    if args.pmc_child:            # inside a rocprofv3 PMC pass: the timed region above is all that is needed
        return

    # time-to-tol (the second half of BASELINE.json's metric): the same problem from the same start,
    # run until the reference's stop rule fires with ITS default tolerances (nmf/mur.py:52-53:
    # min_iter=100, tol1=tol2=1e-5), or the cap.  Factors stay on the device; the stop rule is
    # evaluated on the device every iteration; the host looks at the flag once per 256 iterations.
    ttt = None
    if rank == 0 and world == 1 and args.tol_max_iter > 0:
        ttt = []
        for tol in (1e-5, 1e-2, 1e-3):    # the reference's default (capped), and two looser absolute tolerances
            rule, stop_i, done_t, secs, ref = converge_on_device(eng, w0, h0, tol, args.tol_max_iter)
            ttt.append({"tol1": tol, "tol2": tol, "min_iter": 100, "max_iter": args.tol_max_iter,
                        "converged": bool(rule), "stop_rule": int(rule),
                        "iterations": int(stop_i + 1) if rule else int(done_t), "seconds": secs,
                        "objective": float(ref.history[-1]),
                        "stop_guard": ref.guard, "iterations_refereed_in_f64": ref.walked,
                        "note": "the product's loop: 64 queued iterations per host round trip; near the stop the rule is refereed "
                                "with the float64 objective of the device's iterate (nmfx_objective_f64), one iteration at a time"})
        # the converged leg against the f64 oracle: same stop index, same rule (needs the host copy of V: done in the cpu_baseline leg)
        if not args.no_cpu:
            for leg in ttt:
                # (only where the decrease still changes fast against the transient an oracle restarted from an f32 iterate goes
                # through -- tol >= 1e-2 here; the 1e-3 leg is pinned in DESIGN.md 2 with a 100-iteration lead, 80 s of host time)
                if leg["converged"] and leg["tol2"] >= 1e-2:
                    v_chk = v_local if (r0, r1) == (0, m) else planted_matrix(m, n, k, seed=0, dtype=np.float32)
                    leg["oracle_stop_check"] = oracle_stop_check(eng, v_chk, w0, h0, leg["tol1"], leg["stop_rule"], leg["iterations"] - 1)
                    if not os.environ.get("NMFX_BENCH_NOASSERT"):
                        assert leg["oracle_stop_check"]["agree"], f"time-to-tol: device and oracle stop differently: {leg['oracle_stop_check']}"

    # parity at the full size, GPU leg: the same cpu_iters + 1 iterations the cpu_baseline leg runs, from
    # the same start, in the arithmetic that was timed; compared with the oracle's factors further down
    par_gpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        p_it = args.cpu_iters + 1
        eng.set_factors(w0, h0)
        eng.mur_run(0, 0.0, 0.0, NEVER, 1e-5, 1e-5, 0, p_it)
        eng.mur_finish(0, NEVER, 1e-5, 1e-5, p_it)
        w_g, h_g = eng.get_factors()
        par_gpu = (w_g, h_g, eng.objectives(0, p_it + 1))

    loop_name = (loop + " / " + run.mode) if sharded else "library"

    def build_line(others, smodel, roof, cpu, parity, ttt):
        ms = dt / args.steps * 1e3
        iter_flops = 4.0 * m * n * k + 4.0 * k * k * (m + n)
        iter_bytes = 2.0 * m * n * 4 + 3.0 * (m + n) * k * 4           # V once per phase, W and H read + write
        return {
            "metric": "NMF outer iterations/sec (MUR-eu, V=16384x8192 f32, k=64)",
            "value": args.steps / dt, "unit": "iter/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "preheat_iterations": args.preheat,
            "strong_scaling_quoted_on": ("`value` = config 2 (V=16384x8192, k=64), the fixed matrix row-sharded over n_gpus ranks: a 0.22 ms "
                                         "iteration on one GPU, so its curve is bound by launch + all-reduce latency, not bandwidth.  The "
                                         "bandwidth-bound strong-scaling figure (north_star's >= 6x at 8 GPUs) is config 5 (V=131072x16384, "
                                         "k=128): other_configs[config == 'cfg5'].iter_per_s of the N-GPU line over "
                                         "other_configs[config == 'cfg5_on_1_gpu'] of the N = 1 line (same matrix)"),
            "dtype": "bf16 hi+lo split MFMA (3 terms), f32 accumulate, f64 objective" if precision == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": f"MUR Euclidean, V={m}x{n} float32, k={k}, planted low-rank + 1% noise, "
                                   "|randn| init, objective every iteration",
                       "rows_per_gpu": (m + world - 1) // world, "parallelism": f"row-shard x{world}",
                       "loop": loop_name},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
            "time_to_tol": ttt,
            "iteration": {"algorithmic_gflop": iter_flops / 1e9,
                          "algorithmic_gbytes": iter_bytes / 1e9,
                          "tflops": iter_flops / (dt / args.steps) / 1e12,
                          "hbm_gbs": iter_bytes / (dt / args.steps) / 1e9,
                          "frac_of_hbm_peak": iter_bytes / (dt / args.steps) / 1e9 / PEAK_HBM_GBS / world},
            "kernels": prof,
            "other_configs": others,
            "scaling_model": smodel,
        }

    others = None
    if sharded and not args.no_others and ((m, n, k) == (M, N, K) or os.environ.get("NMFX_BENCH_CFG5_SHAPE")):
        # config 5 proper (the 8-GPU config of BASELINE.json), sharded over this run's ranks: every rank takes part
        eng.close()
        progress = {}

        def expired():
            leg = dict(progress.get("result") or {"config": "cfg5"})
            leg["timed_out_in"] = progress.get("running", "set-up")
            if "iter_per_s" not in leg:
                leg["error"] = f"the config-5 leg did not come back (in {leg['timed_out_in']})"
            emit(build_line([leg], None, roof, None, None, ttt), json_fd)

        guard = Deadline(float(os.environ.get("NMFX_BENCH_CFG5_DEADLINE", "240")), rank, expired).start()
        try:
            others = [cfg5_sharded(torch, dist, nd, rank, world, local_rank, progress=progress)]
        except Exception as e:  # noqa: BLE001  (reported in its slot, never hidden; a rank that failed alone would hang the others'
            others = [{"config": "cfg5", "error": f"{type(e).__name__}: {e}"}]          # collectives -- the deadline above ends that)
        guard.cancel()
    smodel = None
    if rank == 0 and world == 1 and not sharded and not args.no_others and (m, n, k) == (M, N, K):
        eng.close()               # free the HBM: config 5 on one GPU holds three 8 GiB copies of V
        others = other_configs(torch, torch.device(f"cuda:{local_rank}"))
        try:
            c5 = next((o for o in others if o.get("config") == "cfg5_on_1_gpu" and "ms_per_step" in o), None)
            smodel = scaling_model(torch, torch.device(f"cuda:{local_rank}"),
                                   {"cfg2": dt / args.steps * 1e3, "cfg5": c5["ms_per_step"] if c5 else None})
        except Exception as e:  # noqa: BLE001
            smodel = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and roof is not None and not args.no_traffic:
        eng.close()               # free the HBM before the child passes allocate their own
        roof["traffic"] = hbm_traffic("xyt32_bf16_kernel<true" if precision == "bf16" else "wphase_kernel", m, n, k)
        roof["traffic_note"] = ("HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                                "FETCH_SIZE x2 (gfx950 correction)")

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        v_full = v_local if (r0, r1) == (0, m) else planted_matrix(m, n, k, seed=0, dtype=np.float32)
        val, w_r, h_r, obj_r = cpu_baseline(v_full, k, args.cpu_iters)
        parity = parity_block(v_full, par_gpu[0], par_gpu[1], par_gpu[2], w_r, h_r, obj_r)
        if not os.environ.get("NMFX_BENCH_NOASSERT"):
            assert parity["wh_rel_err"] < 1e-4 and parity["obj_max_rel_diff"] < 2e-4, f"full-size parity failed: {parity}"
        cpu = {"value": val, "unit": "iter/s", "cores": HOST_THREADS, "kind": "port",
               "sample": f"full {m}x{n} k={k} shape, {args.cpu_iters} iterations after 1 warm-up, numpy {np.__version__}, "
                         f"BLAS pool of {HOST_THREADS} threads = the container's CPU quota (os.cpu_count() = {os.cpu_count()})"}

    if rank == 0:
        emit(build_line(others, smodel, roof, cpu, parity, ttt), json_fd)
    if sharded:
        try:                      # (the line is out: a peer that has left -- the deadline above -- must not turn the run into a failure)
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"bench.py: rank {rank}: tear-down of the process group failed ({type(e).__name__}: {e})\n")


if __name__ == "__main__":
    main()
