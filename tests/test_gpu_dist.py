"""The row-sharded device path on the GPU box.
 - world 1 over RCCL ("nccl"): exchange buffers owned by torch, engine on
   torch's stream, RCCL all-reduce in place -- must equal the plain engine.
 - world 2 on ONE GPU (two processes, gloo staged through the host, because
   RCCL refuses two ranks on one device): real shards, real kernels."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import spawn_ranks

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _join(backend, rdzv, rank, world):
    """Process group + (shard class, comm factory) for a backend name: "nccl" (RCCL through torch.distributed), "gloo" (two ranks
    on one GPU, staged through the host), "native" (RCCL behind the C ABI -- nmfx_comm_*; torch.distributed, on gloo, only
    carries the 128-byte id)."""
    import torch.distributed as dist
    from nmf_amd import dist as nd
    dist.init_process_group("gloo" if backend == "native" else backend, init_method=rdzv, rank=rank, world_size=world)
    if backend == "native":
        return nd.NativeShard, (lambda shard: nd.NativeComm.create(shard))
    return nd.DeviceShard, (lambda shard: nd.TorchComm(stage_through_host=(backend == "gloo")))


def _case(k=12):
    from oracle import nmf_ref as R
    m, n = 700, 330
    v = R.planted_matrix(m, n, k, seed=21, dtype=np.float32)
    rs = np.random.RandomState(22)
    return m, n, k, v, np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))


def _case_wide(k):
    from oracle import nmf_ref as R
    m, n = 520, 1400                       # padded n = 1408: two column chunks of at least 512
    v = R.planted_matrix(m, n, min(k, 32), seed=23, dtype=np.float32)
    rs = np.random.RandomState(24)
    return m, n, k, v, np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))


KW = dict(distance_type="eu", min_iter=14, max_iter=14, lambda_w=0.01, lambda_h=0.02)


def _run_mur_case(nd, Shard, make_comm, rank, world, outdir, tag, k=12, wide=False):
    """One MUR case over this process group; returns False when negotiate() refused (recorded)."""
    m, n, k, v, w0, h0 = (_case_wide if wide else _case)(k)
    r0, r1 = nd.row_range(m, rank, world)
    shard = Shard(v[r0:r1], k, w0[r0:r1], h0, 0)
    comm = make_comm(shard)
    try:
        shard.negotiate(comm)
    except RuntimeError as e:
        np.savez(os.path.join(outdir, f"{tag}rank{rank}.npz"), refused=str(e))
        shard.close()
        return False
    pieces = shard.chunk_ranges(0, nd._exchange_chunks())
    res = nd.mur_sharded(shard, comm, batch=5, **KW)
    if isinstance(comm, nd.NativeComm):                  # did the iterations take the reduce-scatter / all-gather exchange?
        sliced = shard.eng.comm_get_exchange() == 1 and shard.eng.mur_slice_info(0, world)[0] > 0
    else:
        sliced = nd._slice_plan(shard, comm, 0) is not None
    np.savez(os.path.join(outdir, f"{tag}rank{rank}.npz"), w=res.w, h=res.h, i=res.i, obj=np.asarray(res.obj_history),
             pieces=len(pieces) if pieces else 1, merged=int(shard.merge_objective()), sliced=int(sliced))
    shard.close()
    return True


def _worker(rank, world, rdzv, backend, outdir, k=12, wide=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv          # (file:// rendezvous made by conftest.spawn_ranks; nmf_amd.dist reads it too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    Shard, make_comm = _join(backend, rdzv, rank, world)
    from nmf_amd import dist as nd
    if rank == 1 and os.environ.get("NMFX_TEST_RANK1_ENV"):          # one rank in another mode (negotiate() must settle it)
        key, val = os.environ["NMFX_TEST_RANK1_ENV"].split("=")
        os.environ[key] = val
    _run_mur_case(nd, Shard, make_comm, rank, world, outdir, "", k, wide)
    dist.barrier()
    dist.destroy_process_group()


MUR_KS = [12, 40, 100, 160]     # exact-f32 path; split-bf16 kp = 64 (fused epilogues); kp = 128; generic path (k > 128)
WIDE_KS = [40, 100]
RSAG_KS = [12, 40, 100]         # (12: exact-f32 epilogues have no sliced phase B -> all-reduce; 40, 100: k padded to 64 / 128)
SOLVERS = ["ao_admm", "ao_admm_bf16", "ao_admm_early", "ao_admm_unfused", "ao_admm_kl", "ao_admm_k160", "ao_admm_k520", "ao_admm_kl_k160", "admm",
           "admm_bf16", "admm_kl", "admm_k160", "anls", "anls_k160"]
BACKENDS = [(1, "nccl"), (2, "gloo"), (1, "native")]


def _batch_worker(rank, world, rdzv, backend, outdir):
    """Every (world, backend) case of this file in ONE set of rank processes (r5: the suite spawned 54 process groups for them, 3 s of
    interpreter + torch start-up each): the MUR cases, the chunked-exchange cases and the twelve solver cases, one after the other
    over the same process group, each with its own shard, communicator and output files `<job>.rank<r>.npz`.  A job that raises
    leaves `<job>.rank<r>.err` and ends this rank's batch (its peers' collectives fail with it)."""
    import traceback
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    Shard, make_comm = _join(backend, rdzv, rank, world)
    from nmf_amd import dist as nd
    jobs = [(f"mur{k}", "mur", k) for k in MUR_KS] + [(f"wide{k}", "wide", k) for k in WIDE_KS] + [(f"rsag{k}", "rsag", k) for k in RSAG_KS] + \
           [(f"solver_{s}", "solver", s) for s in SOLVERS]
    for tag, kind, arg in jobs:
        try:
            os.environ.pop("NMFX_DIST_CHUNKS", None)
            os.environ.pop("NMFX_DIST_EXCHANGE", None)
            if kind == "wide":
                os.environ["NMFX_DIST_CHUNKS"] = "2"
            if kind == "rsag":
                os.environ["NMFX_DIST_EXCHANGE"] = "rsag"
            if kind in ("mur", "wide", "rsag"):
                _run_mur_case(nd, Shard, make_comm, rank, world, outdir, tag + ".", arg, kind == "wide")
            else:
                _run_solver_case(nd, Shard, make_comm, rank, world, outdir, tag + ".", arg)
        except BaseException:  # noqa: BLE001
            with open(os.path.join(outdir, f"{tag}.rank{rank}.err"), "w") as fh:
                fh.write(traceback.format_exc())
            break
    os.environ.pop("NMFX_DIST_CHUNKS", None)
    os.environ.pop("NMFX_DIST_EXCHANGE", None)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001  (a peer that left early: the error file says why)
        pass


@pytest.fixture(scope="module")
def batch(tmp_path_factory):
    """batch(world, backend) -> directory with the outputs of _batch_worker, run once per (world, backend)."""
    done = {}

    def get(world, backend):
        key = (world, backend)
        if key not in done:
            d = tmp_path_factory.mktemp(f"batch_{world}_{backend}")
            try:
                spawn_ranks(_batch_worker, (world, None, backend, str(d)), world)
            except Exception as e:  # noqa: BLE001  (the jobs that finished are still checked; the others report the failure)
                (d / "spawn.err").write_text(repr(e))
            done[key] = d
        return done[key]
    return get


def _parts(d, tag, world):
    import glob
    errs = sorted(glob.glob(str(d / "*.err")))
    missing = [r for r in range(world) if not (d / f"{tag}.rank{r}.npz").exists()]
    if missing:
        pytest.fail(f"job {tag}: no output of rank(s) {missing}; " + " | ".join(f"{os.path.basename(e)}: {open(e).read()[-1500:]}" for e in errs))
    return [np.load(d / f"{tag}.rank{r}.npz") for r in range(world)]


_ORACLE_CACHE = {}


def _mur_oracle(k, wide):
    from oracle import nmf_ref as R
    key = ("mur", k, wide)
    if key not in _ORACLE_CACHE:
        m, n, k, v, w0, h0 = (_case_wide if wide else _case)(k)
        _ORACLE_CACHE[key] = (v, R.mur(v.astype(np.float64), k, w0=w0, h0=h0, **KW))
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("k", MUR_KS)
@pytest.mark.parametrize("world,backend", BACKENDS)
def test_sharded_device_path(world, backend, k, batch):
    parts = _parts(batch(world, backend), f"mur{k}", world)
    v, ref = _mur_oracle(k, False)
    w = np.concatenate([p["w"] for p in parts])
    h = parts[0]["h"]
    err = np.linalg.norm(w @ h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=2e-5)      # (measured: 1.7e-6)
        np.testing.assert_array_equal(p["h"], h)        # replicated H is bit-identical on all ranks


@pytest.mark.parametrize("k", WIDE_KS)
@pytest.mark.parametrize("world,backend", BACKENDS)
def test_sharded_device_path_with_a_chunked_exchange(world, backend, k, batch):
    """NMFX_DIST_CHUNKS=2: phase A in two column chunks (nmfx_mur_phase_a_head / _cols), each chunk's range of the exchange
    buffer reduced on its own -- over RCCL asynchronously, behind the product of the next chunk.  Same bars as the
    one-piece exchange (the chunks only change the summation order of the product's splits); H bit-identical on all ranks."""
    parts = _parts(batch(world, backend), f"wide{k}", world)
    v, ref = _mur_oracle(k, True)
    assert all(int(p["pieces"]) == 2 for p in parts)
    w = np.concatenate([p["w"] for p in parts])
    h = parts[0]["h"]
    err = np.linalg.norm(w @ h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=2e-5)
        np.testing.assert_array_equal(p["h"], h)


@pytest.mark.parametrize("k", RSAG_KS)
@pytest.mark.parametrize("world,backend", BACKENDS)
def test_sharded_device_path_with_the_reduce_scatter_all_gather_exchange(world, backend, k, batch):
    """NMFX_DIST_EXCHANGE=rsag (r5): reduce-scatter of the W^T V part of the exchange buffer, every rank updates ITS
    columns of H (nmfx_mur_phase_b_slice) and leaves them in its range, all-gather, the other ranks' columns into H and the bf16
    images (nmfx_mur_phase_b_rest) -- through torch.distributed ("nccl": world of one), staged through the host (two ranks on
    one GPU) and inside nmfx_mur_run_sharded ("native").  Same bars as the all-reduce; H bit-identical on all ranks; and on this box
    (world of one: the scatter is a copy; staged: the scatter IS an all-reduce) the very bits of the all-reduce run.  k = 12 runs
    the exact-f32 epilogues, which have no sliced phase B: every rank takes the all-reduce."""
    d = batch(world, backend)
    parts = _parts(d, f"rsag{k}", world)
    base = _parts(d, f"mur{k}", world)
    v, ref = _mur_oracle(k, False)
    assert all(int(p["sliced"]) == (1 if k > 32 else 0) for p in parts), [int(p["sliced"]) for p in parts]
    assert all(int(p["sliced"]) == 0 for p in base)
    w = np.concatenate([p["w"] for p in parts])
    h = parts[0]["h"]
    err = np.linalg.norm(w @ h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    for p, q in zip(parts, base):
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=2e-5)
        np.testing.assert_array_equal(p["h"], h)
        np.testing.assert_array_equal(p["h"], q["h"])
        np.testing.assert_array_equal(p["w"], q["w"])
        np.testing.assert_array_equal(p["obj"], q["obj"])


def test_ranks_agree_on_the_collective_sequence_before_the_first_exchange(tmp_path, monkeypatch):
    """every rank used to pick its collective sequence from its OWN engine state (merged objective: one all-reduce
    instead of two; chunk unit) -- a rank that had silently fallen back to the exact-f32 kernels would have sent another
    sequence than its peers.  negotiate(): a rank that cannot merge makes every rank keep the separate f64 exchange (same iterates
    as the all-merged run); ranks in different arithmetic modes are refused on every rank, before any exchange."""
    from oracle import nmf_ref as R
    d = tmp_path / "merge"
    d.mkdir()
    monkeypatch.setenv("NMFX_TEST_RANK1_ENV", "NMFX_DIST_MERGE=0")
    spawn_ranks(_worker, (2, None, "gloo", str(d), 40), 2)
    parts = [np.load(d / f"rank{r}.npz") for r in range(2)]
    assert [int(p["merged"]) for p in parts] == [0, 0]
    m, n, k, v, w0, h0 = _case(40)
    ref = R.mur(v.astype(np.float64), k, w0=w0, h0=h0, **KW)
    w = np.concatenate([p["w"] for p in parts])
    assert np.linalg.norm(w @ parts[0]["h"] - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)) < 1e-4
    np.testing.assert_array_equal(parts[0]["h"], parts[1]["h"])
    d = tmp_path / "precision"
    d.mkdir()
    monkeypatch.setenv("NMFX_TEST_RANK1_ENV", "NMFX_PRECISION=f32")
    spawn_ranks(_worker, (2, None, "gloo", str(d), 40), 2)
    for r in range(2):
        assert "different arithmetic modes" in str(np.load(d / f"rank{r}.npz")["refused"])


def test_sharded_device_path_with_the_separate_objective_exchange(tmp_path, monkeypatch):
    """NMFX_DIST_MERGE=0: the f64 objective partial in its own all-reduce (the only form for the exact-f32 epilogues) instead of
    inside the f32 buffer -- the same iterates bit for bit, the same stop index."""
    import torch.multiprocessing as mp
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NMFX_DIST_MERGE", mode)
        d = tmp_path / mode
        d.mkdir()
        spawn_ranks(_worker, (2, None, "gloo", str(d), 40), 2)
        outs[mode] = [np.load(d / f"rank{r}.npz") for r in range(2)]
    for r in range(2):
        np.testing.assert_array_equal(outs["1"][r]["w"], outs["0"][r]["w"])
        np.testing.assert_array_equal(outs["1"][r]["h"], outs["0"][r]["h"])
        assert int(outs["1"][r]["i"]) == int(outs["0"][r]["i"])
        np.testing.assert_allclose(outs["1"][r]["obj"], outs["0"][r]["obj"], rtol=1e-14)


def _graph_worker(rank, world, rdzv, outdir, backend="nccl", k=12):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv          # (file:// rendezvous made by conftest.spawn_ranks; nmf_amd.dist reads it too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    Shard, make_comm = _join(backend, rdzv, rank, world)
    from nmf_amd import dist as nd
    m, n, k, v, w0, h0 = _case(k)
    out = {}
    # a run that the stop rule ends in the middle of a replayed pair, and one that exhausts max_iter
    for name, kw in (("stop", dict(distance_type="eu", min_iter=5, max_iter=400, tol1=1e-5, tol2=2e-1)),
                     ("full", dict(distance_type="eu", min_iter=99, max_iter=37)),
                     ("kl", dict(distance_type="kl", min_iter=99, max_iter=21, lambda_w=0.01))):
        for graph in (False, True):
            shard = Shard(v, k, w0, h0, 0)
            comm = make_comm(shard)
            runner_modes = []
            orig = nd.Runner.__call__

            def spy(self, first, count, _o=orig, _m=runner_modes):
                _o(self, first, count)
                _m.append(self.mode)
            nd.Runner.__call__ = spy
            try:
                res = nd.mur_sharded(shard, comm, batch=7, graph=graph, **kw)
            finally:
                nd.Runner.__call__ = orig
            tag = f"{name}_{'graph' if graph else 'eager'}"
            out[tag + "_w"], out[tag + "_h"] = res.w, res.h
            out[tag + "_i"], out[tag + "_obj"] = res.i, np.asarray(res.obj_history)
            out[tag + "_mode"] = np.array(runner_modes[-1])
            out[tag + "_replays"] = shard.eng.comm_graph_replays() if backend == "native" else -1
            shard.close()
    np.savez(os.path.join(outdir, "graph.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,k,exchange", [("nccl", 12, None), ("native", 12, None), ("native", 40, None), ("nccl", 40, None), ("native", 40, "rsag"), ("nccl", 40, "rsag")])
def test_graphed_loop_equals_eager_loop(backend, k, exchange, tmp_path, monkeypatch):
    """hipGraph replays of the sharded iteration (RCCL all-reduce captured inside) give the very
    same iterates, objective history and stop index as the eager loop -- captured by torch around the phase calls ("nccl") or
    by the library itself inside nmfx_mur_run_sharded ("native": the whole loop is one C call, no torch on the data path)."""
    import torch.multiprocessing as mp
    if exchange:                                         # (r5: the reduce-scatter . slice . all-gather . rest sequence inside the captured pair)
        monkeypatch.setenv("NMFX_DIST_EXCHANGE", exchange)
    spawn_ranks(_graph_worker, (1, None, str(tmp_path), backend, k), 1)
    z = np.load(tmp_path / "graph.npz")
    for name in ("stop", "full", "kl"):
        if backend == "native":
            assert str(z[f"{name}_graph_mode"]) == "native-hipgraph" and str(z[f"{name}_eager_mode"]) == "native"
            assert int(z[f"{name}_graph_replays"]) > 0 and int(z[f"{name}_eager_replays"]) == 0
        else:
            assert str(z[f"{name}_graph_mode"]) == "hipgraph" and str(z[f"{name}_eager_mode"]) == "eager"
        assert int(z[f"{name}_graph_i"]) == int(z[f"{name}_eager_i"])
        if backend == "nccl" and name == "kl" and k > 32:
            # MUR-KL on the split-bf16 kernels starts an iteration from the H images / row-sum partials its predecessor's epilogue
            # left (kl_h_iter == j - 1) or rebuilds them -- another order of additions.  The library's own capture tells the captured
            # pair which of the two the eager loop does at that point (comm.hip: g_klfresh); a pair captured by torch AROUND the phase
            # calls starts at the relative index 0 and rebuilds in every replay: the same iterates to rounding, not to the bit.
            np.testing.assert_allclose(z[f"{name}_graph_obj"], z[f"{name}_eager_obj"], rtol=2e-6)      # (measured: 1.7e-7)
            for key in ("w", "h"):
                a, b = z[f"{name}_graph_{key}"], z[f"{name}_eager_{key}"]
                assert np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(b)
            continue
        np.testing.assert_array_equal(z[f"{name}_graph_obj"], z[f"{name}_eager_obj"])
        np.testing.assert_array_equal(z[f"{name}_graph_w"], z[f"{name}_eager_w"])
        np.testing.assert_array_equal(z[f"{name}_graph_h"], z[f"{name}_eager_h"])
    assert int(z["stop_graph_i"]) < 399 and int(z["full_graph_i"]) == 36


# ---- AO-ADMM and ANLS over row shards (device engine) ------------------------
def _solver_case(solver):
    from oracle import nmf_ref as R
    if solver in ("ao_admm_early", "ao_admm_unfused"):     # uniform data: the inner loops stop early -> the REPAIR launch runs
        m, n, k = 192, 128, 5
        v = np.random.RandomState(12).rand(m, n).astype(np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        kw = dict(reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=2, max_iter=60, admm_iter=10, tol1=1e-3, tol2=1e-2)
        return m, n, k, v, w0, h0, kw
    if solver.startswith("admm"):
        m, n, k = 520, 300, (40 if solver.endswith("bf16") else 12)
        kw = dict(rho=1.0, distance_type="kl" if solver == "admm_kl" else "eu", reg_w=(0.05, "l1n"),
                  reg_h=(0.05, "l1n") if solver == "admm_kl" else (0.2, "l2n"), min_iter=6, max_iter=6)
        v = R.planted_matrix(m, n, k, seed=33, dtype=np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        return m, n, k, v, w0, h0, kw
    if solver == "ao_admm_k520":                           # r5: beyond 512 components the W rounds take the rhs / solve / prox launches (k pads to 640)
        m, n, k = 700, 640, 520
        kw = dict(reg_w=(0.1, "l1n"), reg_h=(0.05, "l1n"), min_iter=3, max_iter=3, admm_iter=8)
        v = R.planted_matrix(m, n, 24, seed=40, dtype=np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        return m, n, k, v, w0, h0, kw
    if solver == "ao_admm_kl_k160":                        # r5: the KL loss beyond 128 components over row shards (generic kernels, nmfx_generic_aoadmm_kl_phase)
        m, n, k = 384, 320, 160
        kw = dict(distance_type="kl", reg_w=(0.02, "l1n"), reg_h=(0, "nn"), min_iter=3, max_iter=3, admm_iter=6)
        v = R.planted_matrix(m, n, 24, seed=36, dtype=np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        return m, n, k, v, w0, h0, kw
    if solver == "ao_admm_kl":                             # KL loss: one exchange per inner round of both sub-problems
        m, n, k = 320, 200, 8
        kw = dict(distance_type="kl", reg_w=(0.02, "l1n"), reg_h=(0, "nn"), min_iter=4, max_iter=4, admm_iter=6)
        v = R.planted_matrix(m, n, k, seed=35, dtype=np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        return m, n, k, v, w0, h0, kw
    if solver == "admm_k160":
        m, n, k = 520, 384, 160
        kw = dict(rho=1.0, distance_type="eu", reg_w=(0.05, "l1n"), reg_h=(0.05, "l1n"), min_iter=4, max_iter=4)
        v = R.planted_matrix(m, n, 24, seed=38, dtype=np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        return m, n, k, v, w0, h0, kw
    if solver == "anls_k160":
        m, n, k = 240, 200, 160                            # (small: the oracle's per-column scipy NNLS at k = 160 is the slow part)
        kw = dict(lambda_w=0.1, lambda_h=0.05, min_iter=1, max_iter=2, tol1=1e-3, tol2=1e-3)
        v = R.planted_matrix(m, n, 24, seed=39, dtype=np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        return m, n, k, v, w0, h0, kw
    if solver == "ao_admm_k160":                           # r4: beyond 128 components the phases are composed from the generic kernels
        m, n, k = 520, 384, 160
        kw = dict(reg_w=(0.1, "l1n"), reg_h=(0.05, "l1n"), min_iter=4, max_iter=4, admm_iter=10)
        v = R.planted_matrix(m, n, 24, seed=37, dtype=np.float32)
        w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
        return m, n, k, v, w0, h0, kw
    if solver.startswith("ao_admm"):
        m, n, k = 520, 300, (40 if solver.endswith("bf16") else 12)    # k = 40: split-bf16 products, lazy objective
        kw = dict(reg_w=(0.1, "l1n"), reg_h=(0.05, "l1n"), min_iter=8, max_iter=8, admm_iter=10)
    else:
        m, n, k = 200, 150, 6
        kw = dict(lambda_w=0.1, lambda_h=0.05, min_iter=3, max_iter=12, tol1=1e-3, tol2=1e-3)
    v = R.planted_matrix(m, n, k, seed=31, dtype=np.float32)
    w0, h0 = R.svd_init(v.astype(np.float64), k, "zero")
    return m, n, k, v, w0, h0, kw


def _run_solver_case(nd, Shard, make_comm, rank, world, outdir, tag, solver):
    m, n, k, v, w0, h0, kw = _solver_case(solver)
    r0, r1 = nd.row_range(m, rank, world)
    shard = Shard(v[r0:r1], k, w0[r0:r1], h0, 0)
    comm = make_comm(shard)
    if solver.startswith("ao_admm"):
        res = nd.aoadmm_sharded(shard, comm, batch=16, fused=(False if solver == "ao_admm_unfused" else None), **kw)
    elif solver.startswith("admm"):
        res = nd.admm_sharded(shard, comm, batch=3, **kw)
    else:
        res = nd.anls_sharded(shard, comm, batch=3, **kw)
    inner = (shard.eng.inner_counts(0, res.i + 1) & 0xFFFF) if solver.startswith("ao_admm") else np.zeros(0)
    np.savez(os.path.join(outdir, f"{tag}rank{rank}.npz"), w=res.w, h=res.h, i=res.i, obj=np.asarray(res.obj_history),
             inner=inner)
    shard.close()


def _solver_oracle(solver):
    from gpu_common import slow_oracle, slow_signature
    from oracle import nmf_ref as R
    if solver not in _ORACLE_CACHE:
        m, n, k, v, w0, h0, kw = _solver_case(solver)
        oracle = R.ao_admm if solver.startswith("ao_admm") else R.admm if solver.startswith("admm") else R.anls
        if solver == "anls_k160":      # (13 s of scipy NNLS: committed under tests/golden/slow/)
            ref = slow_oracle("dist_anls_k160", slow_signature(v, k, kw, w0, h0), lambda: oracle(v.astype(np.float64), k, w0=w0, h0=h0, **kw))
        else:
            ref = oracle(v.astype(np.float64), k, w0=w0, h0=h0, **kw)
        _ORACLE_CACHE[solver] = (v, ref)
    return _ORACLE_CACHE[solver]


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("world,backend", BACKENDS)
def test_sharded_aoadmm_anls_device_path(world, backend, solver, batch):
    parts = _parts(batch(world, backend), f"solver_{solver}", world)
    v, ref = _solver_oracle(solver)
    w = np.concatenate([p["w"] for p in parts])
    h = parts[0]["h"]
    err = np.linalg.norm(w @ h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    if solver == "ao_admm_early":
        assert any(t[1] < 10 for t in ref.trace["inner"]), "case must exercise the repair launch of the W sub-problem"
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=5e-5 if solver in ("admm_kl", "ao_admm_kl") else 2e-4 if solver == "ao_admm_kl_k160" else 3e-4 if solver.startswith("admm") else 1e-4)      # (measured: 4e-6 KL; 6.1e-5 ADMM, see test_gpu_admm.py; 2e-5 the others)
        np.testing.assert_array_equal(p["h"], h)
        if solver.startswith("ao_admm"):
            assert [tuple(r) for r in p["inner"]] == [tuple(t) for t in ref.trace["inner"]]


def _api_worker(rank, world, rdzv, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv          # (file:// rendezvous made by conftest.spawn_ranks; nmf_amd.dist reads it too)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from nmf_amd import dist as nd
    from oracle import nmf_ref as R
    v = R.planted_matrix(600, 260, 12, seed=41, dtype=np.float32)
    out = {}
    np.random.seed(7)
    res = nd.factorize(v, 12, method="mur", backend="gloo", distance_type="eu", min_iter=10, max_iter=10)
    out["mur_w"], out["mur_h"], out["mur_obj"] = res.w, res.h, np.asarray(res.obj_history)
    res = nd.factorize(v, 12, method="ao_admm", backend="gloo", reg_w=(0.05, "l1n"), reg_h=(0.05, "l1n"), min_iter=4, max_iter=4)
    out["ao_w"], out["ao_h"], out["ao_obj"] = res.w, res.h, np.asarray(res.obj_history)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_factorize_api_two_ranks_on_one_gpu(tmp_path):
    """nmf_amd.dist.factorize end to end on the device engine: reference keywords / defaults, seeds, NNDSVD from
    rank 0, gathered W on rank 0 (two ranks share the GPU: exchange staged through the host over gloo)."""
    import torch.multiprocessing as mp
    from oracle import nmf_ref as R
    spawn_ranks(_api_worker, (2, None, str(tmp_path)), 2)
    v = R.planted_matrix(600, 260, 12, seed=41, dtype=np.float32)
    z = np.load(tmp_path / "rank0.npz")
    np.random.seed(7)
    ref = R.mur(v.astype(np.float64), 12, distance_type="eu", min_iter=10, max_iter=10)
    assert z["mur_w"].shape == (600, 12)
    assert np.linalg.norm(z["mur_w"] @ z["mur_h"] - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)) < 1e-4
    np.testing.assert_allclose(z["mur_obj"], ref.obj_history, rtol=1e-6)      # (measured: 1.7e-8)
    ref = R.ao_admm(v.astype(np.float64), 12, reg_w=(0.05, "l1n"), reg_h=(0.05, "l1n"), min_iter=4, max_iter=4)
    assert np.linalg.norm(z["ao_w"] @ z["ao_h"] - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)) < 1e-4
    np.testing.assert_allclose(z["ao_obj"], ref.obj_history, rtol=2e-5)      # (measured: 1.2e-6)


def _api_native_worker(rank, world, rdzv, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from nmf_amd import dist as nd
    from oracle import nmf_ref as R
    v = R.planted_matrix(600, 260, 12, seed=41, dtype=np.float32)
    out = {}
    taken = []
    real = nd.make_sharded
    nd.make_sharded = lambda *a, **kw: (lambda r: (taken.append(r[2]), r)[1])(real(*a, **kw))
    for tag, env, brk in (("torch", None, False), ("native", "1", False), ("broken", "1", True)):
        if env is None:
            os.environ.pop("NMFX_DIST_NATIVE", None)
        else:
            os.environ["NMFX_DIST_NATIVE"] = env
        create = nd.NativeComm.create
        if brk:                                          # the communicator does not come up on this rank: everybody falls back
            nd.NativeComm.create = staticmethod(lambda shard: (_ for _ in ()).throw(RuntimeError("no communicator (test)")))
        try:
            np.random.seed(7)
            res = nd.factorize(v, 12, method="mur", backend="nccl", distance_type="eu", min_iter=10, max_iter=10)
        finally:
            nd.NativeComm.create = create
        out[tag + "_w"], out[tag + "_h"], out[tag + "_obj"] = res.w, res.h, np.asarray(res.obj_history)
    out["taken"] = np.array(taken)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_factorize_takes_the_native_exchange_only_on_request_and_falls_back_together(tmp_path):
    """nmf_amd.dist.factorize on RCCL runs torch.distributed's collectives by default; NMFX_DIST_NATIVE=1 asks for the
    communicator behind the C ABI, and a rank on which it does not come up takes every rank to the torch path (the outcome of each
    step is all-reduced: nmf_amd.dist.make_sharded) instead of leaving its peers in a collective.  World of one on RCCL."""
    from oracle import nmf_ref as R
    spawn_ranks(_api_native_worker, (1, None, str(tmp_path)), 1)
    z = np.load(tmp_path / "rank0.npz")
    taken = [str(t) for t in z["taken"]]
    assert taken[0].startswith("torch.distributed") and taken[1].startswith("native") and taken[2].startswith("torch.distributed"), taken
    v = R.planted_matrix(600, 260, 12, seed=41, dtype=np.float32)
    np.random.seed(7)
    ref = R.mur(v.astype(np.float64), 12, distance_type="eu", min_iter=10, max_iter=10)
    for tag in ("torch", "native", "broken"):
        assert np.linalg.norm(z[tag + "_w"] @ z[tag + "_h"] - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)) < 1e-4
        np.testing.assert_allclose(z[tag + "_obj"], ref.obj_history, rtol=1e-6)
    np.testing.assert_array_equal(z["torch_obj"], z["broken_obj"])


@pytest.mark.parametrize("launch", ["torchrun", "self", "deadline"])
def test_bench_two_ranks_on_one_gpu_runs_both_sharded_legs(launch, tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank) and as a plain
    `python bench.py --gpus 2` (no WORLD_SIZE: bench.py starts torch.distributed.run itself, as a child, before anything touches the
    GPU), rehearsed with two ranks on ONE GPU (gloo staged through the host, shrunken shapes): the strong-scaling leg of config 2
    and the config-5 leg (each rank draws its own rows of the same matrix on the device) both come back in the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NMFX_BENCH_BACKEND="gloo", NMFX_BENCH_CFG5_SHAPE="2048x1024x128", NMFX_BENCH_SHAPE="1024x512x64",
               NMF_AMD_QUIET="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NMFX_BENCH_DETAIL=str(tmp_path / "detail.json"))
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NMFX_DIST_INIT_METHOD"):
        env.pop(key, None)
    tail = ["--gpus", "2", "--steps", "4", "--warmup", "2", "--preheat", "0", "--profile-steps", "2", "--no-cpu", "--no-traffic",
            "--tol-max-iter", "0"]
    if launch == "deadline":                             # (r5) the config-5 leg "never comes back": the line is printed without it, exit code 0
        env["NMFX_BENCH_CFG5_DEADLINE"] = "0.01"
    for _ in range(1 if launch != "torchrun" else 4):    # (torchrun binds the port probed here a moment later; bench.py's own launch handles that itself)
        head = [sys.executable] if launch != "torchrun" else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(_free_port())]
        p = subprocess.run(head + [os.path.join(root, "bench.py")] + tail, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        if p.returncode == 0 or "EADDRINUSE" not in p.stderr:
            break
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["rows_per_gpu"] == 512
    assert len(lines[0]) < 4096                          # (the compact headline: the full record is in the detail file)
    if launch == "deadline":
        assert "error" in line["cfg5"] and "did not come back" in line["cfg5"]["error"], line
        return
    assert line["cfg5"]["n_gpus"] == 2 and line["cfg5"]["iter_per_s"] > 0, line
    detail = json.loads((tmp_path / "detail.json").read_text())
    assert "strong_scaling_quoted_on" in detail and detail["value"] == line["value"]
    leg = detail["other_configs"][0]
    assert leg.get("config") == "cfg5" and "error" not in leg, leg
    assert leg["n_gpus"] == 2 and leg["rows_per_gpu"] == 1024 and leg["objective_decreasing"] and leg["iter_per_s"] > 0


def test_sharded_entry_points_need_a_communicator():
    """nmfx_mur_run_sharded / nmfx_comm_all_reduce without nmfx_comm_init_rank: NMFX_E_STATE, not a crash."""
    from nmf_amd._lib import NmfxError
    from nmf_amd.engine import Engine
    m, n, k, v, w0, h0 = _case()
    with Engine(m, n, k) as e:
        e.upload_v(v)
        e.set_factors(w0, h0)
        with pytest.raises(NmfxError, match="communicator"):
            e.mur_run_sharded(0, 0.0, 0.0, 10 ** 9, 1e-5, 1e-5, 0, 1)
        with pytest.raises(NmfxError, match="communicator"):
            e.comm_all_reduce(0, 0, 8)
        assert e.comm_info()[:3] == (0, 1, False)
