"""Row-sharded MUR (nmf_amd/dist.py) with world_size 2 over gloo on the CPU.
The HIP engine is replaced by tests/host_shard.py; the loop, the exchange
protocol and the stop logic are the product code."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import spawn_ranks
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, rdzv, case, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv          # (file:// rendezvous made by conftest.spawn_ranks; nmf_amd.dist reads it too)
    dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
    from host_shard import HostShard
    from nmf_amd import dist as nd
    from oracle import nmf_ref as R
    m, n, k = case["m"], case["n"], case["k"]
    v = R.planted_matrix(m, n, k, seed=case["seed"], dtype=np.float64)
    rs = np.random.RandomState(case["seed"] + 1)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    r0, r1 = nd.row_range(m, rank, world)
    if case.get("chunks"):
        from host_shard import ChunkedHostShard
        os.environ["NMFX_DIST_CHUNKS"] = str(case["chunks"])
        shard = ChunkedHostShard(v[r0:r1], k, w0[r0:r1], h0)
    elif case.get("exchange"):
        from host_shard import SlicedHostShard
        os.environ["NMFX_DIST_EXCHANGE"] = case["exchange"]
        shard = SlicedHostShard(v[r0:r1], k, w0[r0:r1], h0)
    else:
        shard = HostShard(v[r0:r1], k, w0[r0:r1], h0)
    res = nd.mur_sharded(shard, nd.TorchComm(), batch=case["batch"], **case["kw"])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=res.w, h=res.h, i=res.i,
             obj=np.asarray(res.obj_history), r0=r0, r1=r1, pieces=len(getattr(shard, "cols_seen", [0])),
             slices=np.asarray(sorted(set(getattr(shard, "slices_seen", []))), dtype=np.int64).reshape(-1, 2))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    dict(m=150, n=90, k=5, seed=3, batch=7,
         kw=dict(distance_type="eu", min_iter=12, max_iter=12, lambda_w=0.05, lambda_h=0.1)),
    dict(m=131, n=77, k=4, seed=4, batch=16,       # converges mid-batch: every rank must stop at the same i
         kw=dict(distance_type="eu", min_iter=5, max_iter=400, tol1=1e-9, tol2=2e-4)),
    dict(m=96, n=64, k=3, seed=5, batch=5,
         kw=dict(distance_type="kl", min_iter=9, max_iter=9, lambda_w=0.0, lambda_h=0.02)),
    # NMFX_DIST_CHUNKS: phase A in column chunks, every chunk's range of the [column][factor] buffer reduced on its own
    dict(m=150, n=90, k=5, seed=3, batch=7, chunks=3,
         kw=dict(distance_type="eu", min_iter=12, max_iter=12, lambda_w=0.05, lambda_h=0.1)),
    dict(m=131, n=77, k=4, seed=4, batch=16, chunks=2,
         kw=dict(distance_type="eu", min_iter=5, max_iter=400, tol1=1e-9, tol2=2e-4)),
    # NMFX_DIST_EXCHANGE=rsag: reduce-scatter of the W^T V part, each rank updates its half of H's columns, all-gather (r5)
    dict(m=150, n=90, k=5, seed=3, batch=7, exchange="rsag",
         kw=dict(distance_type="eu", min_iter=12, max_iter=12, lambda_w=0.05, lambda_h=0.1)),
    dict(m=131, n=80, k=4, seed=4, batch=16, exchange="rsag",     # converges mid-batch: the stop must leave H alone on every rank
         kw=dict(distance_type="eu", min_iter=5, max_iter=400, tol1=1e-9, tol2=2e-4)),
    dict(m=96, n=64, k=3, seed=5, batch=5, exchange="rsag",       # KL loss: no sliced form -> every rank takes the all-reduce
         kw=dict(distance_type="kl", min_iter=9, max_iter=9, lambda_w=0.0, lambda_h=0.02)),
    dict(m=150, n=91, k=5, seed=3, batch=7, exchange="rsag",      # n not a multiple of the world: all-reduce as well
         kw=dict(distance_type="eu", min_iter=12, max_iter=12, lambda_w=0.05, lambda_h=0.1)),
]


@pytest.mark.parametrize("case", CASES, ids=["eu_lambda", "eu_converge", "kl", "eu_lambda_3_chunks", "eu_converge_2_chunks",
                                             "eu_lambda_rsag", "eu_converge_rsag", "kl_rsag_falls_back", "eu_odd_n_rsag_falls_back"])
def test_sharded_mur_equals_single_process_oracle(case, tmp_path):
    world = 2
    spawn_ranks(_worker, (world, None, case, str(tmp_path)), world)
    from oracle import nmf_ref as R
    m, n, k = case["m"], case["n"], case["k"]
    v = R.planted_matrix(m, n, k, seed=case["seed"], dtype=np.float64)
    rs = np.random.RandomState(case["seed"] + 1)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    ref = R.mur(v, k, w0=w0, h0=h0, **case["kw"])
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    w = np.concatenate([p["w"] for p in parts])
    assert [(int(p["r0"]), int(p["r1"])) for p in parts] == [(0, m // 2), (m // 2, m)]
    if case.get("chunks"):
        assert all(int(p["pieces"]) >= 2 for p in parts)          # (the last iteration's phase A came in pieces)
    if case.get("exchange") == "rsag":
        sliced = case["kw"]["distance_type"] == "eu" and n % world == 0
        for r, p in enumerate(parts):                            # each rank updated exactly its own columns -- or none at all
            assert p["slices"].tolist() == ([[r * n // world, (r + 1) * n // world]] if sliced else [])
        np.testing.assert_array_equal(parts[0]["h"], parts[1]["h"])      # replicated H bit-identical (each column computed once, copied)
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=1e-10)
        np.testing.assert_allclose(p["h"], ref.h, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(w, ref.w, rtol=1e-9, atol=1e-300)
    if case["kw"]["max_iter"] > 100:
        assert ref.trace["stop_rule"] == 2 and ref.i < case["kw"]["max_iter"] - 1


def test_row_range_partitions_any_shape():
    from nmf_amd.dist import row_range
    for m in (1, 7, 64, 1000, 16384):
        for world in (1, 2, 3, 8):
            spans = [row_range(m, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- AO-ADMM and ANLS over row shards ---------------------------------------
def _solver_worker(rank, world, rdzv, case, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv          # (file:// rendezvous made by conftest.spawn_ranks; nmf_amd.dist reads it too)
    dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
    from host_shard import HostShard
    from nmf_amd import dist as nd
    v, w0, h0 = _solver_inputs(case)
    r0, r1 = nd.row_range(case["m"], rank, world)
    shard = HostShard(v[r0:r1], case["k"], w0[r0:r1], h0)
    run = nd.aoadmm_sharded if case["solver"] == "ao_admm" else nd.anls_sharded
    res = run(shard, nd.TorchComm(), batch=case["batch"], **case["kw"])
    inner = np.array([[shard.inner[(j, 0)], shard.inner[(j, 1)]] for j in range(res.i + 1)]) \
        if case["solver"] == "ao_admm" else np.zeros(0)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=res.w, h=res.h, i=res.i,
             obj=np.asarray(res.obj_history), inner=inner)
    dist.barrier()
    dist.destroy_process_group()


def _solver_inputs(case):
    from oracle import nmf_ref as R
    m, n, k = case["m"], case["n"], case["k"]
    if case.get("uniform"):
        v = np.random.RandomState(case["seed"]).rand(m, n)
    else:
        v = R.planted_matrix(m, n, k, seed=case["seed"], dtype=np.float64)
    w0, h0 = R.svd_init(v, k, "zero") if case.get("svd") else \
        (lambda rs: (np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))))(np.random.RandomState(case["seed"] + 1))
    return v, w0, h0


SOLVER_CASES = [
    dict(solver="ao_admm", m=120, n=84, k=6, seed=11, batch=3, svd=True,
         kw=dict(reg_w=(0.1, "l1n"), reg_h=(0.05, "l1n"), min_iter=6, max_iter=6, admm_iter=10)),
    dict(solver="ao_admm", m=96, n=70, k=5, seed=12, batch=2, svd=True, uniform=True,   # inner loops stop early
         kw=dict(reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=2, max_iter=60, admm_iter=10, tol1=1e-3, tol2=1e-2)),
    dict(solver="anls", m=60, n=44, k=4, seed=13, batch=3, svd=True,
         kw=dict(lambda_w=0.1, lambda_h=0.05, min_iter=3, max_iter=40, tol1=1e-3, tol2=1e-3)),
]


@pytest.mark.parametrize("case", SOLVER_CASES, ids=["ao_admm_l1n", "ao_admm_early_exit", "anls"])
def test_sharded_aoadmm_anls_equal_single_process_oracle(case, tmp_path):
    """Two row shards give what the single-process restatement of the reference gives: same
    objective history, stop index, inner-iteration counts (the `terminate` norms span both
    shards) and factors."""
    world = 2
    spawn_ranks(_solver_worker, (world, None, case, str(tmp_path)), world)
    from oracle import nmf_ref as R
    v, w0, h0 = _solver_inputs(case)
    ref = (R.ao_admm if case["solver"] == "ao_admm" else R.anls)(v, case["k"], w0=w0, h0=h0, **case["kw"])
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    w = np.concatenate([p["w"] for p in parts])
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=1e-8)
        np.testing.assert_allclose(p["h"], ref.h, rtol=1e-6, atol=1e-9)
        if case["solver"] == "ao_admm":
            assert [tuple(r) for r in p["inner"]] == [tuple(t) for t in ref.trace["inner"]]
    np.testing.assert_allclose(w, ref.w, rtol=1e-6, atol=1e-9)
    if case.get("uniform"):
        assert any(t[0] < 10 or t[1] < 10 for t in ref.trace["inner"]), "case must exercise the early exit"


# ---- the round-by-round exchange of the W sub-problem, ADMM, and the SPMD entry point -------------------------------
def _generic_worker(rank, world, rdzv, case, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["NMFX_DIST_INIT_METHOD"] = rdzv          # (file:// rendezvous made by conftest.spawn_ranks; nmf_amd.dist reads it too)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    from host_shard import HostShard
    from nmf_amd import dist as nd
    v, w0, h0 = _solver_inputs(case)
    if case.get("api"):
        # nmf_amd.dist.factorize: the reference's call surface, SPMD; the numpy stand-in replaces the HIP engine
        np.random.seed(case["seed"])
        res = nd.factorize(v, case["k"], method=case["solver"], backend="gloo", shard_factory=HostShard, **case["kw"])
        w_local = None
    else:
        dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
        r0, r1 = nd.row_range(case["m"], rank, world)
        shard = HostShard(v[r0:r1], case["k"], w0[r0:r1], h0)
        run = {"ao_admm": nd.aoadmm_sharded, "admm": nd.admm_sharded}[case["solver"]]
        res = run(shard, nd.TorchComm(), batch=case["batch"], **case["kw"], **case.get("extra", {}))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=res.w, h=res.h, i=res.i, obj=np.asarray(res.obj_history),
             exp=json.dumps([str(x) for x in (res.experiment or [])]))
    dist.barrier()
    dist.destroy_process_group()


import json  # noqa: E402

GENERIC_CASES = [
    dict(solver="ao_admm", m=96, n=70, k=5, seed=12, batch=2, svd=True, uniform=True, extra=dict(fused=False),
         kw=dict(reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=2, max_iter=60, admm_iter=10, tol1=1e-3, tol2=1e-2)),
    dict(solver="admm", m=110, n=80, k=5, seed=21, batch=4, svd=True,
         kw=dict(rho=2, distance_type="eu", reg_w=(0.05, "l1n"), reg_h=(0.3, "l2n"), min_iter=8, max_iter=8)),
    dict(solver="admm", m=90, n=66, k=4, seed=22, batch=3, svd=True,
         kw=dict(rho=1, distance_type="kl", reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=6, max_iter=6)),
    dict(solver="admm", m=90, n=66, k=4, seed=23, batch=3, svd=True,       # the row-coupled prox on the replicated side
         kw=dict(rho=1, distance_type="eu", reg_w=(0.05, "l1n"), reg_h=(0.05, "l1inf"), min_iter=3, max_iter=3)),
    dict(solver="ao_admm", m=90, n=66, k=4, seed=24, batch=2, svd=True,    # KL loss: one exchange per inner round
         kw=dict(distance_type="kl", reg_w=(0.02, "l1n"), reg_h=(0, "nn"), min_iter=4, max_iter=4, admm_iter=6)),
    dict(solver="ao_admm", m=96, n=70, k=5, seed=12, batch=2, svd=True, uniform=True,   # KL loss, long inner loops / stop rule
         kw=dict(distance_type="kl", reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=2, max_iter=12, admm_iter=25, tol1=1e-3, tol2=1e-1)),
]


@pytest.mark.parametrize("case", GENERIC_CASES, ids=["ao_admm_round_by_round", "admm_eu_l1n_l2n", "admm_kl", "admm_l1inf_h",
                                                     "ao_admm_kl", "ao_admm_kl_long_inner"])
def test_sharded_admm_and_unfused_aoadmm_equal_single_process_oracle(case, tmp_path):
    world = 2
    spawn_ranks(_generic_worker, (world, None, case, str(tmp_path)), world)
    from oracle import nmf_ref as R
    v, w0, h0 = _solver_inputs(case)
    ref = {"ao_admm": R.ao_admm, "admm": R.admm}[case["solver"]](v, case["k"], w0=w0, h0=h0, **case["kw"])
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    w = np.concatenate([p["w"] for p in parts])
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=1e-8)
        np.testing.assert_allclose(p["h"], ref.h, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(w, ref.w, rtol=1e-6, atol=1e-9)


API_CASES = [
    dict(api=True, solver="mur", m=101, n=63, k=4, seed=31, kw=dict(distance_type="eu", min_iter=9, max_iter=9, lambda_w=0.02)),
    dict(api=True, solver="mur", m=64, n=48, k=3, seed=32, kw=dict(min_iter=7, max_iter=7)),                 # default distance: 'kl'
    dict(api=True, solver="ao_admm", m=90, n=70, k=4, seed=33,
         kw=dict(reg_w=(0.05, "l1n"), reg_h=(0.05, "l1n"), min_iter=4, max_iter=4)),                           # default NNDSVD start
    dict(api=True, solver="admm", m=90, n=70, k=4, seed=34, kw=dict(reg_h=(0.2, "l2n"), min_iter=5, max_iter=5)),
    dict(api=True, solver="anls", m=50, n=40, k=3, seed=35, kw=dict(distance_type="kl", min_iter=3, max_iter=3, lambda_h=0.05)),
    dict(api=True, solver="ao_admm", m=70, n=50, k=3, seed=36,
         kw=dict(distance_type="kl", reg_w=(0, "nn"), reg_h=(0.02, "l1n"), min_iter=3, max_iter=3, admm_iter=5)),
]


@pytest.mark.parametrize("case", API_CASES, ids=["mur_eu", "mur_default_kl", "ao_admm", "admm", "anls_kl", "ao_admm_kl"])
def test_factorize_api_over_two_ranks_matches_the_single_process_reference_semantics(case, tmp_path):
    """nmf_amd.dist.factorize: reference keyword names / defaults per method, the global numpy RNG consumed in the
    reference's order on every rank, NNDSVD from rank 0 broadcast, rank 0 returns the gathered m x k factor."""
    world = 2
    spawn_ranks(_generic_worker, (world, None, case, str(tmp_path)), world)
    from oracle import nmf_ref as R
    v, _, _ = _solver_inputs(case)
    np.random.seed(case["seed"])
    ref = R.SOLVERS[case["solver"]](v.copy(), case["k"], **case["kw"])
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert parts[0]["w"].shape == (case["m"], case["k"])                 # gathered on rank 0
    assert parts[1]["w"].shape == (case["m"] - case["m"] // 2, case["k"])  # its own rows elsewhere
    np.testing.assert_allclose(parts[0]["w"], ref.w, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(parts[0]["w"][case["m"] // 2:], parts[1]["w"], rtol=0, atol=0)
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=1e-8)
        np.testing.assert_allclose(p["h"], ref.h, rtol=1e-6, atol=1e-9)
    exp = json.loads(str(parts[0]["exp"]))
    assert exp[0] == case["solver"] and exp[1] == str(case["k"])


def test_factorize_rejects_what_the_reference_rejects():
    from nmf_amd import dist as nd
    v = np.random.RandomState(0).rand(12, 9)
    with pytest.raises(Exception, match="Method not known"):
        nd.factorize(v, 3, method="nope")
