"""Row-sharded MUR (nmf_amd/dist.py) with world_size 2 over gloo on the CPU.
The HIP engine is replaced by tests/host_shard.py; the loop, the exchange
protocol and the stop logic are the product code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from host_shard import HostShard
    from nmf_amd import dist as nd
    from oracle import nmf_ref as R
    m, n, k = case["m"], case["n"], case["k"]
    v = R.planted_matrix(m, n, k, seed=case["seed"], dtype=np.float64)
    rs = np.random.RandomState(case["seed"] + 1)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    r0, r1 = nd.row_range(m, rank, world)
    shard = HostShard(v[r0:r1], k, w0[r0:r1], h0)
    res = nd.mur_sharded(shard, nd.TorchComm(), batch=case["batch"], **case["kw"])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=res.w, h=res.h, i=res.i,
             obj=np.asarray(res.obj_history), r0=r0, r1=r1)
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    dict(m=150, n=90, k=5, seed=3, batch=7,
         kw=dict(distance_type="eu", min_iter=12, max_iter=12, lambda_w=0.05, lambda_h=0.1)),
    dict(m=131, n=77, k=4, seed=4, batch=16,       # converges mid-batch: every rank must stop at the same i
         kw=dict(distance_type="eu", min_iter=5, max_iter=400, tol1=1e-9, tol2=2e-4)),
    dict(m=96, n=64, k=3, seed=5, batch=5,
         kw=dict(distance_type="kl", min_iter=9, max_iter=9, lambda_w=0.0, lambda_h=0.02)),
]


@pytest.mark.parametrize("case", CASES, ids=["eu_lambda", "eu_converge", "kl"])
def test_sharded_mur_equals_single_process_oracle(case, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    from oracle import nmf_ref as R
    m, n, k = case["m"], case["n"], case["k"]
    v = R.planted_matrix(m, n, k, seed=case["seed"], dtype=np.float64)
    rs = np.random.RandomState(case["seed"] + 1)
    w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    ref = R.mur(v, k, w0=w0, h0=h0, **case["kw"])
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    w = np.concatenate([p["w"] for p in parts])
    assert [(int(p["r0"]), int(p["r1"])) for p in parts] == [(0, m // 2), (m // 2, m)]
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
def _solver_worker(rank, world, port, case, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    mp.spawn(_solver_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
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
