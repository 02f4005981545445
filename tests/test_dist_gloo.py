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
