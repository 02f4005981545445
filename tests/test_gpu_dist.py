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

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    from oracle import nmf_ref as R
    m, n, k = 700, 330, 12
    v = R.planted_matrix(m, n, k, seed=21, dtype=np.float32)
    rs = np.random.RandomState(22)
    return m, n, k, v, np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))


KW = dict(distance_type="eu", min_iter=14, max_iter=14, lambda_w=0.01, lambda_h=0.02)


def _worker(rank, world, port, backend, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["NMF_AMD_QUIET"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from nmf_amd import dist as nd
    m, n, k, v, w0, h0 = _case()
    r0, r1 = nd.row_range(m, rank, world)
    shard = nd.DeviceShard(v[r0:r1], k, w0[r0:r1], h0, 0)
    comm = nd.TorchComm(stage_through_host=(backend == "gloo"))
    res = nd.mur_sharded(shard, comm, batch=5, **KW)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=res.w, h=res.h, i=res.i, obj=np.asarray(res.obj_history))
    shard.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo")])
def test_sharded_device_path(world, backend, tmp_path):
    import torch.multiprocessing as mp
    from oracle import nmf_ref as R
    mp.spawn(_worker, args=(world, _free_port(), backend, str(tmp_path)), nprocs=world, join=True)
    m, n, k, v, w0, h0 = _case()
    ref = R.mur(v.astype(np.float64), k, w0=w0, h0=h0, **KW)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    w = np.concatenate([p["w"] for p in parts])
    h = parts[0]["h"]
    err = np.linalg.norm(w @ h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    for p in parts:
        assert int(p["i"]) == ref.i
        np.testing.assert_allclose(p["obj"], ref.obj_history, rtol=2e-4)
        np.testing.assert_array_equal(p["h"], h)        # replicated H is bit-identical on all ranks
