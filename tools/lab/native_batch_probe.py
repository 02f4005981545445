"""Where do the ~0.6 ms go that a 20-iteration batch of the native sharded loop (world of one) takes over 20 x the steady-state step?"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from nmf_amd import dist as nd
from nmf_amd.synth import planted_matrix

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
m, n, k = 16384, 8192, 64
v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
rs = np.random.RandomState(0)
w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
native = os.environ.get("NATIVE", "1") == "1"
shard = (nd.NativeShard if native else nd.DeviceShard)(v, k, w0, h0, 0)
comm = nd.NativeComm.create(shard) if native else nd.TorchComm(stage_through_host=False)
run = nd.Runner(shard, comm, 0, 0.0, 0.0, 10 ** 12, 1e-5, 1e-5, 4000)
eng = shard.eng
def fence():
    eng.synchronize(); torch.cuda.synchronize()
run(0, 200); fence()
j = 200
for batch in (20, 20, 20, 100, 20, 20):
    t0 = time.perf_counter(); run(j, batch); t1 = time.perf_counter(); fence(); t2 = time.perf_counter()
    print(f"native={native} batch {batch:4d}: queued in {1e3 * (t1 - t0):7.3f} ms, done in {1e3 * (t2 - t0):7.3f} ms = {1e6 * (t2 - t0) / batch:7.1f} us/step")
    j += batch
    if os.environ.get("IDLE"): time.sleep(float(os.environ["IDLE"]))
if os.environ.get("BENCHFLOW"):
    for rep in range(3):
        eng.set_factors(w0, h0)
        run(0, 5); fence()
        if os.environ.get("BARRIER"): dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(5, 20); fence(); t2 = time.perf_counter()
        print(f"native={native} bench flow rep {rep}: {1e6 * (t2 - t0) / 20:7.1f} us/step")
