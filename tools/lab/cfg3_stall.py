"""A ~70-80 ms device-side gap shows up once per run of the AO-ADMM loop (kernels queued, none of them long; ADMM and
MUR on the same engine never show it).  Where?  The iteration in its sharded phases, synchronised one by one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np
import torch
from nmf_amd.engine import Engine
from nmf_amd import utils
import bench
m, n, k, T = 16384, 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 10
dev = torch.device("cuda:0")
eng = Engine(m, n, k)
bench.device_planted(eng, torch, m, n, k, 0, dev)
class S: shape = (m, n)
u, sv, vt, _, _ = eng.topk_svd(k)
w0, h0 = utils._nndsvd_from_triplets(S, u, sv, vt, k, "zero")
NEVER = 10 ** 12
eng.set_factors(w0, h0)
phases = [("h_products", lambda j: eng.aoadmm_phase_h_products(j)),
          ("h_solve", lambda j: eng.aoadmm_phase_h_solve(1, 0.1, T, NEVER, 1e-3, 1e-3, j)),
          ("w_products", lambda j: eng.aoadmm_phase_w_products(NEVER, 1e-3, 1e-3, j)),
          ("w_fused", lambda j: eng.aoadmm_phase_w_fused(1, 0.1, T)),
          ("w_repair", lambda j: eng.aoadmm_phase_w_repair(1, 0.1, T, j))]
def cpu_stat():
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            return {a: int(b) for a, b in (ln.split() for ln in open(path))}
        except OSError:
            pass
    return {}


import threading
print("threads in this process:", threading.active_count(), "os-level:", len(os.listdir("/proc/self/task")), "cpu.max:",
      open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
c0 = cpu_stat()
slow = []
for j in range(200):
    for name, fn in phases:
        t0 = time.perf_counter()
        fn(j)
        eng.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 5.0:
            slow.append((j, name, round(dt, 1)))
c1 = cpu_stat()
print("k =", k, "slow phases:", slow, "| cgroup cpu.stat delta:", {a: c1[a] - c0.get(a, 0) for a in c1 if c1[a] != c0.get(a, 0)})
