"""Synthetic inputs of the benchmark configurations (BASELINE.json / SURVEY 8d): a planted low-rank
matrix plus uniform noise, generated block-row-wise so that a row shard can be drawn without
materialising the whole matrix (every rank of a sharded run draws the same global stream)."""
import numpy as np


def planted_matrix(m, n, k, seed=0, dtype=np.float32, noise=0.01, rows=None):
    """V = (U1 @ U2)/k + noise*U3 with U* ~ U(0,1) from RandomState(seed).
    `rows=(r0, r1)` returns only that row block (same values as the full
    matrix) -- the stream is drawn as U1 (m*k), U2 (k*n) then U3 row by row."""
    rs = np.random.RandomState(seed)
    left = rs.rand(m, k)
    right = rs.rand(k, n)
    r0, r1 = (0, m) if rows is None else rows
    out = np.empty((r1 - r0, n), dtype=dtype)
    step = 2048
    for a in range(0, m, step):
        b = min(m, a + step)
        nz = rs.rand(b - a, n)
        lo, hi = max(a, r0), min(b, r1)
        if lo < hi:
            blk = (left[lo:hi] @ right) / k + noise * nz[lo - a:hi - a]
            out[lo - r0:hi - r0] = blk.astype(dtype)
    return out


def usable_cpus():
    """CPUs this process may actually use: os.cpu_count() capped by the cgroup's CPU quota (cpu.max).  On a box that
    shows 256 CPUs to a container with a 16-CPU quota, a 256-thread BLAS pool burns the quota in spinning idle threads
    (OpenBLAS workers busy-wait ~100 ms after every call) and the kernel then throttles EVERY thread of the process for
    the rest of the 100 ms period -- seen as one 70-80 ms hole in a GPU loop that had been queued right after a numpy
    call (cpu.stat: nr_throttled + 1).  bench.py and the tests size their BLAS pools with this."""
    import os
    n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return n


def limit_blas_threads(n=None):
    """Cap the BLAS / OpenMP pools of this process at `n` (default usable_cpus()); returns the count in effect."""
    n = n or usable_cpus()
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=n)
    except Exception:  # noqa: BLE001  (threadpoolctl missing: leave the pools alone)
        pass
    return n
