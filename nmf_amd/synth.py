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
