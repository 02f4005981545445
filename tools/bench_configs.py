#!/usr/bin/env python3
"""Timing of the non-headline BASELINE.json configs on one MI355X (GPU box).

    python tools/bench_configs.py cfg3 cfg4 [cfg2] [cfg5_1gpu]

cfg3: AO-ADMM Euclidean, reg_w = reg_h = (0.1, 'l1n'), V 16384x8192 f32, k=128, admm_iter=10
cfg4: MUR KL, V 32768x16384 f32, k=64
cfg5_1gpu: MUR Euclidean, V 131072x16384 f32, k=128 on ONE GPU (the 8-GPU config's strong-scaling base)
Prints one JSON line per config: iterations/s, per-kernel mean times, algorithmic
flops per iteration and the achieved TFLOP/s."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402

from nmf_amd.engine import Engine  # noqa: E402
from nmf_amd.synth import planted_matrix  # noqa: E402

NEVER = 10 ** 12
KERNELS = ("wphase", "wphase_noobj", "objective", "hphase", "gram_nt", "gram_tn", "sum_hht", "w_update", "pack",
           "h_update", "images", "row_sums", "prepare", "inner_h", "inner_w", "sums", "kl_vaux", "kl_vaux_fused", "nnls", "small")


def run(name, m, n, k, queue, finish=None, steps=20, warmup=3, init="random", flops=None, note=""):
    t0 = time.time()
    v = planted_matrix(m, n, k, seed=0, dtype=np.float32)
    rs = np.random.RandomState(0)
    if init == "random":
        w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    else:                                   # healthy start for AO-ADMM without a host SVD: planted factors + noise
        w0 = rs.rand(m, k) + 0.01
        h0 = rs.rand(k, n) / k + 0.01
    gen = time.time() - t0
    eng = Engine(m, n, k)
    eng.upload_v(v)
    eng.set_factors(w0, h0)
    queue(eng, 0, warmup)
    eng.synchronize()
    t0 = time.perf_counter()
    queue(eng, warmup, steps)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rule, stop_i, n_obj = eng.state()
    obj = eng.objectives(0, n_obj)
    eng.profile_enable(True)
    eng.profile_reset()
    queue(eng, warmup + steps, 5)
    eng.synchronize()
    prof = {}
    for kn in KERNELS:
        ms, cnt = eng.profile_get(kn)
        if cnt:
            prof[kn] = {"us": round(ms / cnt * 1e3, 1), "per_iter": cnt / 5}
    inner = None
    if "admm" in name.lower():
        inner = (eng.inner_counts(0, warmup + steps) & 0xFFFF).mean(axis=0).tolist()
        note = (note + "; " if note else "") + "inner paths (stood, cut back, continued, continued + cut back) = %s" % (eng.inner_paths(),)
    out = {"config": name, "shape": [m, n, k], "iter_per_s": 1.0 / dt, "ms_per_iter": dt * 1e3,
           "algorithmic_gflop_per_iter": flops / 1e9 if flops else None,
           "tflops": flops / dt / 1e12 if flops else None, "stop_rule": rule,
           "obj_first_last": [float(obj[0]), float(obj[-1])], "mean_inner_rounds_h_w": inner,
           "kernels": prof, "host_gen_s": round(gen, 1), "note": note}
    print(json.dumps(out), flush=True)
    eng.close()


def main():
    want = sys.argv[1:] or ["cfg3", "cfg4"]
    if "cfg2" in want:
        m, n, k = 16384, 8192, 64
        run("cfg2 MUR-eu", m, n, k, lambda e, f, c: e.mur_run(0, 0, 0, NEVER, 1e-5, 1e-5, f, c),
            flops=4.0 * m * n * k + 4.0 * k * k * (m + n), steps=50)
    if "cfg3" in want:
        m, n, k, T = 16384, 8192, 128, 10
        run("cfg3 AO-ADMM-eu l1n", m, n, k,
            lambda e, f, c: e.aoadmm_run(0, 1, 0.1, 1, 0.1, T, NEVER, 1e-3, 1e-3, f, c), init="planted",
            flops=4.0 * m * n * k + 2.0 * k * k * (m + n) * (1 + T) + 2.0 * k ** 3 / 3,
            note="flops counted with T = admm_iter = 10 rounds per sub-problem (SURVEY 8d); see mean_inner_rounds")
    if "cfg4" in want:
        m, n, k = 32768, 16384, 64
        run("cfg4 MUR-kl", m, n, k, lambda e, f, c: e.mur_run(1, 0, 0, NEVER, 1e-5, 1e-5, f, c),
            flops=8.0 * m * n * k, steps=10)
    if "cfg5_1gpu" in want:
        m, n, k = 131072, 16384, 128
        run("cfg5 MUR-eu on ONE GPU", m, n, k, lambda e, f, c: e.mur_run(0, 0, 0, NEVER, 1e-5, 1e-5, f, c),
            flops=4.0 * m * n * k + 4.0 * k * k * (m + n), steps=5, warmup=2)


if __name__ == "__main__":
    main()
