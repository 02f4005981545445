"""The environment knobs that select an alternative kernel path (read once per process, hence child
processes): every path must keep the parity bar, ||W H - W_ref H_ref|| / ||V|| < 1e-4 against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
spec = json.loads(sys.argv[1])
m, n, k = spec["shape"]
v = R.planted_matrix(m, n, min(k, 32), seed=m + n + k, dtype=np.float32)
import nmf_amd.mur, nmf_amd.ao_admm, nmf_amd.admm, nmf_amd.anls
fn = {"mur": nmf_amd.mur.mur, "ao_admm": nmf_amd.ao_admm.ao_admm, "admm": nmf_amd.admm.admm, "anls": nmf_amd.anls.anls}[spec["method"]]
ref_fn = {"mur": R.mur, "ao_admm": R.ao_admm, "admm": R.admm, "anls": R.anls}[spec["method"]]
kw = {key: (tuple(val) if isinstance(val, list) else val) for key, val in spec["kwargs"].items()}
np.random.seed(5); res = fn(v.copy(), k, **kw)
np.random.seed(5)
with np.errstate(all="ignore"):
    ref = ref_fn(v.astype(np.float64), k, **kw)
err = float(np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)))
print(json.dumps({"err": err, "i": int(res.i), "ref_i": int(ref.i)}))
'''

NNDSVD = [True, "zero"]
CASES = [
    # (environment, method, shape, kwargs)
    ({"NMFX_BF16_TERMS": "4"}, "mur", (384, 256, 40), dict(distance_type="eu", min_iter=15, max_iter=15)),
    ({"NMFX_BF16_TERMS": "4"}, "mur", (384, 256, 40), dict(distance_type="kl", min_iter=15, max_iter=15)),
    ({"NMFX_PRECISION": "f32"}, "mur", (384, 256, 40), dict(distance_type="eu", min_iter=15, max_iter=15)),
    ({"NMFX_XYT16": "1"}, "mur", (384, 256, 40), dict(distance_type="eu", min_iter=15, max_iter=15)),      # the 16-row form of the k = 64 product kernel
    ({"NMFX_DROP_V": "1"}, "anls", (320, 256, 40), dict(distance_type="kl", min_iter=4, max_iter=4, nndsvd_init=NNDSVD)),   # row-major V dropped, the KL objective pass brings it back
    ({"NMFX_NNLS128_OCC": "1"}, "anls", (260, 400, 100), dict(min_iter=3, max_iter=3, lambda_w=0.05, lambda_h=0.02, nndsvd_init=NNDSVD)),
    ({"NMFX_PREPARE_SCALAR": "1"}, "ao_admm", (384, 320, 100), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=6, max_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_AO_FUSED": "0"}, "ao_admm", (384, 320, 100), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=6, max_iter=6, nndsvd_init=NNDSVD)),
    # r4: the inversions as side jobs of stream-K products are the default at k padded to 128; the launches of round 3, plain (not
    # cyclic) runs, and a worker count that makes every run cross several row blocks (segments per worker > 2)
    ({"NMFX_AO_OVERLAP": "0"}, "ao_admm", (384, 320, 100), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=6, max_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_SK_CYCLIC": "0"}, "ao_admm", (384, 320, 100), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=6, max_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_SK_WORKERS": "2"}, "ao_admm", (640, 512, 100), dict(reg_w=[0.05, "l1n"], reg_h=[0, "nn"], min_iter=6, max_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_SK_WORKERS": "7"}, "ao_admm", (640, 512, 100), dict(reg_w=[0, "nn"], reg_h=[0.05, "l1n"], min_iter=6, max_iter=6, nndsvd_init=NNDSVD)),
    # r4: the KL-loss variants run their V-sized products, auxiliaries and objective on the split-bf16 kernels; the exact-f32 launches
    ({"NMFX_KL_BF16": "0"}, "ao_admm", (384, 320, 40), dict(distance_type="kl", reg_w=[0, "nn"], reg_h=[0.02, "l1n"], min_iter=4, max_iter=4, admm_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_KL_BF16": "0"}, "admm", (384, 320, 100), dict(rho=1.0, distance_type="kl", reg_w=[0, "nn"], reg_h=[0, "nn"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),
    # r5: the auxiliaries launch of a round also forms the next round's right-hand-side product; the separate launches
    ({"NMFX_KL_FUSE": "0"}, "ao_admm", (384, 320, 100), dict(distance_type="kl", reg_w=[0, "nn"], reg_h=[0.02, "l1n"], min_iter=4, max_iter=4, admm_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_KL_FUSE": "0"}, "ao_admm", (384, 320, 40), dict(distance_type="kl", reg_w=[0.02, "l1n"], reg_h=[0, "nn"], min_iter=4, max_iter=4, admm_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_KL_GATHER": "0"}, "admm", (384, 320, 100), dict(rho=1.0, distance_type="kl", reg_w=[0, "nn"], reg_h=[0, "nn"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),
    ({"NMFX_AO_ROWS_RB": "128"}, "ao_admm", (384, 320, 100), dict(reg_w=[0, "nn"], reg_h=[0, "nn"], min_iter=6, max_iter=6, nndsvd_init=NNDSVD)),
    ({"NMFX_NNLS_LDS": "1"}, "anls", (320, 256, 40), dict(min_iter=4, max_iter=4, nndsvd_init=NNDSVD)),
    ({"NMFX_NNLS_CINV": "0"}, "anls", (320, 256, 40), dict(min_iter=4, max_iter=4, nndsvd_init=NNDSVD)),      # elimination kernels only
    ({"NMFX_NNLS_CINV": "0"}, "anls", (260, 400, 100), dict(min_iter=3, max_iter=3, lambda_w=0.05, lambda_h=0.02, nndsvd_init=NNDSVD)),
    ({"NMFX_NW4": "3"}, "mur", (384, 256, 40), dict(distance_type="eu", min_iter=15, max_iter=15)),        # four-wave blocks, both phases
    ({"NMFX_NW4": "1"}, "mur", (640, 384, 64), dict(distance_type="eu", lambda_w=0.1, lambda_h=0.2, min_iter=12, max_iter=12)),
    # beyond 128 components: the exact-f32 product kernel and the one-workgroup Gram inversion behind the split-bf16 / blocked defaults
    ({"NMFX_PRECISION": "f32"}, "mur", (384, 256, 160), dict(distance_type="eu", min_iter=10, max_iter=10)),
    ({"NMFX_PRECISION": "f32"}, "mur", (384, 256, 160), dict(distance_type="kl", min_iter=10, max_iter=10)),
    ({"NMFX_PRECISION": "f32"}, "ao_admm", (384, 320, 160), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),
    ({"NMFX_PREPARE_SCALAR": "1"}, "ao_admm", (384, 320, 160), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),
    ({"NMFX_PRECISION": "f32"}, "admm", (384, 320, 160), dict(rho=1.0, reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),
    # the bf16 operand planes "do not fit": the handle falls back to the exact-f32 product kernel (and says so in nmfx_get_note)
    # r4: the composed path's alternatives -- short contractions on gxb_gemm_kernel, 256 x 128 tiles everywhere, separate update / image launches,
    # no wave stagger / DMA pieces between the MFMA groups (512 x 512 pads to whole 256 x 256 tiles, so the default run takes gxt2_gemm_kernel)
    ({"NMFX_GXR": "0"}, "mur", (384, 256, 160), dict(distance_type="eu", min_iter=10, max_iter=10)),
    ({"NMFX_GXR": "0"}, "mur", (384, 256, 160), dict(distance_type="kl", min_iter=10, max_iter=10)),
    ({"NMFX_GXT2": "0"}, "mur", (512, 512, 160), dict(distance_type="eu", min_iter=10, max_iter=10)),
    ({"NMFX_GXT2": "0"}, "mur", (512, 512, 160), dict(distance_type="kl", min_iter=10, max_iter=10)),
    ({"NMFX_GXT_NT": "0"}, "mur", (1024, 768, 160), dict(distance_type="eu", lambda_w=0.1, lambda_h=0.05, min_iter=8, max_iter=8)),
    ({"NMFX_GX_FUSE_UPDATE": "0"}, "mur", (512, 512, 160), dict(distance_type="eu", lambda_w=0.1, lambda_h=0.05, min_iter=10, max_iter=10)),
    ({"NMFX_GX_DEN_BF16": "0"}, "mur", (512, 256, 400), dict(distance_type="eu", lambda_w=0.05, lambda_h=0.02, min_iter=8, max_iter=8)),   # k pads to 512: exact-f32 denominator W (H H^T)
    ({"NMFX_GX_STAGGER": "0"}, "mur", (384, 256, 160), dict(distance_type="kl", min_iter=10, max_iter=10)),
    ({"NMFX_GX_STAGGER": "16"}, "mur", (384, 256, 160), dict(distance_type="eu", min_iter=10, max_iter=10)),
    ({"NMFX_GXR": "0"}, "ao_admm", (384, 320, 160), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),
    ({"NMFX_GX_ANLS_BF16": "0"}, "anls", (256, 192, 144), dict(min_iter=2, max_iter=2, lambda_w=0.05, lambda_h=0.02, nndsvd_init=NNDSVD)),   # ANLS beyond 128 on the exact-f32 products
    ({"NMFX_GX_ROUNDS_F32": "1"}, "ao_admm", (384, 320, 160), dict(reg_w=[0.05, "l1n"], reg_h=[0, "nn"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),   # exact-f32 inner products of the any-rank rounds
    ({"NMFX_GXB_NOFIT": "1"}, "mur", (384, 256, 160), dict(distance_type="kl", min_iter=10, max_iter=10)),
    ({"NMFX_GXB_NOFIT": "1"}, "ao_admm", (384, 320, 160), dict(reg_w=[0.05, "l1n"], reg_h=[0.05, "l1n"], min_iter=5, max_iter=5, nndsvd_init=NNDSVD)),
]


@pytest.mark.parametrize("env,method,shape,kwargs", CASES, ids=[f"{list(c[0].items())[0][0]}={list(c[0].items())[0][1]}-{c[1]}-{c[3].get('distance_type', '')}" for c in CASES])
def test_alternative_paths_keep_parity(env, method, shape, kwargs):
    spec = json.dumps({"method": method, "shape": list(shape), "kwargs": kwargs})
    # (NMF_AMD_NO_TORCH: the child never touches torch; its import is 1.5 s of each of these fifty processes)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, spec], env=dict(os.environ, NMF_AMD_NO_TORCH="1", **env),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["i"] == got["ref_i"]
    assert got["err"] < 1e-4, got


def test_two_handles_from_two_threads_and_dropped_row_major_v(monkeypatch):
    """Independent handles are thread safe (include/nmfx.h): two factorisations with different padded ranks run
    concurrently from two threads -- the per-device bookkeeping of the kernels' LDS limits is shared state -- and give
    what they give alone.  One of them also runs with the row-major V dropped after the tile-major copies were built
    (NMFX_DROP_V=1 at create time) and later needs it back (device SVD for an NNDSVD start)."""
    import threading
    from nmf_amd.engine import Engine
    from nmf_amd import utils as U
    from oracle import nmf_ref as R
    out, errs = {}, []

    def work(tag, m, n, k, seed):
        try:
            v = R.planted_matrix(m, n, k, seed=seed, dtype=np.float32)
            rs = np.random.RandomState(seed)
            w0, h0 = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
            with Engine(m, n, k) as eng:
                eng.upload_v(v)
                eng.set_factors(w0, h0)
                eng.mur_run(0, 0.0, 0.0, 10 ** 9, 1e-5, 1e-5, 0, 12)
                eng.mur_finish(0, 10 ** 9, 1e-5, 1e-5, 12)
                w, h = eng.get_factors()
                s = eng.topk_svd(4)[1] if tag == "b" else None        # needs the row-major V again
            ref = R.mur(v.astype(np.float64), k, distance_type="eu", min_iter=12, max_iter=12, w0=w0, h0=h0)
            out[tag] = (float(np.linalg.norm(w @ h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))), s,
                        np.linalg.svd(v.astype(np.float64), compute_uv=False)[:4])
        except Exception as e:  # noqa: BLE001
            errs.append((tag, repr(e)))

    monkeypatch.setenv("NMFX_DROP_V", "1")
    ts = [threading.Thread(target=work, args=("a", 384, 256, 40, 1)), threading.Thread(target=work, args=("b", 256, 384, 100, 2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert out["a"][0] < 1e-4 and out["b"][0] < 1e-4, out
    np.testing.assert_allclose(out["b"][1], out["b"][2], rtol=1e-9)


HINT_CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.ao_admm import ao_admm
m, n, k, T, it = (int(a) for a in sys.argv[1:6])
v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
res = ao_admm(v.copy(), k, distance_type="eu", reg_w=(0.02, "l1n"), reg_h=(0, "nn"), min_iter=it, max_iter=it, admm_iter=T,
              nndsvd_init=(True, "zero"))
np.savez(sys.argv[6], w=res.w, h=res.h, obj=np.asarray(res.obj_history), inner=np.asarray(ao_admm.last_inner_counts),
         paths=np.asarray(ao_admm.last_inner_paths))
'''


@pytest.mark.parametrize("shape,admm_iter,iters,expect", [
    ((320, 448, 100), 16, 14, (1, 3)),      # first legs cut back, and continued + cut back (the count of rounds creeps up again)
    ((320, 448, 100), 8, 14, (1, 2)),       # ... and continued to admm_iter with nothing to cut
    ((300, 520, 24), 16, 30, (1, 3)),       # k padded to 32: the f32-MFMA form of the round kernels
])
def test_hinted_inner_rounds_are_the_unhinted_ones(shape, admm_iter, iters, expect, tmp_path):
    """The first launch of a fused sub-problem runs only as many rounds as the side's previous sub-problem counted
    (DevState::ao_hint).  Whatever path follows -- the leg stands, is cut back, is continued, is continued and cut back --
    the rounds that count are the reference's: inner counts equal to the oracle's, and factors, objectives and counts
    bit-identical to the run without the hint (NMFX_AO_HINT=0: all admm_iter rounds, then the cut)."""
    from oracle import nmf_ref as R
    m, n, k = shape
    got = {}
    for mode in ("1", "0"):
        out = str(tmp_path / f"hint{mode}.npz")
        run = subprocess.run([sys.executable, "-c", HINT_CHILD % {"root": ROOT}, str(m), str(n), str(k), str(admm_iter), str(iters), out],
                             env=dict(os.environ, NMFX_AO_HINT=mode), capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        got[mode] = np.load(out)
    for key in ("w", "h", "obj", "inner"):
        np.testing.assert_array_equal(got["1"][key], got["0"][key], err_msg=key)
    paths = got["1"]["paths"]
    assert paths[0] > 0 and all(paths[i] > 0 for i in expect), paths
    assert got["0"]["paths"][2] == 0 and got["0"]["paths"][3] == 0                     # (without the hint nothing is ever continued)
    v = R.planted_matrix(m, n, k, seed=m + k, dtype=np.float32)
    ref = R.ao_admm(v.astype(np.float64), k, distance_type="eu", reg_w=(0.02, "l1n"), reg_h=(0, "nn"), min_iter=iters, max_iter=iters,
                    admm_iter=admm_iter, nndsvd_init=(True, "zero"))
    assert [tuple(r) for r in got["1"]["inner"]] == [tuple(t) for t in ref.trace["inner"]]
    # (H is not regularised here so that the counts move: the sub-problems are as ill-conditioned as W^T W.  Round 2 had this case at
    # 1.15e-4: the inner product aux = M^-1 rhs ran on two bf16 images per operand (16 bits), and the cancellation in that product
    # amplifies the operand error by cond(G + rho I) <= k + 1.  With three images and six terms (f32-grade; kernels_aoadmm.hip,
    # modelled on the CPU by tools/lab/ao_f32_state.py) it is 5e-6 in every product mode.)
    err = float(np.linalg.norm(got["1"]["w"] @ got["1"]["h"] - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)))
    assert err < 1e-4, err


@pytest.mark.parametrize("shape,admm_iter,iters", [((320, 448, 100), 10, 8), ((384, 256, 40), 6, 10)])
def test_images_left_by_the_fused_rounds_are_those_of_the_images_launch(shape, admm_iter, iters, tmp_path):
    """r3: in a split-bf16 AO-ADMM run the fused round kernels leave the bf16 hi / lo images of the factor they update (both
    layouts) instead of a separate `images` launch per sub-problem (NMFX_AO_IMG=0 keeps the launches): same values, same split,
    hence factors, objectives and inner counts bit for bit -- also when a first leg is cut back or continued (the hint paths)."""
    m, n, k = shape
    got = {}
    for mode in ("1", "0"):
        out = str(tmp_path / f"img{mode}.npz")
        run = subprocess.run([sys.executable, "-c", HINT_CHILD % {"root": ROOT}, str(m), str(n), str(k), str(admm_iter), str(iters), out],
                             env=dict(os.environ, NMFX_AO_IMG=mode), capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        got[mode] = np.load(out)
    for key in ("w", "h", "obj", "inner"):
        np.testing.assert_array_equal(got["1"][key], got["0"][key], err_msg=key)


@pytest.mark.parametrize("solver", ["mur", "ao_admm", "anls"])
def test_refereed_stop_equals_the_oracles_stop_index(solver, monkeypatch):
    """NMFX_VERIFY_STOP=1 forces the float64 referee of the stop rule (nmf_amd._driver.Referee) on small problems, where the plain
    rule is right anyway: the guarded candidate + one-iteration-at-a-time walk with nmfx_objective_f64 must stop where the oracle
    stops (nmf/utils.py:4-15 through nmf/mur.py:131 / ao_admm.py:298 / anls.py:121), with the history and factors of that iterate."""
    from oracle import nmf_ref as R
    monkeypatch.setenv("NMFX_VERIFY_STOP", "1")
    m, n = 300, 260
    if solver == "mur":
        from nmf_amd.mur import mur as fn
        k, kw, ref_fn = 36, dict(distance_type="eu", min_iter=5, max_iter=3000, tol1=1e-9, tol2=5e-3), R.mur
    elif solver == "ao_admm":
        from nmf_amd.ao_admm import ao_admm as fn
        k, kw, ref_fn = 12, dict(reg_w=(0.05, "l1n"), reg_h=(0.02, "l1n"), min_iter=2, max_iter=400, tol1=1e-9, tol2=2e-3, admm_iter=6,
                                 nndsvd_init=(True, "zero")), R.ao_admm
    else:
        from nmf_amd.anls import anls as fn
        k, kw, ref_fn = 6, dict(lambda_w=0.05, lambda_h=0.02, min_iter=2, max_iter=300, tol1=1e-9, tol2=1e-4, nndsvd_init=(True, "zero")), R.anls
    v = R.planted_matrix(m, n, k, seed=312 if solver == "ao_admm" else 77, dtype=np.float32)      # (runs longer than the first batch of 64)
    np.random.seed(3)
    res = fn(v.copy(), k, **kw)
    np.random.seed(3)
    ref = ref_fn(v.astype(np.float64), k, **kw)
    rf = fn.last_referee
    assert rf.guard > 0 and rf.walked > 0, (rf.guard, rf.walked)
    assert res.i == ref.i and len(res.obj_history) == res.i + 2, (res.i, ref.i)
    assert np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)) < 1e-4
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=1e-4)


def test_refereed_stop_at_the_first_tested_index(monkeypatch):
    """a run that has converged before min_iter stops at loop index min_iter + 1 in the reference (nmf/mur.py:131).  With the
    referee armed (guard > 0) the device's candidate IS that index; the walk can only test indices behind a candidate, so it used to
    return min_iter + 2 with one history entry too many.  The recorded pair now confirms the candidate (Referee.confirms)."""
    from oracle import nmf_ref as R
    from nmf_amd.mur import mur
    monkeypatch.setenv("NMFX_VERIFY_STOP", "1")
    m, n, k = 300, 260, 36
    v = R.planted_matrix(m, n, k, seed=77, dtype=np.float32)
    kw = dict(distance_type="eu", max_iter=3000, tol1=1e-9, tol2=5e-3)
    np.random.seed(3)
    i0 = R.mur(v.astype(np.float64), k, min_iter=5, **kw).i
    min_iter = max(i0 + 40, 80)                      # (beyond the first batch of 64: the guard is armed when the first test comes)
    np.random.seed(3)
    ref = R.mur(v.astype(np.float64), k, min_iter=min_iter, **kw)
    np.random.seed(3)
    res = mur(v.copy(), k, min_iter=min_iter, **kw)
    assert ref.i == min_iter + 1
    rf = mur.last_referee
    assert rf.guard > 0 and rf.confirmed == 1 and rf.walked == 0, (rf.guard, rf.confirmed, rf.walked)
    assert res.i == ref.i and len(res.obj_history) == ref.i + 2, (res.i, ref.i, len(res.obj_history))
    assert np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64)) < 1e-4


CHILD_NOTPD = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.ao_admm import ao_admm
m, n, k = 384, 320, 100
v = R.planted_matrix(m, n, 32, seed=11, dtype=np.float32)
kw = dict(distance_type="eu", reg_w=(0, "nn"), reg_h=(1e6, "l1n"), admm_iter=10, nndsvd_init=(True, "zero"))
out = {}
for who, fn, x in (("ref", R.ao_admm, v.astype(np.float64)), ("dev", ao_admm, v)):
    raised_at = None
    for t in (1, 2):
        try:
            with np.errstate(all="ignore"):
                fn(x.copy(), k, min_iter=t, max_iter=t, **kw)
        except np.linalg.LinAlgError:
            raised_at = t
            break
    out[who] = raised_at
print(json.dumps(out))
'''


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_aoadmm_k128_bf16_w_side_not_positive_definite_raises_in_the_reference_iteration(overlap):
    """a huge l1n lambda wipes H out inside the first H sub-problem, so H H^T + rho I = 0 and the W side's Cholesky
    (nmf/ao_admm.py:55) raises in that SAME outer iteration -- also when the inversion runs as the side job of the stream-K product
    (the default at k padded to 128, split bf16, admm_iter >= 2), where no objective or stop rule follows the W side."""
    out = subprocess.run([sys.executable, "-c", CHILD_NOTPD % {"root": ROOT}], env=dict(os.environ, NMFX_AO_OVERLAP=overlap, NMFX_PRECISION="bf16", NMF_AMD_NO_TORCH="1"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["ref"] == 1 and got["dev"] == got["ref"], got


CHILD_KLFUSE = r'''
import json, os, sys, hashlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
os.environ["NMF_AMD_QUIET"] = "1"
import numpy as np
from oracle import nmf_ref as R
from nmf_amd.ao_admm import ao_admm
from nmf_amd.admm import admm
out = []
for (m, n, k, seed, kw) in [
        (384, 320, 100, 1, dict(reg_w=(0, "nn"), reg_h=(0.02, "l1n"), min_iter=8, max_iter=8, admm_iter=10)),
        (256, 640, 128, 2, dict(reg_w=(0.05, "l1n"), reg_h=(0, "nn"), min_iter=5, max_iter=5, admm_iter=3)),
        (130, 200, 70, 3, dict(reg_w=(0, "nn"), reg_h=(0, "nn"), min_iter=3, max_iter=3, admm_iter=1)),
        (320, 256, 40, 4, dict(reg_w=(0.05, "l1n"), reg_h=(0, "nn"), min_iter=6, max_iter=6, admm_iter=8)),      # k padded to 64
        (192, 448, 64, 5, dict(reg_w=(0, "nn"), reg_h=(0.02, "l1n"), min_iter=4, max_iter=4, admm_iter=2)),
        (200, 520, 90, 6, dict(rho=1.0, reg_w=(0, "nn"), reg_h=(0.02, "l1n"), min_iter=6, max_iter=6)),           # no admm_iter: admm.py's loop
        (448, 192, 30, 7, dict(rho=2.0, reg_w=(0.05, "l1n"), reg_h=(0, "nn"), min_iter=5, max_iter=5))]:
    v = R.planted_matrix(m, n, 24, seed=seed, dtype=np.float32)
    fn = ao_admm if "admm_iter" in kw else admm
    res = fn(v.copy(), k, distance_type="kl", nndsvd_init=(True, "zero"), **kw)
    out.append({"w": hashlib.sha1(np.ascontiguousarray(res.w).tobytes()).hexdigest(), "h": hashlib.sha1(np.ascontiguousarray(res.h).tobytes()).hexdigest(),
                "obj": [float(x) for x in res.obj_history], "inner": [list(map(int, t)) for t in ao_admm.last_inner_counts] if fn is ao_admm else []})
print(json.dumps(out))
'''


def test_kl_admm_fused_auxiliaries_and_gathered_products_equal_the_separate_launches_bit_for_bit():
    """r5.  (a) AO-ADMM with the KL loss, k padded to 64 or 128: `xyt32_bf16_kernel<..., VAUXF>` forms v_aux / dual_v of round r and, from S in
    registers, the right-hand-side product of round r + 1 (nmf/ao_admm.py:85-95); S is stored only in a sub-problem's last round -- the
    admm_iter-th, or the one whose `terminate` (ao_admm.py:97) fires, which the launch finds out itself.  (b) Both KL-loss ADMM variants: the
    first product of a sub-problem reads S from the other orientation's buffer through transposing requests instead of from a transposed copy;
    ADMM (nmf/admm.py:303-314) runs its auxiliaries in the orientation of V^T, where their launch also forms w_aux^T S of the next iteration.  Same grids, same operand values, same order of additions: factors, objective
    history and inner counts equal those of the separate launches (NMFX_KL_FUSE=0) and of the transposed copies (NMFX_KL_GATHER=0) bit for
    bit -- with inner stops (the first case: counts below admm_iter), with admm_iter reached every time, with a single round per sub-problem."""
    runs = {}
    for tag, env in (("default", {}), ("separate", {"NMFX_KL_FUSE": "0"}), ("copies", {"NMFX_KL_GATHER": "0"}),
                     ("r4", {"NMFX_KL_FUSE": "0", "NMFX_KL_GATHER": "0"})):
        out = subprocess.run([sys.executable, "-c", CHILD_KLFUSE % {"root": ROOT}], env=dict(os.environ, NMF_AMD_NO_TORCH="1", **env),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        runs[tag] = json.loads(out.stdout.strip().splitlines()[-1])
    ao = [i for i, r in enumerate(runs["default"]) if r["inner"]]          # the AO-ADMM cases; ADMM's auxiliaries change orientation with
    for tag in ("separate", "copies", "r4"):                                # NMFX_KL_GATHER (another order of the three terms of w_aux h_aux)
        same = range(len(runs["default"])) if tag == "separate" else ao
        for i in same:
            assert runs[tag][i] == runs["default"][i], (tag, i)
        for i in set(range(len(runs["default"]))) - set(same):
            np.testing.assert_allclose(runs[tag][i]["obj"], runs["default"][i]["obj"], rtol=2e-4)     # (measured 4e-5: the KL objective near its optimum is 1e-5 of sum V)
    inner = [c for t in runs["default"][0]["inner"] for c in t]
    assert min(inner) < 10 <= max(inner), inner           # (the case meant to stop inside its rounds does, and not always)
    assert all(np.isfinite(r["obj"]).all() for r in runs["default"])
