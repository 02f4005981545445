"""Device top-k SVD (nmfx_topk_svd) and the NNDSVD built on it against numpy's LAPACK SVD
(what nmf/utils.py:50 calls)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
os.environ.setdefault("NMF_AMD_QUIET", "1")


def _align(u, ref):
    """flip the sign of each column of u to match ref"""
    sgn = np.sign(np.sum(u * ref, axis=0))
    sgn[sgn == 0] = 1
    return u * sgn, sgn


CASES = [
    ("planted", 700, 330, 12),       # rank-12 + noise: large gap after the k-th value
    ("uniform", 512, 256, 8),        # np.random.rand: one dominant value, then a gap-less bulk
    ("wide", 200, 900, 5),
    ("f64", 300, 280, 16),
]


@pytest.mark.parametrize("kind,m,n,k", CASES)
def test_topk_svd_matches_lapack(kind, m, n, k):
    from oracle import nmf_ref as R
    from nmf_amd.engine import Engine
    if kind == "planted":
        v = R.planted_matrix(m, n, k, seed=5, dtype=np.float32)
    else:
        v = np.random.RandomState(6).rand(m, n).astype(np.float64 if kind == "f64" else np.float32)
    with Engine(m, n, k) as eng:
        eng.upload_v(v)
        u, s, vt, sweeps, resid = eng.topk_svd(k)
    v32 = v.astype(np.float32).astype(np.float64)          # the engine holds V in f32
    ur, sr, vtr = np.linalg.svd(v32, full_matrices=False)
    assert resid <= 1e-11, (sweeps, resid)
    np.testing.assert_allclose(s, sr[:k], rtol=1e-12)
    ua, sgn = _align(u, ur[:, :k])
    # vector error ~ residual / gap: compare through the subspace-insensitive rank-k reconstruction
    # and, where the spectrum is separated, vector by vector
    rec, rec_ref = (ua * s) @ (vt * sgn[:, None]), (ur[:, :k] * sr[:k]) @ vtr[:k]
    assert np.linalg.norm(rec - rec_ref) / np.linalg.norm(rec_ref) < 1e-9
    gaps = np.minimum(np.abs(np.diff(sr[:k + 1]))[:-1] if k > 1 else np.inf, np.abs(np.diff(sr[:k + 1]))[1:] if k > 1 else np.inf)
    for j in range(k):
        gap = min(abs(sr[j] - sr[j - 1]) if j else np.inf, abs(sr[j] - sr[j + 1]))
        if gap > 1e-3 * sr[0]:
            assert np.max(np.abs(ua[:, j] - ur[:, j])) < 1e-7, j
            assert np.max(np.abs(vt[j] * sgn[j] - vtr[j])) < 1e-7, j
    assert np.max(np.abs(u.T @ u - np.eye(k))) < 1e-10


@pytest.mark.parametrize("variant", ["zero", "mean"])
def test_device_nndsvd_matches_host_nndsvd(variant):
    from oracle import nmf_ref as R
    from nmf_amd import utils
    from nmf_amd.engine import Engine
    m, n, k = 640, 384, 10
    v = R.planted_matrix(m, n, k, seed=9, dtype=np.float32)
    w_ref, h_ref = utils.nndsvd(v.astype(np.float64), k, variant=variant)
    with Engine(m, n, k) as eng:
        eng.upload_v(v)
        w, h = utils.nndsvd_device(eng, v, k, variant=variant)
    np.testing.assert_allclose(w, w_ref, rtol=0, atol=1e-7 * np.max(w_ref))
    np.testing.assert_allclose(h, h_ref, rtol=0, atol=1e-7 * np.max(h_ref))


def test_solver_with_device_nndsvd_matches_oracle(monkeypatch):
    """End to end: ao_admm started from the device NNDSVD equals the oracle started from LAPACK's."""
    from oracle import nmf_ref as R
    from nmf_amd.ao_admm import ao_admm
    monkeypatch.setenv("NMFX_NNDSVD", "device")
    m, n, k = 520, 300, 12
    v = R.planted_matrix(m, n, k, seed=31, dtype=np.float32)
    kw = dict(distance_type="eu", reg_w=(0.1, "l1n"), reg_h=(0.05, "l1n"), min_iter=8, max_iter=8, admm_iter=10,
              nndsvd_init=(True, "zero"))
    ref = R.ao_admm(v.astype(np.float64), k, **kw)
    res = ao_admm(v.copy(), k, **kw)
    err = np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v.astype(np.float64))
    assert err < 1e-4, err
    assert res.i == ref.i
    np.testing.assert_allclose(res.obj_history, ref.obj_history, rtol=1e-5)      # (measured: 9.8e-7)
