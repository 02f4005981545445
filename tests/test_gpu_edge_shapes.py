"""Odd shapes through every solver on the GPU against the oracle: single rows / columns, k = 1, k equal to a
dimension, shapes straddling the 64 / 128 padding boundaries, k on both sides of the 64 -> 128 rank padding.
Where the reference algorithm itself fails (AO-ADMM from a random start can make the Gram singular,
ao_admm.py:55) both sides must fail the same way."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1), (2, 3, 1), (5, 4, 4), (7, 300, 3), (300, 7, 5), (65, 129, 17), (130, 64, 33), (64, 64, 64),
          (200, 130, 65), (129, 257, 128)]


def _solvers():
    from oracle import nmf_ref as R
    from nmf_amd.admm import admm
    from nmf_amd.anls import anls
    from nmf_amd.ao_admm import ao_admm
    from nmf_amd.mur import mur
    rnd = (False, "zero")
    return {
        "mur-eu": (mur, R.mur, dict(distance_type="eu", min_iter=6, max_iter=6)),
        "mur-kl": (mur, R.mur, dict(distance_type="kl", min_iter=6, max_iter=6)),
        "ao_admm": (ao_admm, R.ao_admm, dict(reg_w=(0.01, "l1n"), reg_h=(0, "nn"), min_iter=4, max_iter=4, nndsvd_init=rnd)),
        "admm": (admm, R.admm, dict(reg_w=(0, "nn"), reg_h=(0.01, "l1n"), min_iter=4, max_iter=4, nndsvd_init=rnd)),
        "anls": (anls, R.anls, dict(min_iter=3, max_iter=3, nndsvd_init=rnd)),
    }


@pytest.mark.parametrize("method", ["mur-eu", "mur-kl", "ao_admm", "admm", "anls"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_odd_shapes_match_oracle(shape, method):
    m, n, k = shape
    fn, ref_fn, kw = _solvers()[method]
    v = np.random.RandomState(m * 7 + n).rand(m, n) + 0.05
    np.random.seed(3)
    try:
        with np.errstate(all="ignore"):
            ref = ref_fn(v.copy(), k, **kw)
    except np.linalg.LinAlgError:
        np.random.seed(3)
        with pytest.raises(np.linalg.LinAlgError):
            fn(v.copy(), k, **kw)
        return
    np.random.seed(3)
    res = fn(v.copy(), k, **kw)
    assert res.i == ref.i and len(res.obj_history) == len(ref.obj_history)
    err = np.linalg.norm(res.w @ res.h - ref.w @ ref.h) / np.linalg.norm(v)
    assert err < 1e-4, err          # north_star tolerance
