"""The C-ABI library loads on a CPU-only box and exports exactly what
include/nmfx.h declares; the product path refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from nmf_amd import _lib


def header_functions():
    text = open(os.path.join(ROOT, "include", "nmfx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nmfx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in nmfx.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.nmfx_version() >= 100
    # ... and nothing beyond the header (a debug export had slipped out of an experiment build's #ifdef)
    import shutil
    import subprocess
    nm = shutil.which("nm")
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True) if nm else None
    if out is not None and out.returncode == 0:
        exported = sorted(set(re.findall(r"\bT (nmfx_[a-z0-9_]+)$", out.stdout, flags=re.M)))
        assert exported == names, sorted(set(exported) ^ set(names))


def test_error_codes_without_device():
    lib = _lib.load()
    if lib.nmfx_device_count() > 0:
        pytest.skip("GPU present")
    import ctypes as C
    h = C.c_void_p()
    assert lib.nmfx_create(C.byref(h), 0, 8, 8, 2) == _lib.NMFX_E_HIP
    assert lib.nmfx_create(C.byref(h), 0, 8, 8, 5000) == _lib.NMFX_E_ARG
    assert b"k > 4096" in lib.nmfx_last_error(None)
    assert lib.nmfx_create(C.byref(h), 0, 8, 8, 500) == _lib.NMFX_E_HIP          # (k > 128 is a valid request: MUR runs it)


def test_product_path_fails_loudly_without_gpu():
    lib = _lib.load()
    if lib.nmfx_device_count() > 0:
        pytest.skip("GPU present")
    from nmf_amd import NMF
    v = np.random.RandomState(0).rand(16, 12)
    for method, kw in (("mur", dict(distance_type="eu", max_iter=2)),):
        with pytest.raises(RuntimeError, match="no HIP device"):
            NMF(v, 3).factorize(method=method, **kw)


def test_unknown_method_and_distance_raise_like_reference():
    from nmf_amd import NMF
    v = np.random.RandomState(0).rand(16, 12)
    with pytest.raises(Exception, match="Method not known"):      # nmf/nmf.py:76
        NMF(v, 3).factorize(method="nope")
    with pytest.raises(KeyError):                                   # nmf/utils.py:31
        NMF(v, 3).factorize(method="mur", distance_type="xx")


def test_ao_admm_l1inf_raises_like_reference():
    """nmf/ao_admm.py:143-195: the ao_admm copy of the operator breaks a Cholesky factorisation (scipy LinAlgError) for every
    placement within a few outer iterations.  Up to 128 components the device runs the operator until that happens (both losses:
    tests/test_gpu_aoadmm.py, r4 / r5); beyond 128 components the exception is still raised before any device work."""
    from nmf_amd.ao_admm import ao_admm
    v = np.random.RandomState(0).rand(16, 12)
    for kind in ("l1inf", "l1inf_transpose"):
        for loss in ("eu", "kl"):
            with pytest.raises(np.linalg.LinAlgError):
                ao_admm(v, 130, distance_type=loss, reg_w=(0.1, "nn"), reg_h=(0.1, kind), nndsvd_init=(False, "zero"))
            with pytest.raises(np.linalg.LinAlgError):
                ao_admm(v, 130, distance_type=loss, reg_w=(0.1, kind), reg_h=(0.1, "nn"), nndsvd_init=(False, "zero"))
