"""The on-disk format: file-name grammar of nmf/nmf.py:95-126 for all four methods and the .npz keys of
nmf/utils.py:96-105 -- no GPU needed, the Results are made up.  (The expected names are the strings the reference's f-strings give
for these Experiment tuples; its Experiment field lists are nmf/mur.py:77, nmf/anls.py:81, nmf/admm.py:262, nmf/ao_admm.py:230.)"""
import numpy as np
import pytest

from nmf_amd import NMF
from nmf_amd._driver import Results

CASES = [
    ("mur", dict(method="mur", components=4, distance_type="kl", nndsvd_init=(False, "zero"), max_iter=9, tol1=1e-5, tol2=1e-5,
                 lambda_w=0.0, lambda_h=0.25), "nmf_mur_4_kl_0.0_0.25_random"),
    ("mur", dict(method="mur", components=4, distance_type="eu", nndsvd_init=(True, "mean"), max_iter=9, tol1=1e-5, tol2=1e-5,
                 lambda_w=1, lambda_h=0), "nmf_mur_4_eu_1_0_nndsvdm"),
    ("anls", dict(method="anls", components=4, distance_type="eu", nndsvd_init=(True, "zero"), max_iter=9, tol1=1e-3, tol2=1e-3,
                  lambda_w=0, lambda_h=0.5, fcnnls=True), "nmf_anls_4_eu_0_0.5_nndsvdz_fcnnls"),
    ("anls", dict(method="anls", components=4, distance_type="kl", nndsvd_init=(True, "random"), max_iter=9, tol1=1e-3, tol2=1e-3,
                  lambda_w=0, lambda_h=0, fcnnls=False), "nmf_anls_4_kl_0_0_nndsvdr"),
    ("admm", dict(method="admm", components=4, rho=2.5, distance_type="eu", nndsvd_init=(True, "zero"), min_iter=1, max_iter=9,
                  tol1=1e-3, tol2=1e-3, lambda_w=0, prox_w="nn", lambda_h=0.1, prox_h="l2n"), "nmf_admm_4_eu_2.5_0:nn_0.1:l2n_nndsvdz"),
    ("ao_admm", dict(method="ao_admm", components=4, distance_type="kl", nndsvd_init=(False, "zero"), min_iter=1, max_iter=9,
                     admm_iter=10, tol1=1e-3, tol2=1e-3, lambda_w=0.5, prox_w="l1n", lambda_h=0, prox_h="nn"),
     "nmf_ao_admm_4_kl_0.5:l1n_0:nn_random"),
]


@pytest.mark.parametrize("method,fields,name", CASES)
def test_default_save_name_and_npz_keys(method, fields, name, tmp_path):
    import importlib
    Experiment = importlib.import_module("nmf_amd." + method).Experiment
    exp = Experiment(**fields)                        # (a TypeError here = the field list differs from the reference's)
    model = NMF(np.ones((6, 5)), 4)
    model.results = Results(w=np.ones((6, 4)), h=np.ones((4, 5)), i=8, obj_history=[3.0, 2.0, 1.0], experiment=exp)
    model.save_factorization(save_dir=str(tmp_path))
    files = [p.name for p in tmp_path.iterdir()]
    assert files == [name + ".npz"], files
    z = np.load(tmp_path / files[0], allow_pickle=True)
    assert sorted(z.files) == ["experiment", "h", "i", "obj_history", "w"]
    assert z["experiment"].item() == exp._asdict() and int(z["i"]) == 8
    model.save_factorization(save_dir=str(tmp_path), save_name="mine")
    assert (tmp_path / "mine.npz").exists()
