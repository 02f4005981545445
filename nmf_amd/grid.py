"""Parameter grids with V resident on the device.

The reference's author ran grids over the number of components and the regularisation weights by
calling the solver once per combination (legacy driver nmf/nmf_old.py:52-110: `itertools.product`
over `features`, `lambda_w`, `lambda_h` -- and `rho` for ADMM).  Every call there starts from the
host array again.  Here V is uploaded once per value of `features` (the engine's padded rank fixes
its buffers) and stays in HBM, tile-major copies included, for all combinations of the other
parameters; the per-combination cost is the initial factors and the iterations only.

    from nmf_amd.grid import factorize_grid
    runs = factorize_grid(v, 'ao_admm', features=(8, 16), lambda_w=(0, 0.1), lambda_h=(0, 0.1),
                          prox_w='l1n', prox_h='l1n', max_iter=200)
    for params, results in runs: ...

Order of the runs, keyword names and defaults are those of the legacy driver; each run draws from
the global numpy RNG exactly as a separate call of the solver would."""
import os
from importlib import import_module
from itertools import product

import numpy as np

from .engine import Engine
from . import utils

_METHODS = ('mur', 'anls', 'admm', 'ao_admm')


def factorize_grid(data, method='mur', *, features, lambda_w=(0.0,), lambda_h=(0.0,), rho=(1,),
                   prox_w='nn', prox_h='nn', save_dir=None, device=0, **common):
    """Run `method` for every (features, [rho,] lambda_w, lambda_h) of the grid; returns a list of
    (params dict, Results).  `common` holds the keywords shared by all runs (distance_type,
    min_iter, max_iter, tol1, tol2, nndsvd_init, admm_iter, use_fcnnls ...).  With `save_dir` every
    result is written like NMF.save_factorization would."""
    if method not in _METHODS:
        raise Exception('Method not known. Choose one from: mur anls admm ao_admm')
    solver = getattr(import_module('.' + method, __package__), method)
    if method == 'mur':                         # the lift of negative data happens once, in place (nmf/mur.py:99-101)
        lowest = np.min(data)
        if lowest < 0:
            data += abs(lowest)
    out = []
    for k in features:
        with Engine(data.shape[0], data.shape[1], k, device=device) as eng:
            eng.upload_v(data)
            combos = product(rho, lambda_w, lambda_h) if method == 'admm' else product((None,), lambda_w, lambda_h)
            for r, lw, lh in combos:
                kw = dict(common)
                if method in ('mur', 'anls'):
                    kw.update(lambda_w=lw, lambda_h=lh)
                else:
                    kw.update(reg_w=(lw, prox_w), reg_h=(lh, prox_h))
                if method == 'admm':
                    kw.update(rho=r)
                res = solver(data, k, device=device, engine=eng, **kw)
                params = dict(features=k, lambda_w=lw, lambda_h=lh)
                if method == 'admm':
                    params['rho'] = r
                out.append((params, res))
                if save_dir is not None:
                    from .nmf import NMF
                    holder = NMF(data, k)
                    holder.results = res
                    holder.save_factorization(save_dir=save_dir)
    return out
