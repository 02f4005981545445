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
import contextlib
import inspect
import io
import os
import sys
from importlib import import_module
from itertools import product

import numpy as np

from .engine import Engine
from . import utils
from ._driver import Referee

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
    pairs = _pairable(method, common) and os.environ.get("NMFX_GRID_PAIR", "1") != "0"
    if pairs:
        return _mur_eu_grid_in_pairs(data, solver, features, lambda_w, lambda_h, save_dir, device, common)
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


def _pairable(method, common):
    return method == 'mur' and common.get('distance_type', 'kl') == 'eu'


def _save(data, k, res, save_dir):
    from .nmf import NMF
    holder = NMF(data, k)
    holder.results = res
    holder.save_factorization(save_dir=save_dir)


def _mur_eu_grid_in_pairs(data, solver, features, lambda_w, lambda_h, save_dir, device, common):
    """MUR with the Euclidean loss: the combinations are taken two at a time, in the legacy driver's order, and each pair is ONE
    run over V (nmf_amd.mur.mur_pair: two problems of k <= 64 in the halves of a k = 128 engine that stays resident for the whole
    grid); combinations with k > 64, a combination left over, and engines on the exact-f32 path run as before.  Results, printed
    lines and RNG consumption are those of the sequential grid."""
    from .mur import mur_pair
    defaults = {name: par.default for name, par in inspect.signature(solver).parameters.items() if par.default is not inspect.Parameter.empty}
    combos = [(k, lw, lh) for k in features for lw, lh in product(lambda_w, lambda_h)]
    out = [None] * len(combos)
    small = [i for i, c in enumerate(combos) if c[0] <= 64]
    pair_kw = {key: val for key, val in common.items() if key != 'distance_type'}
    singles = {}                                               # k -> resident engine for what cannot be paired

    def run_single(i):
        k, lw, lh = combos[i]
        if k not in singles:
            singles[k] = Engine(data.shape[0], data.shape[1], k, device=device)
            singles[k].upload_v(data)
        return solver(data, k, device=device, engine=singles[k], lambda_w=lw, lambda_h=lh, **common)

    big = None
    try:
        if len(small) >= 2:
            big = Engine(data.shape[0], data.shape[1], 128, device=device)
            big.upload_v(data)
            if big.precision() != 'bf16':
                big.close()
                big = None
        # walk the combinations in order; RNG draws happen in that order too
        i = 0
        while i < len(combos):
            nxt = i + 1
            if big is not None and combos[i][0] <= 64 and nxt < len(combos) and combos[nxt][0] <= 64:
                a, b = combos[i], combos[nxt]
                rng_state = np.random.get_state()
                # (the pair's per-iteration lines are held back until it is known that the pair is kept: a pair that is run again singly
                #  below would otherwise print its lines twice -- ADVICE r4)
                held = io.StringIO()
                with contextlib.redirect_stdout(held):
                    res = mur_pair(data, a[0], [dict(k=a[0], lambda_w=a[1], lambda_h=a[2]), dict(k=b[0], lambda_w=b[1], lambda_h=b[2])],
                                   engine=big, device=device, **pair_kw)
                # A pair runs on the plain device rule; single runs referee the rule in float64 once the recorded objective's jitter
                # is no longer negligible against tol2 (nmf_amd._driver.Referee).  Where that referee would have armed, the pair's
                # stop index may differ from the sequential grid's: run the two combinations again singly, from the same RNG state
                # (tight tolerances on large matrices only).
                tol2 = common.get('tol2', defaults['tol2'])
                if any(r.i < common.get('max_iter', defaults['max_iter']) - 1 and Referee.would_arm(r.obj_history, tol2) for r in res):
                    np.random.set_state(rng_state)
                    res = [run_single(i), run_single(nxt)]
                else:
                    sys.stdout.write(held.getvalue())
                out[i], out[nxt] = res
                done_now = (i, nxt)
                i += 2
            else:
                out[i] = run_single(i)
                done_now = (i,)
                i += 1
            if save_dir is not None:                           # after every run, like the sequential loop: a late failure keeps the earlier files
                for q in done_now:
                    _save(data, combos[q][0], out[q], save_dir)
    finally:
        if big is not None:
            big.close()
        for eng in singles.values():
            eng.close()
    return [(dict(features=k, lambda_w=lw, lambda_h=lh), res) for (k, lw, lh), res in zip(combos, out)]
