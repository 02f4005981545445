"""Multiplicative update rules on the MI355X engine.

Same call signature, defaults, side effects and return value as the reference's
`mur.mur` (nmf/mur.py:52-146); the loop body (mur.py:119-131) runs on the
device through libnmfx (nmfx_mur_run, include/nmfx.h)."""
import logging
from collections import namedtuple

import numpy as np

from . import _lib as L
from . import utils
from ._driver import Referee, Results, drive
from .engine import Engine

Experiment = namedtuple('Experiment', 'method components distance_type nndsvd_init max_iter tol1 tol2 lambda_w lambda_h')


def mur(x, k, *, distance_type='kl', min_iter=100, max_iter=100000, tol1=1e-5, tol2=1e-5,
        lambda_w=0.0, lambda_h=0.0, nndsvd_init=(False, 'zero'), save_dir='./results/', device=0, engine=None):
    """Lee-Seung NMF.  x: 2-D non-negative data, k: number of components.

    distance_type 'eu' | 'kl' (default 'kl' as in the reference), min_iter,
    max_iter, tol1, tol2, lambda_w, lambda_h, nndsvd_init=(bool, variant) and
    save_dir have the reference's meaning.  Returns
    Results(w, h, i, obj_history, experiment) with float64 w, h."""
    experiment = Experiment('mur', k, distance_type, nndsvd_init, max_iter, tol1, tol2,
                            lambda_w, lambda_h)
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')   # nmf/utils.py:31
    dist = L.EU if distance_type == 'eu' else L.KL

    # negative data is lifted IN PLACE on the caller's array (nmf/mur.py:99-101)
    lowest = np.min(x)
    lifted = lowest < 0
    if lifted:
        x += abs(lowest)
        logging.info('Data elevated by {}.'.format(abs(lowest)))

    init = utils.initial_factors(x, k, nndsvd_init, defer_device=True)
    with Engine.for_data(x, k, device=device, engine=engine) as eng:
        if lifted and engine is not None:
            eng.upload_v(x)             # a resident engine still holds the data as it was before the lift
        w0, h0 = utils.device_initial_factors(eng, x, k, nndsvd_init, init)
        eng.set_factors(w0, h0)
        logging.info('Entering Main Loop.')
        NEVER = 10 ** 15
        referee = None
        if distance_type == 'eu':                   # (the float64 objective kernel is the Euclidean one)
            referee = Referee(eng, lambda i: eng.mur_run(dist, lambda_w, lambda_h, NEVER, tol1, tol2, i, 1), min_iter, tol1, tol2)
        i, history = drive(
            eng,
            lambda first, count: eng.mur_run(dist, lambda_w, lambda_h, min_iter, tol1, tol2, first, count),
            lambda done: eng.mur_finish(dist, NEVER if referee is not None and referee.walked else min_iter, tol1, tol2, done),
            max_iter, tol1, tol2, referee=referee)
        mur.last_referee = referee                  # diagnostic: guard in force, iterations walked with the float64 objective
        w, h = eng.get_factors()
    return Results(w=w, h=h, i=i, obj_history=history, experiment=experiment)


def mur_pair(x, k, params, *, min_iter=100, max_iter=100000, tol1=1e-5, tol2=1e-5, nndsvd_init=(False, 'zero'),
             save_dir='./results/', engine=None, device=0):
    """TWO Euclidean MUR factorizations of the same data in one pass over it (the parameter grids of the
    reference's author, nmf/nmf_old.py:52-66): `params` = two dicts with `lambda_w`, `lambda_h` (and optionally `k`, each
    <= 64; default: the common `k`).  Equivalent to two consecutive `mur(x, k_p, distance_type='eu', ...)` calls -- same draws
    from the global numpy RNG in the same order (W, H of the first problem, then of the second), same printed lines in the same
    order, same Results -- but V is streamed once per half-iteration for both problems (nmfx_mur_pair_run: the two problems
    sit in the halves of the k = 128 layouts).  `engine`: a k = 128 Engine that already holds x (nmf_amd.grid keeps one
    resident).  Returns [Results, Results]."""
    if len(params) != 2:
        raise ValueError('mur_pair takes exactly two parameter sets')
    ks = [int(p.get('k', k)) for p in params]
    if max(ks) > 64 or min(ks) < 1:
        raise ValueError('mur_pair: each problem needs 1 <= k <= 64')
    lws = [float(p.get('lambda_w', 0.0)) for p in params]
    lhs = [float(p.get('lambda_h', 0.0)) for p in params]
    if max_iter <= 0:
        raise UnboundLocalError("local variable 'i' referenced before assignment")
    lowest = np.min(x)
    lifted = lowest < 0
    if lifted:                                                  # nmf/mur.py:99-101 (the first call lifts, the second sees >= 0)
        x += abs(lowest)
        logging.info('Data elevated by {}.'.format(abs(lowest)))
    m, n = x.shape
    inits = [utils.initial_factors(x, kk, nndsvd_init, defer_device=True) for kk in ks]     # reference RNG order: problem 0, then 1
    with Engine.for_data(x, 128, device=device, engine=engine) as eng:
        if eng.precision() != 'bf16':
            raise RuntimeError('mur_pair runs on the split-bf16 path only (NMFX_PRECISION=f32 is set, or the engine fell back)')
        if lifted and engine is not None:
            eng.upload_v(x)
        w0 = np.zeros((m, 128))
        h0 = np.zeros((128, n))
        for p, kk in enumerate(ks):
            wp, hp = utils.device_initial_factors(eng, x, kk, nndsvd_init, inits[p])
            w0[:, 64 * p:64 * p + kk] = wp
            h0[64 * p:64 * p + kk] = hp
        eng.set_factors(w0, h0)
        logging.info('Entering Main Loop.')
        done, rules = 0, [0, 0]
        from ._driver import BATCH
        while done < max_iter and not all(rules):
            count = min(BATCH, max_iter - done)
            eng.mur_pair_run(lws, lhs, min_iter, tol1, tol2, done, count)
            done += count
            if done == max_iter:
                eng.mur_pair_finish(min_iter, tol1, tol2, done)
            rules = [eng.pair_state(p)[0] for p in (0, 1)]
        out = []
        digits = utils.tol_digits(tol1, tol2)
        for p, kk in enumerate(ks):
            rule, stop_i, n_obj = eng.pair_state(p)
            history = [np.float64(v) for v in eng.pair_objectives(p, 0, n_obj)]
            w, h = eng.pair_get_factors(p, kk)
            if rule:
                i, history = stop_i, history[:stop_i + 2]
            else:
                i = max_iter - 1
            for it, val in enumerate(history[1:]):              # the lines the reference prints, problem by problem
                utils.say('[{}]: {:.{}f}'.format(it, val, digits))
            if rule:
                utils.convergence_message(rule)
                logging.warning('Converged.')
            else:
                logging.info('Max iteration reached.')
            experiment = Experiment('mur', kk, 'eu', nndsvd_init, max_iter, tol1, tol2, lws[p], lhs[p])
            out.append(Results(w=w, h=h, i=i, obj_history=history, experiment=experiment))
    return out
