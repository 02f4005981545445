"""Multiplicative update rules on the MI355X engine.

Same call signature, defaults, side effects and return value as the reference's
`mur.mur` (nmf/mur.py:52-146); the loop body (mur.py:119-131) runs on the
device through libnmfx (nmfx_mur_run, include/nmfx.h)."""
import logging
from collections import namedtuple

import numpy as np

from . import _lib as L
from . import utils
from ._driver import Results, drive
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
        i, history = drive(
            eng,
            lambda first, count: eng.mur_run(dist, lambda_w, lambda_h, min_iter, tol1, tol2, first, count),
            lambda done: eng.mur_finish(dist, min_iter, tol1, tol2, done),
            max_iter, tol1, tol2)
        w, h = eng.get_factors()
    return Results(w=w, h=h, i=i, obj_history=history, experiment=experiment)
