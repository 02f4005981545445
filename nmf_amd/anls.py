"""ANLS on the MI355X engine.

Same call signature, defaults and return value as the reference's `anls.anls`
(nmf/anls.py:50-135).  Per outer iteration every row of W and then every column
of H is the exact solution of a regularised NNLS problem (anls.py:18-47); the
device solves them from the shared Gram matrix by block principal pivoting
(csrc/kernels_anls.hip).  `use_fcnnls` selects between two solvers of the SAME
strictly convex problems in the reference (scipy's Lawson-Hanson vs
nmf/fcnnls.py); here it is recorded in the experiment tuple (and therefore in
the save-file name) and otherwise has no effect.  `distance_type` only selects
the reported objective, exactly as in the reference."""
import logging
from collections import namedtuple

from . import _lib as L
from . import utils
from ._driver import Referee, Results, drive
from .engine import Engine

Experiment = namedtuple('Experiment', 'method components distance_type nndsvd_init max_iter tol1 tol2 lambda_w lambda_h fcnnls')


def anls(x, k, *, distance_type='eu', use_fcnnls=False, lambda_w=0, lambda_h=0, min_iter=10,
         max_iter=1000, tol1=1e-3, tol2=1e-3, nndsvd_init=(True, 'zero'), save_dir='./results/',
         device=0, engine=None):
    experiment = Experiment('anls', k, distance_type, nndsvd_init, max_iter, tol1, tol2, lambda_w,
                            lambda_h, use_fcnnls)
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')   # nmf/utils.py:31 via anls.py:108
    dist = L.EU if distance_type == 'eu' else L.KL
    init = utils.initial_factors(x, k, nndsvd_init, uniform=True, defer_device=True)
    with Engine.for_data(x, k, device=device, engine=engine) as eng:
        w0, h0 = utils.device_initial_factors(eng, x, k, nndsvd_init, init)
        eng.set_factors(w0, h0)
        eng.anls_set_distance(dist)
        NEVER = 10 ** 15
        referee = None
        if distance_type == 'eu':                   # the stop rule refereed in float64 near the stop (nmf_amd._driver.Referee)
            referee = Referee(eng, lambda i: eng.anls_run(lambda_w, lambda_h, NEVER, tol1, tol2, i, 1), min_iter, tol1, tol2)
        anls.last_referee = referee
        i, history = drive(
            eng,
            lambda first, count: eng.anls_run(lambda_w, lambda_h, min_iter, tol1, tol2, first, count),
            lambda done: eng.aoadmm_finish(NEVER if referee is not None and referee.walked else min_iter, tol1, tol2, done),
            max_iter, tol1, tol2, referee=referee)
        w, h = eng.get_factors()
        evicted, capped = eng.diagnostics()
    if capped:                    # (the reference's FCNNLS prints 'Not converged.' there, nmf/fcnnls.py:118)
        logging.warning('%d NNLS solves reached the iteration cap', capped)
    if evicted:
        logging.info('%d passive NNLS variables had a vanished pivot (dead or collinear component) and stay at zero', evicted)
    anls.last_diagnostics = {'nnls_evicted': evicted, 'nnls_capped': capped}
    return Results(w=w, h=h, i=i, obj_history=history, experiment=experiment)
