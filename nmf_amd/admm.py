"""ADMM on the MI355X engine.

Same call signature, defaults and return value as the reference's `admm.admm`
(nmf/admm.py:233-345); the loop body (admm.py:292-334: two shifted-Gram solves,
two prox operators, dual updates, objective) runs on the device through libnmfx
(nmfx_admm_run).  Regularisers 'nn', 'l1n', 'l2n' ('l2n' is the reference's
default reg_h) and 'l1inf' / 'l1inf_transpose' (nmf/admm.py:158-210, as written
there; the iteration they define diverges in the reference too)."""
from collections import namedtuple

import numpy as np

from . import _lib as L
from . import utils
from ._driver import Referee, Results, drive
from .engine import Engine

Experiment = namedtuple('Experiment', 'method components rho distance_type nndsvd_init min_iter max_iter tol1 tol2 lambda_w prox_w lambda_h prox_h')


def _prox_code(kind):
    if kind in L.PROX:
        return L.PROX[kind]
    raise TypeError('Unknown prox_type.')                       # nmf/admm.py:213


def l2n_operator(k, rho, lam):
    """Inverse of the matrix the reference hands to spsolve in prox 'l2n'
    (nmf/admm.py:142-152): a = (lam * T^T T + rho I) / rho with
    T = tridiag(-1, 2, -1) of order k.  k x k, formed once per run on the host."""
    t = 2.0 * np.eye(k) - np.eye(k, k=1) - np.eye(k, k=-1)
    return np.linalg.inv((lam * t.T @ t + rho * np.eye(k)) / rho)


def admm(v, k, *, rho=1, distance_type='eu', reg_w=(0, 'nn'), reg_h=(0, 'l2n'), min_iter=10,
         max_iter=100000, tol1=1e-3, tol2=1e-3, nndsvd_init=(True, 'zero'), save_dir='./results/',
         device=0, engine=None):
    """ADMM NMF.  rho: fixed penalty; reg_w / reg_h = (lambda, 'nn' | 'l1n' | 'l2n');
    other arguments as in the reference.  Returns Results(w, h, i, obj_history, experiment)."""
    experiment = Experiment('admm', k, rho, distance_type, nndsvd_init, min_iter, max_iter, tol1, tol2,
                            reg_w[0], reg_w[1], reg_h[0], reg_h[1])
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')   # nmf/utils.py:31 via admm.py:289
    dist = L.EU if distance_type == 'eu' else L.KL
    init = utils.initial_factors(v, k, nndsvd_init, defer_device=True)
    prox_h = _prox_code(reg_h[1])
    prox_w = _prox_code(reg_w[1])
    if rho == 0 and (reg_h[1] != 'nn' or reg_w[1] != 'nn'):
        # rho = 0 is a plain Gram solve in the reference (admm.py:230); every prox but 'nn' divides by rho (admm.py:135,
        # 150, 161) in the first iteration
        raise ZeroDivisionError('division by zero')
    with Engine.for_data(v, k, device=device, engine=engine) as eng:
        w0, h0 = utils.device_initial_factors(eng, v, k, nndsvd_init, init)
        eng.set_factors(w0, h0)
        if prox_w == L.PROX['l2n']:
            eng.set_l2n_operator(0, l2n_operator(k, rho, reg_w[0]))
        if prox_h == L.PROX['l2n']:
            eng.set_l2n_operator(1, l2n_operator(k, rho, reg_h[0]))
        def will_go(i):
            # stdout of the reference's 'l1inf_transpose' branch (nmf/admm.py:190), H first (admm.py:319-320)
            if reg_h[1] == 'l1inf_transpose':
                utils.say('will go {}'.format(v.shape[1]))
            if reg_w[1] == 'l1inf_transpose':
                utils.say('will go {}'.format(v.shape[0]))

        NEVER = 10 ** 15
        referee = None
        if distance_type == 'eu':                   # the stop rule refereed in float64 near the stop (nmf_amd._driver.Referee)
            referee = Referee(eng, lambda i: eng.admm_run(dist, rho, prox_w, reg_w[0], prox_h, reg_h[0], NEVER, tol1, tol2, i, 1),
                              min_iter, tol1, tol2)
        admm.last_referee = referee
        i, history = drive(
            eng,
            lambda first, count: eng.admm_run(dist, rho, prox_w, reg_w[0], prox_h, reg_h[0], min_iter,
                                              tol1, tol2, first, count),
            lambda done: eng.aoadmm_finish(NEVER if referee is not None and referee.walked else min_iter, tol1, tol2, done),
            max_iter, tol1, tol2, before_line=will_go, referee=referee)
        w, h = eng.get_factors()
    return Results(w=w, h=h, i=i, obj_history=history, experiment=experiment)
