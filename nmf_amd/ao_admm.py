"""AO-ADMM on the MI355X engine.

Same call signature, defaults and return value as the reference's
`ao_admm.ao_admm` (nmf/ao_admm.py:201-311).  Each outer iteration -- the H
sub-problem, then the W sub-problem on the transposed data, each with its Gram
system and up to `admm_iter` inner rounds (ao_admm.py:46-68) -- runs on the
device through libnmfx (nmfx_aoadmm_run); the inner stop test (ao_admm.py:33-43)
is evaluated on the device too.

Regularisers: 'nn' and 'l1n' are built (any number of components, both losses).  'l2n' raises
ValueError exactly like the reference does on numpy >= 1.24 (ao_admm.py:128 builds a ragged array; it is also the reference's
DEFAULT reg_h, so callers must pass reg_h explicitly).  'l1inf' / 'l1inf_transpose' (nmf/ao_admm.py:143-195, word for word the
operator of nmf/admm.py:158-210) run on the device with either loss and at most 128 components (r4: least squares; r5: KL): in the
reference they wipe a factor out within a few outer iterations -- least squares: W = 0 after the first one as `reg_w`, H = 0 inside
the first one as `reg_h`; KL: after one to four -- and the next Cholesky factorisation raises numpy.linalg.LinAlgError; here the
same pivot test (NMFX_E_NOTPD) raises the same exception in the same outer iteration, and a run whose `max_iter` ends before it
returns the reference's Results (tests/golden/aoadmm_{eu,kl}_*_l1inf*.npz).  Beyond 128 components they still raise that exception
up front."""
from collections import namedtuple

import numpy as np

from . import _lib as L
from . import utils
from ._driver import Referee, Results, drive
from .engine import Engine

Experiment = namedtuple('Experiment', 'method components distance_type nndsvd_init min_iter max_iter admm_iter tol1 tol2 lambda_w prox_w lambda_h prox_h')


def _prox_code(kind, on_device=False):
    """on_device: 'l1inf*' is available (single GPU, least-squares loss, k <= 128); otherwise it raises what the reference ends in."""
    if kind in ('nn', 'l1n'):
        return L.PROX[kind]
    if kind in ('l1inf', 'l1inf_transpose') and on_device:
        return L.PROX[kind]
    if kind == 'l2n':
        raise ValueError('setting an array element with a sequence. The requested array has an '
                         'inhomogeneous shape (reference nmf/ao_admm.py:128 on numpy >= 1.24)')
    if kind in ('l1inf', 'l1inf_transpose'):
        raise np.linalg.LinAlgError('1-th leading minor of the array is not positive definite')   # scipy cholesky, ao_admm.py:55
    raise TypeError('Unknown prox_type.')                       # nmf/ao_admm.py:198


def ao_admm(v, k, *, distance_type='eu', reg_w=(0, 'nn'), reg_h=(0, 'l2n'), min_iter=10,
            max_iter=100000, admm_iter=10, tol1=1e-3, tol2=1e-3, nndsvd_init=(True, 'zero'),
            save_dir='./results/', device=0, engine=None):
    """AO-ADMM NMF.  reg_w / reg_h = (lambda, 'nn' | 'l1n'); other arguments as
    in the reference.  Returns Results(w, h, i, obj_history, experiment)."""
    experiment = Experiment('ao_admm', k, distance_type, nndsvd_init, min_iter, max_iter, admm_iter,
                            tol1, tol2, reg_w[0], reg_w[1], reg_h[0], reg_h[1])
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')   # nmf/utils.py:31 via ao_admm.py:256
    dist = L.EU if distance_type == 'eu' else L.KL
    init = utils.initial_factors(v, k, nndsvd_init, defer_device=True)
    # the reference meets the H regulariser first (ao_admm.py:261), then W's
    on_device = k <= 128
    prox_h = _prox_code(reg_h[1], on_device)
    prox_w = _prox_code(reg_w[1], on_device)

    with Engine.for_data(v, k, device=device, engine=engine) as eng:
        w0, h0 = utils.device_initial_factors(eng, v, k, nndsvd_init, init)
        eng.set_factors(w0, h0)
        seen = {}

        def breaks(i):
            if i not in seen:
                lo = i
                for row, pair in enumerate(eng.inner_counts(lo, 1)):
                    seen[lo + row] = pair
            for word in seen.pop(i):
                rounds, fired = int(word) & 0xFFFF, (int(word) >> 16) & 1
                if fired:
                    utils.say('ADMM break after {} iterations.'.format(rounds - 1))

        NEVER = 10 ** 15
        referee = None
        if distance_type == 'eu':                   # the stop rule refereed in float64 near the stop (nmf_amd._driver.Referee)
            referee = Referee(eng, lambda i: eng.aoadmm_run(dist, prox_w, reg_w[0], prox_h, reg_h[0], admm_iter, NEVER, tol1, tol2, i, 1),
                              min_iter, tol1, tol2)
        ao_admm.last_referee = referee
        try:
            i, history = drive(
                eng,
                lambda first, count: eng.aoadmm_run(dist, prox_w, reg_w[0], prox_h, reg_h[0], admm_iter,
                                                    min_iter, tol1, tol2, first, count),
                lambda done: eng.aoadmm_finish(NEVER if referee is not None and referee.walked else min_iter, tol1, tol2, done),
                max_iter, tol1, tol2, before_line=breaks, referee=referee)
        except L.NmfxError as e:
            if e.code == L.NMFX_E_NOTPD:                        # scipy cholesky, ao_admm.py:55
                raise np.linalg.LinAlgError('matrix is not positive definite (Gram + rho I)') from e
            raise
        w, h = eng.get_factors()
        inner = eng.inner_counts(0, i + 1) & 0xFFFF
        paths = eng.inner_paths()
    res = Results(w=w, h=h, i=i, obj_history=history, experiment=experiment)
    ao_admm.last_inner_counts = inner       # diagnostic: inner rounds per outer iteration (h, w)
    ao_admm.last_inner_paths = paths        # diagnostic: how the speculative inner rounds went (Engine.inner_paths)
    return res
