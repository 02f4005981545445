"""Shared outer-loop driver: queues iterations on the device in batches and
replays the reference's observable behaviour (per-iteration print, obj_history,
stop index) from the device-side state."""
import logging
from collections import namedtuple

import numpy as np

from . import utils

Results = namedtuple('Results', 'w h i obj_history experiment')

BATCH = 64     # outer iterations queued between two host syncs


def drive(engine, run_batch, finish, max_iter, tol1, tol2, before_line=None):
    """run_batch(first, count) queues iterations; finish(done) completes the
    bookkeeping of the last one; before_line(i) may print what the reference
    prints inside iteration i before its objective line.  Returns (i, obj_history) like the reference
    loops (nmf/mur.py:119-145): obj_history has i + 2 entries."""
    if max_iter <= 0:
        # reference: `for i in range(0)` never binds i -> UnboundLocalError at mur.py:145
        raise UnboundLocalError("local variable 'i' referenced before assignment")
    digits = utils.tol_digits(tol1, tol2)
    history = []
    done = 0
    rule, stop_i = 0, -1
    while done < max_iter and not rule:
        count = min(BATCH, max_iter - done)
        run_batch(done, count)
        done += count
        if done == max_iter:
            finish(done)
        rule, stop_i, n_obj = engine.state()
        fresh = engine.objectives(len(history), n_obj - len(history))
        for val in fresh:
            history.append(np.float64(val))
            if len(history) >= 2:
                if before_line is not None:
                    before_line(len(history) - 2)
                utils.say('[{}]: {:.{}f}'.format(len(history) - 2, val, digits))
    if rule:
        utils.convergence_message(rule)
        logging.warning('Converged.')
        return stop_i, history[:stop_i + 2]
    logging.info('Max iteration reached.')
    return max_iter - 1, history
