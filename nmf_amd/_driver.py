"""Shared outer-loop driver: queues iterations on the device in batches and
replays the reference's observable behaviour (per-iteration print, obj_history,
stop index) from the device-side state."""
import logging
from collections import namedtuple

import numpy as np

from . import utils

Results = namedtuple('Results', 'w h i obj_history experiment')

BATCH = 64     # outer iterations queued between two host syncs


class Referee:
    """The stop rule refereed in float64 (MUR, Euclidean loss, one GPU).

    The objective the device records every iteration is evaluated with float32 products: fine as a value (1e-6 relative), but its
    iteration-to-iteration jitter (~3e-9 of the objective at 16384 x 8192) is what `new >= old - tol2` (nmf/utils.py:10) sees once
    tol2 is below ~1e-6 of the objective -- the stop came 0.4 % early at tol2 = 1e-3 on the config-2 matrix.  The iterates
    themselves are fine: evaluated in float64 (nmfx_objective_f64) their objective decreases smoothly (jitter 100 x smaller) and
    the reference's rule fires on it at exactly the iteration at which the float64 oracle, continued from the device's iterate,
    stops.  So: the jitter sigma of the recorded decreases is estimated from the history after every batch; when 6 sigma is no
    longer negligible against tol2 the device's rule gets that much head start (nmfx_set_stop_guard: it fires EARLY, as a
    candidate), and from the candidate on the loop walks one iteration at a time with the float64 objective deciding.
    NMFX_VERIFY_STOP=0 turns it off, =1 forces the guard to at least 10 % of tol2."""

    def __init__(self, engine, run_one, min_iter, tol1, tol2):
        import os
        self.eng, self.run_one, self.min_iter, self.tol1, self.tol2 = engine, run_one, min_iter, tol1, tol2
        self.mode = os.environ.get("NMFX_VERIFY_STOP", "auto")
        self.guard = 0.0
        self.jitter6 = 0.0                           # the 6 sigma part of the guard (without the head start for the curvature)
        self.guard_since = 0                         # loop index from which the guard now in force has been tested with (see confirms)
        self.walked = 0
        self.confirmed = 0                           # candidates accepted from the recorded history alone (see confirms)
        self.final_rule = 0

    @staticmethod
    def would_arm(history, tol2):
        """True when update_guard, shown this history, puts a guard in force (mode 'auto').  nmf_amd.grid uses it on the histories
        of a PAIR run (nmf_amd.mur.mur_pair keeps the plain device rule): such a combination is run again singly, refereed."""
        import os
        if os.environ.get("NMFX_VERIFY_STOP", "auto") == "0" or len(history) < 24:
            return False
        tail = np.asarray(history[-68:], dtype=np.float64)
        d4 = np.diff(tail, n=4)
        g = 6.0 * 1.4826 * np.median(np.abs(d4 - np.median(d4))) * np.sqrt(2.0 / 70.0)
        return bool(np.isfinite(g) and g >= 1e-6 * tol2) or os.environ.get("NMFX_VERIFY_STOP", "auto") == "1"

    def update_guard(self, history):
        if self.mode == "0" or len(history) < 24:
            return
        # FOURTH differences of the recorded objectives: what a smooth history leaves of itself there is far below its curvature
        # (second differences mistook the curvature of a young run for jitter and started the walk hundreds of iterations early),
        # while independent noise of spread s per value arrives with spread sqrt(70) s; the jitter of a decrease is sqrt(2) s
        tail = np.asarray(history[-68:], dtype=np.float64)
        d4 = np.diff(tail, n=4)
        sigma_d = 1.4826 * np.median(np.abs(d4 - np.median(d4))) * np.sqrt(2.0 / 70.0)
        g = 6.0 * sigma_d
        self.jitter6 = g if np.isfinite(g) else 0.0
        if g >= 1e-6 * self.tol2:
            # The walk can only test loop indices BEHIND the candidate (the pair in front of it is gone), so the candidate has to
            # come at least one iteration early even without any jitter: twice the change of the decrease per iteration on top.
            g += 2.0 * abs(np.median(np.diff(tail, n=2)))
        g = min(g, self.tol2)                        # (a jitter beyond tol2 itself: the walk starts at twice tol2, not earlier)
        if not np.isfinite(g):                       # (a history with inf / nan in it: nothing to estimate from)
            return
        if self.mode == "1":
            g = max(g, 0.1 * self.tol2)
        # (How far the jitter moves the stop depends on how fast the decrease itself changes, not on its size against tol2 -- at
        # tol2 = 1e-3 on the config-2 matrix a jitter of 0.1 % of tol2 was worth 64 iterations -- and the head start costs about
        # guard / |change of the decrease per iteration| refereed iterations: one or two where the decrease still moves fast.)
        if g < 1e-6 * self.tol2:                     # nothing the rule could see: plain rule, no referee
            g = 0.0
        if g != self.guard:
            self.guard = g
            self.guard_since = len(history) - 1      # (history holds obj[0 .. done]: the next iteration queued is index `done`)
            self.eng.set_stop_guard(g)

    def confirms(self, history, candidate_i):
        """The walk below can only test loop indices BEHIND the candidate (the pair that entered iteration `candidate_i` is gone),
        so a candidate at which the reference's rule already holds -- the first tested index min_iter + 1 of a run that converged
        before min_iter, or a decrease that falls faster than the head start assumed -- would come back one iteration late
        (ADVICE r3).  If the RECORDED pair of objectives satisfies the plain rule with the whole jitter estimate to spare,
        `new >= old - tol2 + 6 sigma`, the float64 values satisfy it too: the candidate is the stop.  (The index before it did
        not fire with the guard, i.e. its recorded decrease exceeded tol2 by more than the jitter, so the rule did not hold there --
        which is only known if that index WAS tested with the guard now in force: the guard is armed or raised between batches, so a
        candidate at the first index of the batch behind that change is left to the walk)"""
        if candidate_i + 1 >= len(history) or candidate_i <= self.min_iter:
            return False
        if candidate_i - 1 > self.min_iter and candidate_i - 1 < self.guard_since:
            return False
        new, old = history[candidate_i + 1], history[candidate_i]
        if new >= old - self.tol2 + self.jitter6:
            self.confirmed += 1
            return True
        return False

    def walk(self, candidate_i, max_iter, pull):
        """From the candidate stop (device rule 2 with the guard at loop index `candidate_i`) on: the current pair is the one that
        entered iteration j = candidate_i + 1.  Returns (rule, stop_i, iterations done)."""
        eng = self.eng
        i = candidate_i + 1
        eng.resume()
        old = eng.objective_f64()
        while i < max_iter:
            self.run_one(i)                          # iteration i in full (stop rule off): the pair i + 1
            pull()                                   # the device's recorded obj[i] -> history, printed like any other line
            new = eng.objective_f64()
            self.walked += 1
            if i > self.min_iter:                    # nmf/mur.py:131 with nmf/utils.py:4-15 on the float64 values
                rule = 1 if new < self.tol1 else 2 if new >= old - self.tol2 else 0
                if rule:
                    return rule, i, i + 1
            old = new
            i += 1
        return 0, -1, max_iter


def drive(engine, run_batch, finish, max_iter, tol1, tol2, before_line=None, referee=None):
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
    def pull():
        _, _, n_obj = engine.state()
        for val in engine.objectives(len(history), n_obj - len(history)):
            history.append(np.float64(val))
            if len(history) >= 2:
                if before_line is not None:
                    before_line(len(history) - 2)
                utils.say('[{}]: {:.{}f}'.format(len(history) - 2, val, digits))

    while done < max_iter and not rule:
        count = min(BATCH, max_iter - done)
        run_batch(done, count)
        done += count
        if done == max_iter:
            finish(done)
        rule, stop_i, n_obj = engine.state()
        pull()
        if referee is not None:
            if rule == 2 and referee.guard > 0 and referee.confirms(history, stop_i):
                pass                                 # the plain rule holds at the candidate beyond the jitter: it IS the stop
            elif rule == 2 and referee.guard > 0:    # a candidate: the float64 objective decides from here on
                rule, stop_i, done = referee.walk(stop_i, max_iter, pull)
                finish(done)                         # the objective of the last pair, evaluated as every other history entry
                pull()
                if not rule:
                    break
            elif not rule:
                referee.update_guard(history)
    if referee is not None:
        referee.final_rule = rule                    # (callers that need it: the device's flag was cleared by a walk)
    if rule:
        utils.convergence_message(rule)
        logging.warning('Converged.')
        return stop_i, history[:stop_i + 2]
    logging.info('Max iteration reached.')
    return max_iter - 1, history
