"""Host logic of the float64 referee of the stop rule (nmf_amd._driver.Referee), on a fake engine (no GPU)."""
import numpy as np

from nmf_amd._driver import Referee


class FakeEngine:
    def __init__(self):
        self.guards = []

    def set_stop_guard(self, g):
        self.guards.append(g)


def _history(n, jitter, seed=0):
    """A smoothly decreasing objective with independent noise of spread `jitter` per value."""
    t = np.arange(n, dtype=np.float64)
    return list(1000.0 + 50.0 * np.exp(-t / 200.0) + jitter * np.random.RandomState(seed).randn(n))


def test_guard_arms_only_when_the_jitter_matters_and_records_since_when():
    eng = FakeEngine()
    ref = Referee(eng, lambda i: None, min_iter=10, tol1=1e-5, tol2=1e-3)
    ref.mode = "auto"
    ref.update_guard(list(1000.0 - 0.01 * np.arange(200.0)))
    assert ref.guard == 0.0 and eng.guards == []                 # a smooth history: plain rule
    hist = _history(200, 1e-4)
    ref.update_guard(hist)
    assert ref.guard > 0 and eng.guards == [ref.guard]
    assert ref.guard_since == len(hist) - 1                      # the next iteration queued is index len(hist) - 1


def test_candidate_in_front_of_the_guard_is_left_to_the_walk():
    """the guard is armed between batches; a candidate at the FIRST index tested with it says nothing about the index
    before it (tested with the plain rule), so the recorded pair must not confirm it -- the float64 walk decides."""
    eng = FakeEngine()
    tol2 = 1e-3
    ref = Referee(eng, lambda i: None, min_iter=10, tol1=1e-5, tol2=tol2)
    ref.mode = "auto"
    hist = _history(129, 1e-4)                                   # after two batches of 64: obj[0 .. 128]
    ref.update_guard(hist)
    assert ref.guard > 0 and ref.guard_since == 128
    flat = hist[-1]
    first = hist + [flat + 0.5 * tol2, flat + 0.5 * tol2]        # a pair that satisfies the plain rule with room to spare
    assert not ref.confirms(first, 128)                          # index 127 was tested without the guard: walk
    assert ref.confirmed == 0
    later = hist + [flat - 10 * tol2, flat - 10 * tol2 + 0.5 * tol2, 0.0]
    assert ref.confirms(later, 129)                              # index 128 WAS tested with the guard and did not fire
    assert ref.confirmed == 1


def test_first_tested_index_confirms_whatever_the_guard_history():
    """candidate = min_iter + 1: the rule is never evaluated before it (nmf/mur.py:131 `i > min_iter`), nothing can have been missed."""
    eng = FakeEngine()
    ref = Referee(eng, lambda i: None, min_iter=128, tol1=1e-5, tol2=1e-3)
    ref.mode = "auto"
    hist = _history(130, 1e-4)
    ref.update_guard(hist)
    assert ref.guard_since == 129
    flat = hist[-1]
    assert ref.confirms(hist + [flat + 1e-3], 129)


def test_exchange_mode_parsing(monkeypatch):
    """NMFX_DIST_EXCHANGE (nmf_amd.dist.exchange_mode): the default, the two spellings of each form, and a loud failure otherwise."""
    import pytest
    from nmf_amd import dist as nd
    monkeypatch.delenv("NMFX_DIST_EXCHANGE", raising=False)
    assert nd.exchange_mode() == "allreduce"
    for val, want in (("allreduce", "allreduce"), ("0", "allreduce"), ("rsag", "rsag"), ("RS+AG", "rsag"), ("1", "rsag")):
        monkeypatch.setenv("NMFX_DIST_EXCHANGE", val)
        assert nd.exchange_mode() == want
    monkeypatch.setenv("NMFX_DIST_EXCHANGE", "ring")
    with pytest.raises(ValueError):
        nd.exchange_mode()
