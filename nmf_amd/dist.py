"""Row-sharded MUR over several GPUs of one node (one process per GPU).

Rank p holds the rows [r0, r1) of V and of W; H (k x n) and H H^T are
replicated.  The W update is row-local (nmf/mur.py:29: row i of W depends only
on row i of V).  The H update (nmf/mur.py:45) needs W^T V = sum_p W_p^T V_p and
W^T W = sum_p W_p^T W_p: ONE sum-all-reduce per outer iteration of the packed
f32 buffer [W^T V | W^T W] (k*n + k*k elements) plus the f64 objective partial.
Every rank then applies the identical H update, so no broadcast is needed and
the device-side stop flag agrees on all ranks.

The collective is `torch.distributed.all_reduce` (backend "nccl" = RCCL over
xGMI on the GPU box; "gloo" in the CPU tests, where a numpy stand-in engine
from the test-suite replaces the HIP engine).  The loop below only talks to an
object with phase_a / phase_b / finish_a / finish_b / state / objectives, so the
sharding logic is the same code in both cases.
"""
import logging

import numpy as np

from . import utils
from ._driver import Results


def row_range(m, rank, world):
    """Contiguous, balanced row block of `rank` (any m, any world)."""
    return (m * rank) // world, (m * (rank + 1)) // world


class TorchComm:
    """Sum-all-reduce of the engine's exchange buffers through torch.distributed.

    `stage_through_host=True` copies device tensors to the host, reduces them
    with a CPU backend (gloo) and copies back: slow, but lets several ranks
    share ONE GPU in tests where RCCL refuses duplicate devices."""

    def __init__(self, group=None, stage_through_host=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stage = stage_through_host
        self._coalesce = hasattr(dist, "_coalescing_manager")

    def all_reduce(self, *tensors):
        # RCCL: put the f32 and the f64 reduction into ONE group launch (ncclGroupStart/End)
        # -- one host call and one kernel instead of two per outer iteration.
        if len(tensors) > 1 and not self.stage and all(t.is_cuda for t in tensors) and self._coalesce:
            try:
                with self.dist._coalescing_manager(group=self.group, device=tensors[0].device, async_ops=False):
                    for t in tensors:
                        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
                return
            except Exception:      # noqa: BLE001  (private torch API: fall back for good)
                self._coalesce = False
        for t in tensors:
            if self.stage and t.is_cuda:
                import torch
                torch.cuda.current_stream().synchronize()
                host = t.cpu()
                self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(host)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def barrier(self):
        self.dist.barrier(group=self.group)


class DeviceShard:
    """HIP engine for one row shard, with exchange buffers allocated by torch so
    that RCCL can reduce them in place, and running on torch's current stream so
    the collective is ordered after phase A and before phase B."""

    def __init__(self, v_local, k, w0_local, h0, device):
        import torch
        from .engine import Engine
        self.torch = torch
        torch.cuda.set_device(device)
        self.eng = Engine(v_local.shape[0], v_local.shape[1], k, device=device)
        n32, n64 = self.eng.exchange_sizes()
        self.xf32 = torch.zeros(n32, dtype=torch.float32, device=f"cuda:{device}")
        self.xf64 = torch.zeros(n64, dtype=torch.float64, device=f"cuda:{device}")
        torch.cuda.synchronize()
        self.eng.set_exchange_buffers(self.xf32.data_ptr(), self.xf64.data_ptr())
        self.eng.set_stream(torch.cuda.current_stream().cuda_stream)
        self.eng.upload_v(v_local)
        self.eng.set_factors(w0_local, h0)

    def buffers(self):
        return self.xf32, self.xf64

    def phase_a(self, dist_code, lambda_w, j):
        self.eng.mur_phase_a(dist_code, lambda_w, j)

    def phase_b(self, dist_code, lambda_h, min_iter, tol1, tol2, j):
        self.eng.mur_phase_b(dist_code, lambda_h, min_iter, tol1, tol2, j)

    def finish_a(self, dist_code, j):
        self.eng.mur_finish_a(dist_code, j)

    def finish_b(self, min_iter, tol1, tol2, j):
        self.eng.mur_finish_b(min_iter, tol1, tol2, j)

    # AO-ADMM / ANLS phases: straight delegation to the engine
    def ao_h_products(self, j):
        self.eng.aoadmm_phase_h_products(j)

    def ao_h_solve(self, prox_h, lam_h, admm_iter, min_iter, tol1, tol2, j):
        self.eng.aoadmm_phase_h_solve(prox_h, lam_h, admm_iter, min_iter, tol1, tol2, j)

    def ao_w_products(self, min_iter, tol1, tol2, j):
        self.eng.aoadmm_phase_w_products(min_iter, tol1, tol2, j)

    def ao_w_round(self, prox_w, lam_w, rnd):
        self.eng.aoadmm_phase_w_round(prox_w, lam_w, rnd)

    def ao_w_close(self, admm_iter, j):
        self.eng.aoadmm_phase_w_close(admm_iter, j)

    def objective_partial(self):
        self.eng.objective_partial()

    def anls_objective(self, j):
        self.eng.anls_phase_objective(j)

    def anls_w(self, lam_w, min_iter, tol1, tol2, j):
        self.eng.anls_phase_w(lam_w, min_iter, tol1, tol2, j)

    def anls_h(self, lam_h, j):
        self.eng.anls_phase_h(lam_h, j)

    def state(self):
        return self.eng.state()

    def objectives(self, first, count):
        return self.eng.objectives(first, count)

    def get_factors(self):
        return self.eng.get_factors()

    def synchronize(self):
        self.eng.synchronize()

    def close(self):
        self.eng.close()


def run_iterations(shard, comm, dist_code, lambda_w, lambda_h, min_iter, tol1, tol2, first, count):
    """Queue `count` sharded outer iterations (no host sync)."""
    bufs = shard.buffers()
    for j in range(first, first + count):
        shard.phase_a(dist_code, lambda_w, j)
        comm.all_reduce(*bufs)
        shard.phase_b(dist_code, lambda_h, min_iter, tol1, tol2, j)


class GraphedIterations:
    """Two sharded outer iterations (phase A -> RCCL all-reduce -> phase B, twice: one W ping-pong
    period) captured into ONE hipGraph and replayed, so that the host issues one graph launch per
    two iterations instead of ~10 kernel launches + 2 collectives.  At 8 ranks an iteration is
    ~0.1 ms of device work and the Python-driven loop is host-bound; the replay is not.

    The iteration index each launch needs (objective slot, `i > min_iter` test, nmf/mur.py:131)
    comes from a device-side base that the graph's last node advances by 2
    (nmfx_shift_iteration_base), so one capture serves the whole run.  The stop rule is evaluated
    on the device by every launch exactly as in the eager loop; after it fires the remaining
    launches of a replay are no-ops.  Requires an engine on torch's stream (DeviceShard) and a
    CUDA/HIP-capturable collective ("nccl" = RCCL)."""

    def __init__(self, shard, comm, dist_code, lambda_w, lambda_h, min_iter, tol1, tol2):
        torch = shard.torch
        self.shard, self.torch = shard, torch
        self.args = (dist_code, lambda_w, lambda_h, min_iter, tol1, tol2)
        self.graph = torch.cuda.CUDAGraph()
        bufs = shard.buffers()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            # thread_local: the process group's watchdog thread may touch the runtime meanwhile
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                shard.eng.set_stream(torch.cuda.current_stream().cuda_stream)
                for j in (0, 1):
                    shard.phase_a(dist_code, lambda_w, j)
                    comm.all_reduce(*bufs)
                    shard.phase_b(dist_code, lambda_h, min_iter, tol1, tol2, j)
                shard.eng.shift_iteration_base(2)
        finally:
            shard.eng.set_stream(torch.cuda.current_stream().cuda_stream)

    def run(self, first, count):
        """Queue `count` (even) iterations starting at the (even) index `first`."""
        if first % 2 or count % 2:
            raise ValueError("graphed iterations come in pairs starting at an even index")
        eng = self.shard.eng
        eng.shift_iteration_base(first)
        for _ in range(count // 2):
            self.graph.replay()
        eng.shift_iteration_base(-(first + count))


class Runner:
    """run(first, count) for the sharded loop: eager for the first pair of iterations (lazy
    allocations, RCCL communicator and kernel attributes come into being there), hipGraph replays
    afterwards.  A capture that fails for any reason leaves the eager loop in charge.
    `graph=None` reads NMFX_DIST_GRAPH (default off, "1" turns it on; only device shards with a device
    collective can be captured)."""

    def __init__(self, shard, comm, dist_code, lambda_w, lambda_h, min_iter, tol1, tol2, max_iter, graph=None):
        import os
        self.shard, self.comm = shard, comm
        self.args = (dist_code, lambda_w, lambda_h, min_iter, tol1, tol2)
        if graph is None:
            # opt-in: capturing RCCL collectives of a multi-rank world cannot be rehearsed on a one-GPU
            # box, and the replay only removes launch gaps (5 % at 2048 rows per rank) -- the eager loop
            # is GPU-bound, not host-bound
            graph = os.environ.get("NMFX_DIST_GRAPH", "0") == "1"
        capturable = isinstance(shard, DeviceShard) and not getattr(comm, "stage", True)
        self.want_graph = bool(graph and capturable and max_iter >= 4)
        self.graph = None
        self.mode = "eager"
        if self.want_graph:
            shard.eng.reserve_objectives(max_iter + 2)     # the history must not move under the graph

    def eager(self, first, count):
        d, lw, lh, mi, t1, t2 = self.args
        run_iterations(self.shard, self.comm, d, lw, lh, mi, t1, t2, first, count)

    def _capture(self):
        self.want_graph = False
        try:
            self.shard.synchronize()
            self.graph = GraphedIterations(self.shard, self.comm, *self.args)
            self.mode = "hipgraph"
        except Exception as exc:      # noqa: BLE001
            logging.warning("hipGraph capture of the sharded iteration failed (%s); running eagerly", exc)
            self.graph = None

    def ensure_graph(self, done):
        """Capture now (outside any timed region) if a capture is still pending and at least one
        eager iteration has created the lazy state (allocations, RCCL communicator)."""
        if self.want_graph and done >= 1:
            self._capture()

    def __call__(self, first, count):
        if self.want_graph and first < 2:
            head = min(count, 2 - first)
            self.eager(first, head)
            first, count = first + head, count - head
        if self.want_graph and count >= 2:
            self._capture()
        if self.graph is None or count <= 0:
            if count > 0:
                self.eager(first, count)
            return
        if first % 2:                 # realign to the graph's even/odd pair
            self.eager(first, 1)
            first, count = first + 1, count - 1
        pairs = count // 2
        if pairs:
            self.graph.run(first, 2 * pairs)
        if count - 2 * pairs:
            self.eager(first + 2 * pairs, 1)


def finish(shard, comm, dist_code, min_iter, tol1, tol2, done):
    shard.finish_a(dist_code, done)
    comm.all_reduce(shard.buffers()[1])
    shard.finish_b(min_iter, tol1, tol2, done)


def mur_sharded(shard, comm, *, distance_type='eu', min_iter=100, max_iter=100000, tol1=1e-5,
                tol2=1e-5, lambda_w=0.0, lambda_h=0.0, batch=32, experiment=None, graph=None):
    """The reference's MUR loop (nmf/mur.py:119-145) over a row-sharded V.
    Returns Results whose `w` is THIS rank's row block; h, i and obj_history
    are identical on every rank."""
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')
    if max_iter <= 0:
        raise UnboundLocalError("local variable 'i' referenced before assignment")
    code = 0 if distance_type == 'eu' else 1
    digits = utils.tol_digits(tol1, tol2)
    history, done, rule, stop_i = [], 0, 0, -1
    runner = Runner(shard, comm, code, lambda_w, lambda_h, min_iter, tol1, tol2, max_iter, graph=graph)
    while done < max_iter and not rule:
        count = min(batch, max_iter - done)
        runner(done, count)
        done += count
        if done == max_iter:
            finish(shard, comm, code, min_iter, tol1, tol2, done)
        rule, stop_i, n_obj = shard.state()
        for val in shard.objectives(len(history), n_obj - len(history)):
            history.append(np.float64(val))
            if len(history) >= 2 and comm.rank == 0:
                utils.say('[{}]: {:.{}f}'.format(len(history) - 2, val, digits))
    w, h = shard.get_factors()
    if rule:
        if comm.rank == 0:
            utils.convergence_message(rule)
        logging.warning('Converged.')
        return Results(w, h, stop_i, history[:stop_i + 2], experiment)
    return Results(w, h, max_iter - 1, history, experiment)


def _sharded_loop(shard, comm, queue, finish_run, *, max_iter, tol1, tol2, batch, experiment):
    """The reference's outer loop shape (obj_history, `[i]: objective` lines, stop bookkeeping)
    around `queue(first, count)`, which only queues device work and collectives."""
    if max_iter <= 0:
        raise UnboundLocalError("local variable 'i' referenced before assignment")
    digits = utils.tol_digits(tol1, tol2)
    history, done, rule, stop_i = [], 0, 0, -1
    while done < max_iter and not rule:
        count = min(batch, max_iter - done)
        queue(done, count)
        done += count
        if done == max_iter:
            finish_run(done)
        rule, stop_i, n_obj = shard.state()
        for val in shard.objectives(len(history), n_obj - len(history)):
            history.append(np.float64(val))
            if len(history) >= 2 and comm.rank == 0:
                utils.say('[{}]: {:.{}f}'.format(len(history) - 2, val, digits))
    w, h = shard.get_factors()
    if rule:
        if comm.rank == 0:
            utils.convergence_message(rule)
        logging.warning('Converged.')
        return Results(w, h, stop_i, history[:stop_i + 2], experiment)
    return Results(w, h, max_iter - 1, history, experiment)


def _prox_code(kind):
    from . import _lib as L
    if kind in ('nn', 'l1n'):
        return L.PROX[kind]
    if kind in ('l2n', 'l1inf', 'l1inf_transpose'):
        raise NotImplementedError(f"prox '{kind}' is not available in the row-sharded AO-ADMM")
    raise TypeError('Unknown prox_type.')                       # nmf/ao_admm.py:198


def aoadmm_sharded(shard, comm, *, reg_w=(0, 'nn'), reg_h=(0, 'nn'), min_iter=10, max_iter=100000,
                   admm_iter=10, tol1=1e-3, tol2=1e-3, batch=4, experiment=None):
    """AO-ADMM, Euclidean loss (nmf/ao_admm.py:259-301), over a row-sharded V.

    H sub-problem (ao_admm.py:263): [W^T V | W^T W | objective] is all-reduced once, the Cholesky
    solve / prox / dual rounds are then replicated work.  W sub-problem (ao_admm.py:265): rank-local
    rows, except that `terminate` (ao_admm.py:33-43) takes norms over the whole factor: the four
    sums of squares of every round are all-reduced (32 bytes), so all ranks stop at the same
    round, as the reference would.  Returns Results with THIS rank's rows of w."""
    prox_h, prox_w = _prox_code(reg_h[1]), _prox_code(reg_w[1])     # H's regulariser is met first
    x32, x64 = shard.buffers()
    norms = x64[1:5]

    def queue(first, count):
        for j in range(first, first + count):
            shard.ao_h_products(j)
            comm.all_reduce(x32, x64)
            shard.ao_h_solve(prox_h, reg_h[0], admm_iter, min_iter, tol1, tol2, j)
            shard.ao_w_products(min_iter, tol1, tol2, j)
            for rnd in range(admm_iter):
                shard.ao_w_round(prox_w, reg_w[0], rnd)
                comm.all_reduce(norms)
            shard.ao_w_close(admm_iter, j)

    def finish_run(done):
        shard.objective_partial()
        comm.all_reduce(x64)
        shard.finish_b(min_iter, tol1, tol2, done)

    return _sharded_loop(shard, comm, queue, finish_run, max_iter=max_iter, tol1=tol1, tol2=tol2,
                         batch=batch, experiment=experiment)


def anls_sharded(shard, comm, *, lambda_w=0, lambda_h=0, min_iter=10, max_iter=1000, tol1=1e-3, tol2=1e-3,
                 batch=4, experiment=None):
    """ANLS (nmf/anls.py:112-126) over a row-sharded V: the rows of W are independent NNLS
    problems (anls.py:18-31, rank-local), the columns of H need sum_p W_p^T W_p and
    sum_p W_p^T V_p (anls.py:34-47; replicated solve after one all-reduce).  The objective
    partial is all-reduced separately because the stop rule is evaluated BEFORE the updates of
    the iteration are queued.  Returns Results with THIS rank's rows of w."""
    x32, x64 = shard.buffers()

    def queue(first, count):
        for j in range(first, first + count):
            shard.anls_objective(j)
            comm.all_reduce(x64)
            shard.anls_w(lambda_w, min_iter, tol1, tol2, j)
            comm.all_reduce(x32)
            shard.anls_h(lambda_h, j)

    def finish_run(done):
        shard.objective_partial()
        comm.all_reduce(x64)
        shard.finish_b(min_iter, tol1, tol2, done)

    return _sharded_loop(shard, comm, queue, finish_run, max_iter=max_iter, tol1=tol1, tol2=tol2,
                         batch=batch, experiment=experiment)
