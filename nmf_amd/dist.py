"""Row-sharded MUR over several GPUs of one node (one process per GPU).

Rank p holds the rows [r0, r1) of V and of W; H (k x n) and H H^T are
replicated.  The W update is row-local (nmf/mur.py:29: row i of W depends only
on row i of V).  The H update (nmf/mur.py:45) needs W^T V = sum_p W_p^T V_p and
W^T W = sum_p W_p^T W_p: ONE sum-all-reduce per outer iteration of the packed
f32 buffer [W^T V | W^T W] (k*n + k*k elements), followed on the same stream by
the 64-byte f64 all-reduce of the objective partial (torch cannot coalesce two
dtypes into one RCCL group).
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
        import inspect
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.stage = stage_through_host
        # One RCCL group launch for several reductions of one dtype (ncclGroupStart / End) through torch's coalescing
        # manager.  It is a private API: its presence and keywords are probed ONCE here -- identically on every rank,
        # they run the same torch -- and the mode then stays fixed for the run.  (Falling back in the middle of a call
        # could reduce a tensor twice, or desynchronise the collective sequence between ranks.)
        mgr = getattr(dist, "_coalescing_manager", None)
        self._coalesce = False
        if mgr is not None and not stage_through_host and dist.get_backend(group) == "nccl":
            try:
                params = inspect.signature(mgr).parameters
                self._coalesce = all(name in params for name in ("group", "device", "async_ops"))
            except (TypeError, ValueError):
                self._coalesce = False

    def all_reduce(self, *tensors):
        # torch coalesces tensors of ONE dtype only (allreduce_coalesced: "Tensors must have identical type"): the f32
        # products and the f64 objective partial therefore travel as two collectives, back to back on the stream.
        # (Round 1 wrapped the mixed call in try/except and never noticed that it always fell back.)
        if self._coalesce and len(tensors) > 1 and all(t.is_cuda for t in tensors):
            by_type = {}
            for t in tensors:
                by_type.setdefault(t.dtype, []).append(t)
            for group in by_type.values():
                if len(group) > 1:
                    with self.dist._coalescing_manager(group=self.group, device=group[0].device, async_ops=False):
                        for t in group:
                            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
                else:
                    self.dist.all_reduce(group[0], op=self.dist.ReduceOp.SUM, group=self.group)
            return
        for t in tensors:
            if self.stage and t.is_cuda:
                import torch
                torch.cuda.current_stream().synchronize()
                host = t.cpu()
                self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(host)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_reduce_async(self, tensor):
        """Start the sum-all-reduce of `tensor` and return a handle with wait(): on RCCL the collective runs on the process
        group's own stream behind what the current stream holds so far, and wait() makes the current stream wait for it --
        the kernels queued in between overlap with it.  Host-staged (tests): reduced at once."""
        if self.stage or not tensor.is_cuda:
            self.all_reduce(tensor)
            return None
        return self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_scatter(self, tensor, elems):
        """Sum-reduce-scatter IN PLACE of tensor[0 : world * elems]: afterwards this rank's range [rank * elems, (rank + 1) * elems)
        holds the sum over the ranks of that range (the other ranges are unspecified).  Host-staged / CPU backends (gloo has no
        reduce-scatter): an all-reduce of the whole head, which leaves the same values in the rank's range."""
        head = tensor[:self.world * elems]
        if self.stage or not tensor.is_cuda or self.dist.get_backend(self.group) != "nccl":
            self.all_reduce(head)
            return
        self.dist.reduce_scatter_tensor(head[self.rank * elems:(self.rank + 1) * elems], head, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, tensor, elems):
        """All-gather IN PLACE: every rank's range [rank * elems, (rank + 1) * elems) of `tensor` reaches every rank."""
        head = tensor[:self.world * elems]
        mine = head[self.rank * elems:(self.rank + 1) * elems]
        if self.stage and tensor.is_cuda:
            import torch
            torch.cuda.current_stream().synchronize()
            host = head.cpu()
            parts = [torch.empty(elems, dtype=host.dtype) for _ in range(self.world)]
            self.dist.all_gather(parts, host[self.rank * elems:(self.rank + 1) * elems].clone(), group=self.group)
            head.copy_(torch.cat(parts))
            return
        if not tensor.is_cuda or self.dist.get_backend(self.group) != "nccl":
            parts = [self.torch_empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine.clone(), group=self.group)
            for r, part in enumerate(parts):
                head[r * elems:(r + 1) * elems].copy_(part)
            return
        self.dist.all_gather_into_tensor(head, mine, group=self.group)

    @staticmethod
    def torch_empty_like(t):
        import torch
        return torch.empty_like(t)

    def barrier(self):
        self.dist.barrier(group=self.group)


class DeviceShard:
    """HIP engine for one row shard, with exchange buffers allocated by torch so
    that RCCL can reduce them in place, and running on torch's current stream so
    the collective is ordered after phase A and before phase B."""

    def __init__(self, v_local, k, w0_local, h0, device, shape=None, fill=None):
        """`v_local`: this rank's rows of V (host array) -- or None with `shape = (rows, n)` and `fill(engine)`, a callable
        that hands the rows over on the device (Engine.upload_v_device), for shards that are produced there."""
        import torch
        from .engine import Engine
        self.torch = torch
        torch.cuda.set_device(device)
        rows, n = v_local.shape if v_local is not None else shape
        self.eng = Engine(rows, n, k, device=device)
        n32, n64 = self.eng.exchange_sizes()
        self.xf32 = torch.zeros(n32, dtype=torch.float32, device=f"cuda:{device}")
        self.xf64 = torch.zeros(n64, dtype=torch.float64, device=f"cuda:{device}")
        torch.cuda.synchronize()
        self._merged = False          # negotiate(comm) settles these two with the other ranks
        self._chunk_unit = 0
        self._negotiated = None
        self.eng.set_exchange_buffers(self.xf32.data_ptr(), n32, self.xf64.data_ptr(), n64)
        self.eng.set_stream(torch.cuda.current_stream().cuda_stream)
        if v_local is not None:
            self.eng.upload_v(v_local)
        else:
            fill(self.eng)
        self.eng.set_factors(w0_local, h0)

    def buffers(self):
        return self.xf32, self.xf64

    def negotiate(self, comm):
        """Agree with the other ranks of `comm` on everything that fixes the SEQUENCE of collectives: whether the
        objective partial travels inside the f32 buffer (one all-reduce per MUR-eu iteration instead of two), the column unit of
        the chunked exchange, and the arithmetic mode.  An engine falls back to the exact-f32 kernels on its own when its GPU
        is short of memory for the tile-major copies (nmfx_get_note): such a rank would otherwise send one collective more
        than its peers, or chunks of another size -- a hang or mis-paired reductions.  Every rank offers what IT can do, the
        minimum over the group wins, and ranks whose arithmetic modes differ fail loudly.  Once per (shard, group)."""
        key = (id(comm), self.eng.precision_epoch)     # (Engine.set_precision voids what was settled: ADVICE r3)
        if self._negotiated == key:
            return
        import os
        torch = self.torch
        rank, world = getattr(comm, "rank", 0), getattr(comm, "world", 1)
        bf16 = 1 if self.eng.precision() == "bf16" else 0
        merge = 0
        if os.environ.get("NMFX_DIST_MERGE", "1") != "0" and bf16 and world <= 64:
            merge = 1
        unit = int(self.eng.mur_chunk_info(0)[0] or 0)
        offer = [merge, unit, bf16, -bf16]              # MIN over ranks: [all can merge, smallest unit, all bf16, -(any bf16)]
        if world > 1:
            staged = getattr(comm, "stage", False) or comm.dist.get_backend(comm.group) != "nccl"
            t = torch.tensor(offer, dtype=torch.int64, device="cpu" if staged else self.xf32.device)
            comm.dist.all_reduce(t, op=comm.dist.ReduceOp.MIN, group=comm.group)
            offer = [int(x) for x in t.tolist()]
        if offer[2] != -offer[3]:
            raise RuntimeError("row-sharded run: the ranks run different arithmetic modes (a rank fell back to the exact-f32 kernels: "
                               f"{self.eng.note() or 'NMFX_PRECISION differs'}); set NMFX_PRECISION=f32 on every rank or free its memory")
        self._merged = bool(offer[0])
        self._chunk_unit = offer[1]
        if self._merged:
            self.eng.set_exchange_rank(rank, world)     # (cannot fail: every rank has just said it runs the split-bf16 epilogues)
        else:
            self.eng.set_exchange_rank(0, 0)
        self._negotiated = key

    def merge_objective(self):
        """True when the objective partial of MUR-eu travels inside the f32 exchange buffer (split-bf16 epilogues on EVERY rank, at
        most 64 ranks; NMFX_DIST_MERGE=0 keeps the separate f64 all-reduce).  Settled by negotiate()."""
        return bool(self._merged)

    def phase_a(self, dist_code, lambda_w, j):
        self.eng.mur_phase_a(dist_code, lambda_w, j)

    def phase_b(self, dist_code, lambda_h, min_iter, tol1, tol2, j):
        self.eng.mur_phase_b(dist_code, lambda_h, min_iter, tol1, tol2, j)

    def chunk_ranges(self, dist_code, chunks):
        """Column ranges [(c0, c1, first element, end element of the f32 exchange buffer)] of a phase A in `chunks` pieces,
        or None where the engine has no chunked phase A for this loss / arithmetic or the matrix is too narrow."""
        unit, npad, kpad = self.eng.mur_chunk_info(dist_code)
        unit = min(unit, self._chunk_unit) if unit else 0           # (negotiate(): 0 unless every rank has the chunked phase A)
        if not unit or chunks < 2:
            return None
        step = max(512, -(-npad // chunks) // unit * unit)
        edges = list(range(0, npad, step)) + [npad]
        if len(edges) > 2 and edges[-1] - edges[-2] < 512:          # a short last piece joins its neighbour
            del edges[-2]
        if len(edges) < 3:
            return None
        total = self.xf32.numel()
        return [(c0, c1, c0 * kpad, c1 * kpad if c1 < npad else total) for c0, c1 in zip(edges[:-1], edges[1:])]

    def slice_info(self, dist_code, world):
        """(columns of H per rank, f32 exchange elements per rank) of the reduce-scatter / all-gather form of phase B, or (0, 0)
        where the engine has none for this loss / arithmetic / shape (nmfx_mur_slice_info)."""
        return self.eng.mur_slice_info(dist_code, world)

    def phase_b_slice(self, dist_code, lambda_h, min_iter, tol1, tol2, j, c0, c1):
        self.eng.mur_phase_b_slice(dist_code, lambda_h, min_iter, tol1, tol2, j, c0, c1)

    def phase_b_rest(self, dist_code, c0, c1):
        self.eng.mur_phase_b_rest(dist_code, c0, c1)

    def phase_a_head(self, dist_code, lambda_w, j):
        self.eng.mur_phase_a_head(dist_code, lambda_w, j)

    def phase_a_cols(self, dist_code, c0, c1):
        self.eng.mur_phase_a_cols(dist_code, c0, c1)

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

    def ao_w_fused(self, prox_w, lam_w, admm_iter):
        self.eng.aoadmm_phase_w_fused(prox_w, lam_w, admm_iter)

    def ao_w_repair(self, prox_w, lam_w, admm_iter, j):
        self.eng.aoadmm_phase_w_repair(prox_w, lam_w, admm_iter, j)

    def ao_kl_h_products(self, j, rnd):
        self.eng.aoadmm_kl_phase_h_products(j, rnd)

    def ao_kl_h_round(self, prox_h, lam_h, rnd, min_iter, tol1, tol2, j):
        self.eng.aoadmm_kl_phase_h_round(prox_h, lam_h, rnd, min_iter, tol1, tol2, j)

    def ao_kl_h_close(self, admm_iter, min_iter, tol1, tol2, j):
        self.eng.aoadmm_kl_phase_h_close(admm_iter, min_iter, tol1, tol2, j)

    def ao_kl_w_round(self, prox_w, lam_w, rnd):
        self.eng.aoadmm_kl_phase_w_round(prox_w, lam_w, rnd)

    def ao_kl_w_close(self, admm_iter, j):
        self.eng.aoadmm_kl_phase_w_close(admm_iter, j)

    def admm_products(self, dist_code, rho, prox_w, prox_h, j):
        self.eng.admm_phase_products(dist_code, rho, prox_w, prox_h, j)

    def admm_update(self, dist_code, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j):
        self.eng.admm_phase_update(dist_code, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j)

    def set_l2n_operator(self, which, p):
        self.eng.set_l2n_operator(which, p)

    def anls_set_distance(self, dist_code):
        self.eng.anls_set_distance(dist_code)

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


class ExRef:
    """A range of one of the engine's exchange buffers (which = 0: f32, 1: f64) -- what NativeComm reduces.  Slices like the torch
    tensors the torch path hands around (`x64[:8]`, `x32[e0:e1]`)."""

    def __init__(self, which, first, count):
        self.which, self.first, self.count = which, first, count

    def __getitem__(self, sl):
        a, b, step = sl.indices(self.count)
        if step != 1:
            raise ValueError("contiguous ranges only")
        return ExRef(self.which, self.first + a, max(0, b - a))

    def numel(self):
        return self.count


class NativeShard(DeviceShard):
    """One row shard whose exchange runs BEHIND the C ABI (comm.hip): library-owned exchange buffers, an RCCL communicator on
    the engine's own stream, the sharded MUR loop as one C call (nmfx_mur_run_sharded) -- no torch tensor, stream or collective on
    the data path.  torch.distributed is only the launcher's means of handing rank 0's 128-byte RCCL id to the other ranks
    (`NativeComm.create`)."""

    def __init__(self, v_local, k, w0_local, h0, device, shape=None, fill=None):
        from .engine import Engine
        self.torch = None
        rows, n = v_local.shape if v_local is not None else shape
        self.eng = Engine(rows, n, k, device=device)
        n32, n64 = self.eng.exchange_sizes()
        self.xf32, self.xf64 = ExRef(0, 0, n32), ExRef(1, 0, n64)
        self._merged, self._chunk_unit, self._negotiated = False, 0, None
        if v_local is not None:
            self.eng.upload_v(v_local)
        else:
            fill(self.eng)
        self.eng.set_factors(w0_local, h0)

    def negotiate(self, comm):
        key = (id(comm), self.eng.precision_epoch)     # (nmfx_set_precision voids the library's side as well)
        if self._negotiated == key:
            return
        self.eng.comm_negotiate()            # NmfxError (NMFX_E_STATE) when the ranks' arithmetic modes differ
        self._merged = self.eng.comm_info()[2]
        self._chunk_unit = int(self.eng.mur_chunk_info(0)[0] or 0)       # (the library keeps the negotiated unit itself)
        self._negotiated = key


class NativeComm:
    """The exchange of a NativeShard: nmfx_comm_* (RCCL through the C ABI) on the engine's stream."""
    stage = False

    def __init__(self, shard, rank, world, uid):
        self.eng, self.rank, self.world = shard.eng, rank, world
        self.eng.comm_init_rank(uid, rank, world)

    @classmethod
    def create(cls, shard, group=None):
        """Collective over the torch.distributed job: rank 0 draws the RCCL id, everybody receives it, every rank joins."""
        import torch.distributed as tdist
        rank, world = tdist.get_rank(group), tdist.get_world_size(group)
        box = [shard.eng.comm_unique_id() if rank == 0 else None]
        if world > 1:
            tdist.broadcast_object_list(box, src=0, group=group)
        return cls(shard, rank, world, box[0])

    def all_reduce(self, *refs):
        for r in refs:
            self.eng.comm_all_reduce(r.which, r.first, r.count)

    def all_reduce_async(self, ref):
        self.all_reduce(ref)
        return None

    def barrier(self):
        self.eng.comm_barrier()

    def close(self):
        self.eng.comm_destroy()


def _negotiate(shard, comm):
    if comm is not None and getattr(shard, "negotiate", None) is not None:
        shard.negotiate(comm)


def _mur_buffers(shard, dist_code=None):
    """The exchange of one MUR iteration.  A device shard with the Euclidean loss carries its objective partial inside the f32
    buffer (nmfx_set_exchange_rank: every rank's partial as exact 16-bit digits in its own slot), so ONE all-reduce
    suffices; otherwise the f32 buffer and the head of the f64 buffer (its tail is the norm table of sharded AO-ADMM)."""
    x32, x64 = shard.buffers()
    if dist_code == 0 and getattr(shard, "merge_objective", None) is not None and shard.merge_objective():
        return (x32,)
    return x32, x64[:8]


def _exchange_chunks():
    """NMFX_DIST_CHUNKS=n (default 1 = off): phase A of sharded MUR-eu in n column chunks, the all-reduce of each chunk
    running while the next one is computed.  Opt-in: what it gains depends on how many CUs the collective needs beside a
    product kernel that fills all of them, which only a multi-GPU box can settle."""
    import os
    try:
        return max(1, int(os.environ.get("NMFX_DIST_CHUNKS", "1")))
    except ValueError:
        return 1


def exchange_mode():
    """NMFX_DIST_EXCHANGE: "allreduce" (default: one sum-all-reduce of [W^T V | W^T W | objective] per iteration) or "rsag" (SURVEY 8e:
    reduce-scatter of the W^T V part, every rank updates ITS n / world columns of H, all-gather -- what the all-reduce does on xGMI's
    point-to-point links anyway, with the replicated H update cut to a slice in between).  The same on every rank."""
    import os
    val = os.environ.get("NMFX_DIST_EXCHANGE", "allreduce").strip().lower()
    if val in ("", "0", "allreduce", "all_reduce", "ar"):
        return "allreduce"
    if val in ("1", "rsag", "rs+ag", "reduce_scatter"):
        return "rsag"
    raise ValueError(f"NMFX_DIST_EXCHANGE={val!r}: 'allreduce' or 'rsag'")


def _slice_plan(shard, comm, dist_code):
    """(c0, c1, elems) of this rank's column slice for the reduce-scatter / all-gather exchange, or None (mode off, a shard or
    communicator without it, or the engine has no sliced phase B for this run: every rank then takes the all-reduce -- the answer
    depends on the shape, the loss and what negotiate() settled, which are the same everywhere)."""
    if exchange_mode() != "rsag" or getattr(shard, "slice_info", None) is None or not hasattr(comm, "reduce_scatter"):
        return None
    world, rank = getattr(comm, "world", 1), getattr(comm, "rank", 0)
    cols, elems = shard.slice_info(dist_code, world)
    if not cols:
        return None
    return rank * cols, (rank + 1) * cols, elems


def _iteration(shard, comm, bufs, plan, dist_code, lambda_w, lambda_h, min_iter, tol1, tol2, j):
    """One sharded outer iteration, exchange in one piece: all-reduce, or reduce-scatter . slice . all-gather . rest."""
    shard.phase_a(dist_code, lambda_w, j)
    if plan is None:
        comm.all_reduce(*bufs)
        shard.phase_b(dist_code, lambda_h, min_iter, tol1, tol2, j)
        return
    c0, c1, elems = plan
    x32 = bufs[0]
    world = getattr(comm, "world", 1)
    comm.reduce_scatter(x32, elems)
    comm.all_reduce(x32[world * elems:], *bufs[1:])               # W^T W, the objective digits (and the f64 partial where not merged)
    shard.phase_b_slice(dist_code, lambda_h, min_iter, tol1, tol2, j, c0, c1)
    comm.all_gather(x32, elems)
    shard.phase_b_rest(dist_code, c0, c1)


def run_iterations(shard, comm, dist_code, lambda_w, lambda_h, min_iter, tol1, tol2, first, count, chunks=None):
    """Queue `count` sharded outer iterations (no host sync)."""
    _negotiate(shard, comm)
    bufs = _mur_buffers(shard, dist_code)
    plan = _slice_plan(shard, comm, dist_code)
    if plan is not None:
        for j in range(first, first + count):
            _iteration(shard, comm, bufs, plan, dist_code, lambda_w, lambda_h, min_iter, tol1, tol2, j)
        return
    chunks = _exchange_chunks() if chunks is None else chunks
    ranges = None
    if chunks > 1 and dist_code == 0 and getattr(shard, "chunk_ranges", None) is not None and hasattr(comm, "all_reduce_async"):
        ranges = shard.chunk_ranges(dist_code, chunks)
    if ranges is None:
        for j in range(first, first + count):
            shard.phase_a(dist_code, lambda_w, j)
            comm.all_reduce(*bufs)
            shard.phase_b(dist_code, lambda_h, min_iter, tol1, tol2, j)
        return
    x32 = bufs[0]
    for j in range(first, first + count):
        shard.phase_a_head(dist_code, lambda_w, j)
        pending = []
        for c0, c1, e0, e1 in ranges:
            shard.phase_a_cols(dist_code, c0, c1)
            pending.append(comm.all_reduce_async(x32[e0:e1]))
        if len(bufs) > 1:                                            # (the objective partial when it is not inside the f32 buffer)
            comm.all_reduce(*bufs[1:])
        for work in pending:
            if work is not None:
                work.wait()
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
        _negotiate(shard, comm)
        bufs = _mur_buffers(shard, dist_code)
        plan = _slice_plan(shard, comm, dist_code)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            # thread_local: the process group's watchdog thread may touch the runtime meanwhile
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                shard.eng.set_stream(torch.cuda.current_stream().cuda_stream)
                for j in (0, 1):
                    _iteration(shard, comm, bufs, plan, dist_code, lambda_w, lambda_h, min_iter, tol1, tol2, j)
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
        self.native = isinstance(comm, NativeComm)
        if self.native:
            # the whole loop is one C call per batch (nmfx_mur_run_sharded); hipGraph replay is the library's business too
            if graph is None:
                graph = os.environ.get("NMFX_DIST_GRAPH", "0") == "1"
            shard.negotiate(comm)
            shard.eng.comm_set_exchange(1 if exchange_mode() == "rsag" else 0)
            shard.eng.comm_set_graph(bool(graph) and max_iter >= 4)
            self.want_graph, self.graph = False, None
            self.mode = "native-hipgraph" if (graph and max_iter >= 4) else "native"
            return
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
        if self.native:
            self.shard.eng.comm_set_graph(False)
            self.shard.eng.mur_run_sharded(d, lw, lh, mi, t1, t2, first, count)
            self.shard.eng.comm_set_graph(self.mode == "native-hipgraph")
            return
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
        if self.native:
            d, lw, lh, mi, t1, t2 = self.args
            self.shard.eng.mur_run_sharded(d, lw, lh, mi, t1, t2, first, count)
            return
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
    if isinstance(comm, NativeComm):
        shard.eng.mur_finish_sharded(dist_code, min_iter, tol1, tol2, done)
        return
    shard.finish_a(dist_code, done)
    comm.all_reduce(_mur_buffers(shard)[1])
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
    _negotiate(shard, comm)
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
    from .ao_admm import _prox_code as single        # same codes and the reference's errors ('l2n': ValueError, 'l1inf*': LinAlgError)
    return single(kind)


MAX_FUSED_ROUNDS = 64         # rows of the norm table in the engine's f64 exchange buffer


def aoadmm_sharded(shard, comm, *, distance_type='eu', reg_w=(0, 'nn'), reg_h=(0, 'nn'), min_iter=10, max_iter=100000,
                   admm_iter=10, tol1=1e-3, tol2=1e-3, batch=4, experiment=None, fused=None):
    """AO-ADMM (nmf/ao_admm.py:259-301) over a row-sharded V.  Euclidean loss:

    H sub-problem (ao_admm.py:263): [W^T V | W^T W | objective] is all-reduced once, the Cholesky
    solve / prox / dual rounds are then replicated work.  W sub-problem (ao_admm.py:265): rank-local
    rows, except that `terminate` (ao_admm.py:33-43) takes norms over the whole factor: all rounds run
    speculatively, the [admm_iter x 4] table of their sums of squares is all-reduced ONCE and every rank
    derives the same stopping round from it and repairs its rows if that round is not the last
    (nmfx_aoadmm_phase_w_fused / _repair; 3 collectives per outer iteration).

    KL loss (admm_kl_update, ao_admm.py:71-101): the V-sized products w.T @ (v_aux + dual_v) and (v_aux + dual_v) @ h.T sit
    INSIDE the inner rounds, so every round of the H sub-problem all-reduces [W^T S | W^T W] and every round of the W
    sub-problem its four norm sums; v_aux and dual_v (m x n) are rank-local rows.  Returns Results with THIS rank's rows
    of w."""
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')
    prox_h, prox_w = _prox_code(reg_h[1]), _prox_code(reg_w[1])     # H's regulariser is met first
    x32, x64 = shard.buffers()
    norms = x64[1:5]
    if distance_type == 'kl':
        def queue_kl(first, count):
            for j in range(first, first + count):
                for rnd in range(admm_iter):
                    shard.ao_kl_h_products(j, rnd)
                    if rnd == 0:
                        comm.all_reduce(x32, x64[:8])
                    else:
                        comm.all_reduce(x32)
                    shard.ao_kl_h_round(prox_h, reg_h[0], rnd, min_iter, tol1, tol2, j)
                shard.ao_kl_h_close(admm_iter, min_iter, tol1, tol2, j)
                for rnd in range(admm_iter):
                    shard.ao_kl_w_round(prox_w, reg_w[0], rnd)
                    comm.all_reduce(norms)
                shard.ao_kl_w_close(admm_iter, j)

        def finish_kl(done):
            shard.objective_partial()
            comm.all_reduce(x64[:8])
            shard.finish_b(min_iter, tol1, tol2, done)

        return _sharded_loop(shard, comm, queue_kl, finish_kl, max_iter=max_iter, tol1=tol1, tol2=tol2,
                             batch=batch, experiment=experiment)
    # ONE exchange for the whole W sub-problem where the shard can run its rounds speculatively (the HIP engine, for
    # 2 <= admm_iter <= 64): the [admm_iter x 4] table of norm sums.  `fused=False` keeps one exchange per round.
    if fused is None:
        # (beyond 128 components the device composes the iteration from generic kernels -- r4 -- and exchanges the norm sums of
        #  the W sub-problem round by round: the speculative fused rounds keep a k-wide panel on chip)
        small_k = getattr(getattr(shard, "eng", None), "k", 0) <= 128
        fused = hasattr(shard, "ao_w_fused") and 2 <= admm_iter <= MAX_FUSED_ROUNDS and small_k
    table = x64[8:8 + 4 * admm_iter] if fused else None

    def queue(first, count):
        for j in range(first, first + count):
            shard.ao_h_products(j)
            comm.all_reduce(x32, x64[:8])
            shard.ao_h_solve(prox_h, reg_h[0], admm_iter, min_iter, tol1, tol2, j)
            shard.ao_w_products(min_iter, tol1, tol2, j)
            if fused:
                shard.ao_w_fused(prox_w, reg_w[0], admm_iter)
                comm.all_reduce(table)
                shard.ao_w_repair(prox_w, reg_w[0], admm_iter, j)
            else:
                for rnd in range(admm_iter):
                    shard.ao_w_round(prox_w, reg_w[0], rnd)
                    comm.all_reduce(norms)
                shard.ao_w_close(admm_iter, j)

    def finish_run(done):
        shard.objective_partial()
        comm.all_reduce(x64[:8])
        shard.finish_b(min_iter, tol1, tol2, done)

    return _sharded_loop(shard, comm, queue, finish_run, max_iter=max_iter, tol1=tol1, tol2=tol2,
                         batch=batch, experiment=experiment)


def admm_sharded(shard, comm, *, rho=1, distance_type='eu', reg_w=(0, 'nn'), reg_h=(0, 'nn'), min_iter=10,
                 max_iter=100000, tol1=1e-3, tol2=1e-3, batch=8, experiment=None):
    """ADMM (nmf/admm.py:292-334), Euclidean or KL loss, over a row-sharded V: h_aux solves the shifted Gram system of
    sum_p w_aux_p^T w_aux_p with the right-hand side sum_p w_aux_p^T V_p (KL: V replaced by v_aux + dual_v, whose rows
    are rank-local) -- ONE all-reduce per outer iteration, then replicated work; w_aux, both prox operators and the
    dual updates are row-local.  reg_h may be any of 'nn', 'l1n', 'l2n', 'l1inf', 'l1inf_transpose' (replicated);
    reg_w 'nn', 'l1n', 'l2n' (the l1inf operators couple all rows of W).  Returns Results with THIS rank's rows of w."""
    from . import _lib as L
    from .admm import l2n_operator
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')
    for kind in (reg_h[1], reg_w[1]):
        if kind not in L.PROX:
            raise TypeError('Unknown prox_type.')               # nmf/admm.py:213
    if reg_w[1] in ('l1inf', 'l1inf_transpose'):
        raise NotImplementedError(f"prox '{reg_w[1]}' on W couples all rows and is not available row-sharded")
    code = 0 if distance_type == 'eu' else 1
    prox_h, prox_w = L.PROX[reg_h[1]], L.PROX[reg_w[1]]
    k = shard.get_factors()[1].shape[0]
    if reg_w[1] == 'l2n':
        shard.set_l2n_operator(0, l2n_operator(k, rho, reg_w[0]))
    if reg_h[1] == 'l2n':
        shard.set_l2n_operator(1, l2n_operator(k, rho, reg_h[0]))
    x32, x64 = shard.buffers()

    def queue(first, count):
        for j in range(first, first + count):
            shard.admm_products(code, rho, prox_w, prox_h, j)
            comm.all_reduce(x32, x64[:8])
            shard.admm_update(code, rho, prox_w, reg_w[0], prox_h, reg_h[0], min_iter, tol1, tol2, j)

    def finish_run(done):
        shard.objective_partial()
        comm.all_reduce(x64[:8])
        shard.finish_b(min_iter, tol1, tol2, done)

    return _sharded_loop(shard, comm, queue, finish_run, max_iter=max_iter, tol1=tol1, tol2=tol2,
                         batch=batch, experiment=experiment)


def anls_sharded(shard, comm, *, distance_type='eu', lambda_w=0, lambda_h=0, min_iter=10, max_iter=1000, tol1=1e-3,
                 tol2=1e-3, batch=4, experiment=None):
    """ANLS (nmf/anls.py:112-126) over a row-sharded V: the rows of W are independent NNLS
    problems (anls.py:18-31, rank-local), the columns of H need sum_p W_p^T W_p and
    sum_p W_p^T V_p (anls.py:34-47; replicated solve after one all-reduce).  The objective
    partial is all-reduced separately because the stop rule is evaluated BEFORE the updates of
    the iteration are queued.  Returns Results with THIS rank's rows of w."""
    if distance_type not in ('eu', 'kl'):
        raise KeyError('Distance type unknown: use "kl" or "eu"')
    if hasattr(shard, "anls_set_distance"):
        shard.anls_set_distance(0 if distance_type == 'eu' else 1)
    elif distance_type != 'eu':
        raise NotImplementedError("this shard type reports the Euclidean objective only")
    x32, x64 = shard.buffers()

    def queue(first, count):
        for j in range(first, first + count):
            shard.anls_objective(j)
            comm.all_reduce(x64[:8])
            shard.anls_w(lambda_w, min_iter, tol1, tol2, j)
            comm.all_reduce(x32)
            shard.anls_h(lambda_h, j)

    def finish_run(done):
        shard.objective_partial()
        comm.all_reduce(x64[:8])
        shard.finish_b(min_iter, tol1, tol2, done)

    return _sharded_loop(shard, comm, queue, finish_run, max_iter=max_iter, tol1=tol1, tol2=tol2,
                         batch=batch, experiment=experiment)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's call surface for a matrix sharded over the GPUs of one node
# ---------------------------------------------------------------------------------------------------------------------
_SOLVERS = {'mur': mur_sharded, 'ao_admm': aoadmm_sharded, 'admm': admm_sharded, 'anls': anls_sharded}


def _defaults(method):
    """Keyword defaults and the Experiment tuple of the single-GPU solver with the same name (= the reference's)."""
    import inspect
    from importlib import import_module
    mod = import_module('.' + method, __package__)
    sig = inspect.signature(getattr(mod, method))
    kw = {name: p.default for name, p in sig.parameters.items()
          if p.kind is p.KEYWORD_ONLY and name not in ('device', 'engine', 'save_dir')}
    return mod, kw


def _experiment(mod, method, k, kw):
    if method == 'mur':
        return mod.Experiment('mur', k, kw['distance_type'], kw['nndsvd_init'], kw['max_iter'], kw['tol1'], kw['tol2'],
                              kw['lambda_w'], kw['lambda_h'])
    if method == 'anls':
        return mod.Experiment('anls', k, kw['distance_type'], kw['nndsvd_init'], kw['max_iter'], kw['tol1'], kw['tol2'],
                              kw['lambda_w'], kw['lambda_h'], kw['use_fcnnls'])
    if method == 'admm':
        return mod.Experiment('admm', k, kw['rho'], kw['distance_type'], kw['nndsvd_init'], kw['min_iter'], kw['max_iter'],
                              kw['tol1'], kw['tol2'], kw['reg_w'][0], kw['reg_w'][1], kw['reg_h'][0], kw['reg_h'][1])
    return mod.Experiment('ao_admm', k, kw['distance_type'], kw['nndsvd_init'], kw['min_iter'], kw['max_iter'],
                          kw['admm_iter'], kw['tol1'], kw['tol2'], kw['reg_w'][0], kw['reg_w'][1], kw['reg_h'][0],
                          kw['reg_h'][1])


def init_process_group(backend=None):
    """Join the torch.distributed job this process was started in (torchrun / torch.distributed.run sets RANK,
    WORLD_SIZE, LOCAL_RANK, MASTER_*): "nccl" (= RCCL over xGMI) with one GPU per rank by default.  Call it -- or
    `factorize`, which calls it -- BEFORE anything touches a GPU.  Returns (rank, world, local_rank)."""
    import os
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not dist.is_initialized():
        # NMFX_DIST_INIT_METHOD: a torch.distributed init_method URL (e.g. file:///shared/path) instead of the MASTER_* variables
        method = os.environ.get("NMFX_DIST_INIT_METHOD") or None
        if method is None:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.device_count() > 0 else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", init_method=method, rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, init_method=method, rank=rank, world_size=world)
    return dist.get_rank(), dist.get_world_size(), local


def factorize(data, k, method='mur', *, gather=True, device=None, backend=None, shard_factory=None, **method_params):
    """`NMF(data, k).factorize(method=...)` of the reference (nmf/nmf.py:48) for a matrix whose rows are sharded over
    the GPUs of one node -- north_star's "V too large for one GPU".  SPMD: every rank of a torch.distributed job calls
    it with the same arguments,

        torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 my_script.py
            np.random.seed(0)                                  # the SAME seed on every rank (reference: global numpy RNG)
            res = nmf_amd.dist.factorize(np.load('v.npy', mmap_mode='r'), 64, method='mur', distance_type='eu')

    `data`: the full m x n array on every rank (a read-only np.memmap is enough: rank p only reads its rows
    [m p / P, m (p + 1) / P), plus all of it on rank 0 when an NNDSVD start is asked for).  Keyword arguments and
    their defaults are those of the reference's solver function of that name (nmf/mur.py:52, nmf/anls.py:50,
    nmf/admm.py:233, nmf/ao_admm.py:201).  Returns the reference's Results(w, h, i, obj_history, experiment): with
    gather=True `w` is the full m x k factor on rank 0 (its own row block on the other ranks), h / i / obj_history are
    identical everywhere.  Exchange per outer iteration: one RCCL all-reduce of [W^T V | W^T W] (+ objective) -- MUR,
    ADMM, ANLS -- or three (AO-ADMM: + the H products' objective, the norm table of the W sub-problem); AO-ADMM with the
    KL loss exchanges in every inner round (its V-sized products sit inside the rounds, nmf/ao_admm.py:71-101).

    Not available sharded (NotImplementedError): prox 'l1inf*' on W (couples all rows).  The in-place lift of
    negative data (nmf/mur.py:99-101) is applied to a private copy of the rank's rows when `data` is not writeable."""
    import torch
    import torch.distributed as tdist
    if method not in _SOLVERS:
        raise Exception('Method not known. Choose one from: mur anls admm ao_admm')      # nmf/nmf.py:76
    rank, world, local = init_process_group(backend)
    mod, kw = _defaults(method)
    unknown = set(method_params) - set(kw)
    if unknown:
        raise TypeError(f"{method}() got an unexpected keyword argument '{sorted(unknown)[0]}'")
    kw.update(method_params)
    experiment = _experiment(mod, method, k, kw)
    m, n = data.shape
    if world > m:
        raise ValueError('more ranks than rows')
    r0, r1 = row_range(m, rank, world)
    on_gpu = tdist.get_backend() == "nccl"
    dev = torch.device(f"cuda:{local if device is None else device}") if on_gpu else torch.device("cpu")
    comm = TorchComm(stage_through_host=(shard_factory is None and not on_gpu))
    v_local = data[r0:r1]
    if method == 'mur':                                       # nmf/mur.py:99-101, with the minimum over ALL ranks' rows
        low = torch.tensor([float(np.min(v_local))], dtype=torch.float64, device=dev)
        tdist.all_reduce(low, op=tdist.ReduceOp.MIN)
        if float(low) < 0:
            if not (isinstance(v_local, np.ndarray) and v_local.flags.writeable):
                v_local = np.array(v_local)
            v_local += abs(float(low))
            logging.info('Data elevated by {}.'.format(abs(float(low))))
    # initial factors, in the reference's RNG order; NNDSVD from rank 0's view of the whole matrix
    nndsvd_init = kw['nndsvd_init']
    if nndsvd_init[0]:
        if rank == 0:
            full = np.asarray(data)
            if method == 'mur' and float(low) < 0:             # (the lifted matrix, as the reference's nndsvd would see it)
                full = full + abs(float(low))
            if utils.nndsvd_on_device(full, k) and shard_factory is None:
                from .engine import Engine
                with Engine(m, n, k, device=dev.index or 0) as tmp:
                    tmp.upload_v(full)
                    w0, h0 = utils.nndsvd_device(tmp, full, k, variant=nndsvd_init[1])
            else:
                w0, h0 = utils.nndsvd(np.asarray(full, dtype=np.float64), k, variant=nndsvd_init[1])
            w0, h0 = np.ascontiguousarray(w0), np.ascontiguousarray(h0)
        else:
            w0, h0 = np.empty((m, k)), np.empty((k, n))
        for a in (w0, h0):
            t = torch.from_numpy(a).to(dev)
            tdist.broadcast(t, src=0)
            a[...] = t.cpu().numpy()
    else:
        w0, h0 = utils.initial_factors(_Shape(m, n), k, nndsvd_init, uniform=(method == 'anls'))
    # The exchange: torch.distributed's collectives on torch's stream between the phase calls (the default), or -- NMFX_DIST_NATIVE=1 --
    # behind the C ABI (NativeShard / NativeComm: the engine's own RCCL communicator and stream, the MUR loop as one C call).  The
    # native path stays OPT-IN until a run with more than one rank on RCCL has been recorded (it has only ever run with a
    # world of one), and every rank takes the same branch: the choice is all-reduced before and after the communicator comes up.
    if shard_factory is not None:
        shard = shard_factory(v_local, k, w0[r0:r1], h0)
    else:
        shard, comm, _ = make_sharded(lambda cls: cls(v_local, k, w0[r0:r1], h0, dev.index or 0), rank, dev, on_gpu)
    run_kw = {key: val for key, val in kw.items()
              if key not in ('nndsvd_init', 'use_fcnnls')}
    try:
        res = _SOLVERS[method](shard, comm, experiment=experiment, **run_kw)
        w = res.w
        if gather and world > 1:
            rows = max(row_range(m, p, world)[1] - row_range(m, p, world)[0] for p in range(world))
            mine = torch.zeros(rows, k, dtype=torch.float64, device=dev)
            mine[:r1 - r0] = torch.from_numpy(np.ascontiguousarray(res.w)).to(dev)
            parts = [torch.empty_like(mine) for _ in range(world)]
            tdist.all_gather(parts, mine)
            if rank == 0:
                w = np.concatenate([parts[p][:row_range(m, p, world)[1] - row_range(m, p, world)[0]].cpu().numpy()
                                    for p in range(world)])
    finally:
        if hasattr(shard, "close"):
            shard.close()
    return Results(w, res.h, res.i, res.obj_history, experiment)


def make_sharded(build, rank, dev, on_gpu, want_native=None, break_native=False):
    """This rank's shard and its exchange: (shard, comm, name of the loop).  `build(cls)` makes the shard (DeviceShard or
    NativeShard).  Native = RCCL behind the C ABI (nmfx_comm_init_rank + nmfx_mur_run_sharded, no torch on the data path): taken when
    `want_native` (default: NMFX_DIST_NATIVE=1) and EVERY rank can bind RCCL and bring its communicator up -- a rank that raised
    before a broadcast or inside ncclCommInitRank would leave its peers blocked, so the outcome of each step is MIN-all-reduced and
    all ranks fall back to torch.distributed's collectives together."""
    import os
    import sys
    import torch
    import torch.distributed as tdist
    if want_native is None:
        want_native = os.environ.get("NMFX_DIST_NATIVE", "0") == "1"
    want = bool(on_gpu and want_native)

    def all_ok(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        return int(flag.item()) == 1

    if want:
        ok = True
        try:
            from .engine import Engine
            Engine.comm_unique_id()                     # binds RCCL (dlopen) or raises
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"rank {rank}: RCCL behind the C ABI is not available ({e}); torch.distributed's collectives instead\n")
            ok = False
        want = all_ok(ok)
    if want:
        shard = build(NativeShard)
        comm, ok = None, True
        try:
            if break_native:                            # (rehearsal of this fall-back)
                raise RuntimeError("native communicator disabled for a rehearsal")
            comm = NativeComm.create(shard)             # ncclCommInitRank on every rank
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"rank {rank}: the communicator behind the C ABI did not come up ({e}); torch.distributed's collectives instead\n")
            ok = False
        if all_ok(ok):
            return shard, comm, "native (nmfx_mur_run_sharded: RCCL behind the C ABI)"
        if comm is not None:                            # some other rank failed: everybody takes the torch path
            comm.close()
        shard.eng.close()
    shard = build(DeviceShard)
    return shard, TorchComm(stage_through_host=not on_gpu), "torch.distributed collectives between the phase calls"


class _Shape:
    """Just enough of an array for utils.initial_factors' random branch (it only reads .shape)."""

    def __init__(self, m, n):
        self.shape = (m, n)
