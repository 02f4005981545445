"""Thin object wrapper over the C handle (include/nmfx.h)."""
import ctypes as C
import logging

import numpy as np

from . import _lib as L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One device-resident factorisation problem: V (or a row shard of it), W, H
    and all scratch live in HBM for the lifetime of the object."""

    def __init__(self, m, n, k, device=0):
        self.lib = L.require_gpu()
        self.m, self.n, self.k = int(m), int(n), int(k)
        h = C.c_void_p()
        L.check(self.lib.nmfx_create(C.byref(h), int(device), self.m, self.n, self.k))
        self.h = h
        self.precision_epoch = 0
        note = self.lib.nmfx_get_note(h)
        if note:                                  # e.g. the fall back to the exact-f32 kernels for lack of memory
            logging.warning('nmf_amd: %s', note.decode())

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.nmfx_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    @classmethod
    def for_data(cls, x, k, device=0, engine=None):
        """Context manager over the engine a solver runs on: a fresh one with `x` uploaded (closed on
        exit), or `engine` -- one that already holds this V for this k (nmf_amd.grid keeps V resident
        across the factorizations of a parameter grid) -- which is left open."""
        return _EngineScope(cls, x, k, device, engine)

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        L.check(rc, self.h)

    def set_stream(self, stream_ptr):
        self._ck(self.lib.nmfx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_precision(self, mode):
        """'f32' (exact f32 MFMA) or 'bf16' (split-bf16 products: every solver's Euclidean products and MUR-KL's
        quotient products when k pads to 64 or 128)."""
        self._ck(self.lib.nmfx_set_precision(self.h, {"f32": 0, "bf16": 1}[mode]))
        self.precision_epoch += 1           # row-sharded drivers negotiate the collective sequence again (dist.*Shard.negotiate)

    def precision(self):
        return "bf16" if self.lib.nmfx_get_precision(self.h) == 1 else "f32"

    def note(self):
        """What nmfx_create had to say (e.g. the fall back to the exact-f32 kernels for lack of memory), or ''."""
        n = self.lib.nmfx_get_note(self.h)
        return n.decode() if n else ""

    def reset_stream(self):
        self._ck(self.lib.nmfx_reset_stream(self.h))

    def synchronize(self):
        self._ck(self.lib.nmfx_synchronize(self.h))

    # -- data --------------------------------------------------------------
    def upload_v(self, v, row0=0):
        v = np.asarray(v)
        if v.dtype not in (np.float32, np.float64):
            v = v.astype(np.float64)
        if v.ndim != 2 or v.shape[1] != self.n:
            raise ValueError("V block has the wrong shape")
        if not (v.strides[1] == v.itemsize and v.strides[0] % v.itemsize == 0 and v.strides[0] > 0):
            v = np.ascontiguousarray(v)
        ld = v.strides[0] // v.itemsize
        self._ck(self.lib.nmfx_upload_v(self.h, _ptr(v), L.F32 if v.dtype == np.float32 else L.F64,
                                        ld, int(row0), v.shape[0]))

    def upload_v_device(self, ptr, rows, row0=0, dtype=np.float32, ld=None):
        """Rows [row0, row0 + rows) of V from device memory of this GPU (`ptr` = a raw device
        pointer such as torch.Tensor.data_ptr(), row stride `ld` elements, default n).  The
        producer of the block must have finished (torch.cuda.synchronize())."""
        code = L.F32 if np.dtype(dtype) == np.float32 else L.F64
        self._ck(self.lib.nmfx_upload_v_device(self.h, C.c_void_p(int(ptr)), code, int(ld or self.n),
                                               int(row0), int(rows)))

    def set_factors(self, w, h):
        w = np.ascontiguousarray(w, dtype=np.float64)
        h = np.ascontiguousarray(h, dtype=np.float64)
        if w.shape != (self.m, self.k) or h.shape != (self.k, self.n):
            raise ValueError("factor shapes do not match the engine")
        self._ck(self.lib.nmfx_set_factors(self.h, _ptr(w), _ptr(h)))

    def get_factors(self):
        w = np.empty((self.m, self.k), dtype=np.float64)
        h = np.empty((self.k, self.n), dtype=np.float64)
        self._ck(self.lib.nmfx_get_factors(self.h, _ptr(w), _ptr(h)))
        return w, h

    def get_matrix(self, name):
        wlike = name in ("dual_w", "w_aux")
        out = np.empty((self.m, self.k) if wlike else (self.k, self.n), dtype=np.float64)
        self._ck(self.lib.nmfx_get_matrix(self.h, name.encode(), _ptr(out)))
        return out

    def set_matrix(self, name, a):
        wlike = name in ("dual_w", "w_aux")
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != ((self.m, self.k) if wlike else (self.k, self.n)):
            raise ValueError("matrix shape does not match the engine")
        self._ck(self.lib.nmfx_set_matrix(self.h, name.encode(), _ptr(a)))

    def prox_apply(self, side, kind, rho, lam, update_dual=False):
        """x = prox(x_aux, dual) for side 'w' | 'h' with the l1inf operators of nmf/admm.py:158-210."""
        self._ck(self.lib.nmfx_prox_apply(self.h, {"w": 0, "h": 1}[side], L.PROX[kind], float(rho), float(lam),
                                          1 if update_dual else 0))

    # -- state -------------------------------------------------------------
    def state(self):
        rule, stop_i, n_obj = C.c_int(), C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_get_state(self.h, C.byref(rule), C.byref(stop_i), C.byref(n_obj)))
        return rule.value, stop_i.value, n_obj.value

    def objectives(self, first, count):
        out = np.empty(int(count), dtype=np.float64)
        if count:
            self._ck(self.lib.nmfx_get_objectives(self.h, int(first), int(count), _ptr(out)))
        return out

    def inner_counts(self, first, count):
        out = np.zeros((int(count), 2), dtype=np.int32)
        if count:
            self._ck(self.lib.nmfx_get_inner_counts(self.h, int(first), int(count), _ptr(out)))
        return out

    # -- solvers -----------------------------------------------------------
    def mur_run(self, dist, lambda_w, lambda_h, min_iter, tol1, tol2, first, count):
        self._ck(self.lib.nmfx_mur_run(self.h, dist, float(lambda_w), float(lambda_h), int(min_iter),
                                       float(tol1), float(tol2), int(first), int(count)))

    def mur_finish(self, dist, min_iter, tol1, tol2, iters_done):
        self._ck(self.lib.nmfx_mur_finish(self.h, dist, int(min_iter), float(tol1), float(tol2),
                                          int(iters_done)))

    def mur_phase_a(self, dist, lambda_w, j):
        self._ck(self.lib.nmfx_mur_phase_a(self.h, dist, float(lambda_w), int(j)))

    def mur_chunk_info(self, dist):
        """(unit, padded n, padded k) of the chunked phase A; unit = 0: not available for this handle / distance."""
        u, n, k = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_mur_chunk_info(self.h, int(dist), C.byref(u), C.byref(n), C.byref(k)))
        return u.value, n.value, k.value

    def mur_phase_a_head(self, dist, lambda_w, j):
        self._ck(self.lib.nmfx_mur_phase_a_head(self.h, int(dist), float(lambda_w), int(j)))

    def mur_phase_a_cols(self, dist, c0, c1):
        self._ck(self.lib.nmfx_mur_phase_a_cols(self.h, int(dist), int(c0), int(c1)))

    def mur_phase_b(self, dist, lambda_h, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_mur_phase_b(self.h, dist, float(lambda_h), int(min_iter), float(tol1),
                                           float(tol2), int(j)))

    # -- row-sharded AO-ADMM / ANLS phases (include/nmfx.h) --------------------
    def aoadmm_phase_h_products(self, j):
        self._ck(self.lib.nmfx_aoadmm_phase_h_products(self.h, int(j)))

    def aoadmm_phase_h_solve(self, prox_h, lam_h, admm_iter, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_aoadmm_phase_h_solve(self.h, prox_h, float(lam_h), int(admm_iter), int(min_iter),
                                                    float(tol1), float(tol2), int(j)))

    def aoadmm_phase_w_products(self, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_aoadmm_phase_w_products(self.h, int(min_iter), float(tol1), float(tol2), int(j)))

    def aoadmm_phase_w_round(self, prox_w, lam_w, rnd):
        self._ck(self.lib.nmfx_aoadmm_phase_w_round(self.h, prox_w, float(lam_w), int(rnd)))

    def aoadmm_phase_w_close(self, admm_iter, j):
        self._ck(self.lib.nmfx_aoadmm_phase_w_close(self.h, int(admm_iter), int(j)))

    def aoadmm_phase_w_fused(self, prox_w, lam_w, admm_iter):
        self._ck(self.lib.nmfx_aoadmm_phase_w_fused(self.h, prox_w, float(lam_w), int(admm_iter)))

    def aoadmm_phase_w_repair(self, prox_w, lam_w, admm_iter, j):
        self._ck(self.lib.nmfx_aoadmm_phase_w_repair(self.h, prox_w, float(lam_w), int(admm_iter), int(j)))

    # row-sharded AO-ADMM, KL loss: one exchange per inner round (include/nmfx.h)
    def aoadmm_kl_phase_h_products(self, j, rnd):
        self._ck(self.lib.nmfx_aoadmm_kl_phase_h_products(self.h, int(j), int(rnd)))

    def aoadmm_kl_phase_h_round(self, prox_h, lam_h, rnd, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_aoadmm_kl_phase_h_round(self.h, prox_h, float(lam_h), int(rnd), int(min_iter), float(tol1),
                                                       float(tol2), int(j)))

    def aoadmm_kl_phase_h_close(self, admm_iter, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_aoadmm_kl_phase_h_close(self.h, int(admm_iter), int(min_iter), float(tol1), float(tol2), int(j)))

    def aoadmm_kl_phase_w_round(self, prox_w, lam_w, rnd):
        self._ck(self.lib.nmfx_aoadmm_kl_phase_w_round(self.h, prox_w, float(lam_w), int(rnd)))

    def aoadmm_kl_phase_w_close(self, admm_iter, j):
        self._ck(self.lib.nmfx_aoadmm_kl_phase_w_close(self.h, int(admm_iter), int(j)))

    def admm_phase_products(self, dist, rho, prox_w, prox_h, j):
        self._ck(self.lib.nmfx_admm_phase_products(self.h, dist, float(rho), prox_w, prox_h, int(j)))

    def admm_phase_update(self, dist, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_admm_phase_update(self.h, dist, float(rho), prox_w, float(lam_w), prox_h, float(lam_h),
                                                 int(min_iter), float(tol1), float(tol2), int(j)))

    def objective_partial(self):
        self._ck(self.lib.nmfx_objective_partial(self.h))

    def anls_phase_objective(self, j):
        self._ck(self.lib.nmfx_anls_phase_objective(self.h, int(j)))

    def anls_phase_w(self, lam_w, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_anls_phase_w(self.h, float(lam_w), int(min_iter), float(tol1), float(tol2), int(j)))

    def anls_phase_h(self, lam_h, j):
        self._ck(self.lib.nmfx_anls_phase_h(self.h, float(lam_h), int(j)))

    def topk_svd(self, k, block=0, tol=0.0, max_sweeps=0, seed=0):
        """Leading k singular triplets of the uploaded V (f64, on the device).
        Returns (u m x k, s k, vt k x n, sweeps, relative residual)."""
        u = np.empty((self.m, k)); s = np.empty(k); vt = np.empty((k, self.n))
        sweeps, resid = C.c_int(), C.c_double()
        self._ck(self.lib.nmfx_topk_svd(self.h, int(k), int(block), float(tol), int(max_sweeps), int(seed),
                                        _ptr(u), _ptr(s), _ptr(vt), C.byref(sweeps), C.byref(resid)))
        return u, s, vt, sweeps.value, resid.value

    def reserve_objectives(self, count):
        self._ck(self.lib.nmfx_reserve_objectives(self.h, int(count)))

    def shift_iteration_base(self, delta):
        self._ck(self.lib.nmfx_shift_iteration_base(self.h, int(delta)))

    def mur_finish_a(self, dist, j):
        self._ck(self.lib.nmfx_mur_finish_a(self.h, dist, int(j)))

    def mur_finish_b(self, min_iter, tol1, tol2, j):
        self._ck(self.lib.nmfx_mur_finish_b(self.h, int(min_iter), float(tol1), float(tol2), int(j)))

    def aoadmm_run(self, dist, prox_w, lam_w, prox_h, lam_h, admm_iter, min_iter, tol1, tol2, first, count):
        self._ck(self.lib.nmfx_aoadmm_run(self.h, dist, prox_w, float(lam_w), prox_h, float(lam_h),
                                          int(admm_iter), int(min_iter), float(tol1), float(tol2),
                                          int(first), int(count)))

    def aoadmm_finish(self, min_iter, tol1, tol2, done):
        self._ck(self.lib.nmfx_aoadmm_finish(self.h, int(min_iter), float(tol1), float(tol2), int(done)))

    def set_l2n_operator(self, which, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        self._ck(self.lib.nmfx_set_l2n_operator(self.h, int(which), _ptr(p)))

    def admm_run(self, dist, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, first, count):
        self._ck(self.lib.nmfx_admm_run(self.h, dist, float(rho), prox_w, float(lam_w), prox_h,
                                        float(lam_h), int(min_iter), float(tol1), float(tol2),
                                        int(first), int(count)))

    def anls_set_distance(self, dist):
        self._ck(self.lib.nmfx_anls_set_distance(self.h, int(dist)))

    def diagnostics(self):
        """(NNLS variables dropped for a vanished pivot, NNLS solves that hit the iteration cap)."""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_get_diagnostics(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def nnls_fallbacks(self):
        """(problems, half-steps) the inverse + complement NNLS pass left to the elimination kernels."""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_get_nnls_fallbacks(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def inner_paths(self):
        """AO-ADMM fused inner rounds: sub-problems whose first leg (stood, was cut back, was continued, was continued and cut back)."""
        out = (C.c_int64 * 4)()
        self._ck(self.lib.nmfx_get_inner_paths(self.h, out))
        return tuple(int(v) for v in out)

    def anls_run(self, lam_w, lam_h, min_iter, tol1, tol2, first, count):
        self._ck(self.lib.nmfx_anls_run(self.h, float(lam_w), float(lam_h), int(min_iter), float(tol1),
                                        float(tol2), int(first), int(count)))

    # -- the f64 referee of the stop rule (include/nmfx.h) --------------------------------------
    def objective_f64(self):
        """1/2 ||V - W H||^2 of the current pair with product and sum in float64 on the device."""
        out = C.c_double()
        self._ck(self.lib.nmfx_objective_f64(self.h, C.byref(out)))
        return out.value

    def set_stop_guard(self, guard):
        self._ck(self.lib.nmfx_set_stop_guard(self.h, float(guard)))

    def resume(self):
        self._ck(self.lib.nmfx_resume(self.h))

    # -- pair mode: two MUR-eu problems stacked into a k = 128 handle (include/nmfx.h) -----------
    def mur_pair_run(self, lambda_w, lambda_h, min_iter, tol1, tol2, first, count):
        lw = (C.c_double * 2)(float(lambda_w[0]), float(lambda_w[1]))
        lh = (C.c_double * 2)(float(lambda_h[0]), float(lambda_h[1]))
        self._ck(self.lib.nmfx_mur_pair_run(self.h, lw, lh, int(min_iter), float(tol1), float(tol2), int(first), int(count)))

    def mur_pair_finish(self, min_iter, tol1, tol2, done):
        self._ck(self.lib.nmfx_mur_pair_finish(self.h, int(min_iter), float(tol1), float(tol2), int(done)))

    def pair_state(self, p):
        a, b, c = C.c_int(), C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_pair_get_state(self.h, int(p), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def pair_objectives(self, p, first, count):
        out = np.zeros(max(0, int(count)), dtype=np.float64)
        if count > 0:
            self._ck(self.lib.nmfx_pair_get_objectives(self.h, int(p), int(first), int(count), _ptr(out)))
        return out

    def pair_get_factors(self, p, k_p):
        w = np.empty((self.m, k_p), dtype=np.float64)
        h = np.empty((k_p, self.n), dtype=np.float64)
        self._ck(self.lib.nmfx_pair_get_factors(self.h, int(p), int(k_p), _ptr(w), _ptr(h)))
        return w, h

    # -- exchange buffers (row-sharded runs) ---------------------------------
    def set_exchange_rank(self, rank, world):
        """The objective partial travels inside the f32 exchange buffer (one collective per MUR-Euclidean iteration);
        world = 0 switches back.  Raises NmfxError where the engine's path does not support it."""
        self._ck(self.lib.nmfx_set_exchange_rank(self.h, int(rank), int(world)))

    def exchange_sizes(self):
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_exchange_sizes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_exchange_buffers(self, f32_ptr, n_f32, f64_ptr, n_f64):
        self._ck(self.lib.nmfx_set_exchange_buffers(self.h, C.c_void_p(f32_ptr), int(n_f32), C.c_void_p(f64_ptr), int(n_f64)))

    # -- the exchange step behind the C ABI (RCCL, comm.hip) -------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to the other ranks)."""
        buf = C.create_string_buffer(128)
        L.check(L.load().nmfx_comm_unique_id(buf))
        return buf.raw

    def comm_init_rank(self, uid, rank, world):
        self._ck(self.lib.nmfx_comm_init_rank(self.h, C.c_char_p(bytes(uid)), int(rank), int(world)))

    def comm_destroy(self):
        self._ck(self.lib.nmfx_comm_destroy(self.h))

    def comm_info(self):
        """(rank, world, objective partial merged into the f32 exchange, RCCL version code)"""
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self.lib.nmfx_comm_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, bool(c.value), d.value

    def comm_negotiate(self):
        self._ck(self.lib.nmfx_comm_negotiate(self.h))

    def comm_all_reduce(self, which, first, count):
        self._ck(self.lib.nmfx_comm_all_reduce(self.h, int(which), int(first), int(count)))

    def comm_all_min(self, values):
        arr = (C.c_int64 * len(values))(*[int(v) for v in values])
        self._ck(self.lib.nmfx_comm_all_min(self.h, arr, len(values)))
        return list(arr)

    def comm_barrier(self):
        self._ck(self.lib.nmfx_comm_barrier(self.h))

    def comm_set_graph(self, enable):
        self._ck(self.lib.nmfx_comm_set_graph(self.h, 1 if enable else 0))

    def comm_set_exchange(self, mode):
        """0 = all-reduce, 1 = reduce-scatter . sliced H update . all-gather inside nmfx_mur_run_sharded (include/nmfx.h)."""
        self._ck(self.lib.nmfx_comm_set_exchange(self.h, int(mode)))

    def comm_get_exchange(self):
        m = C.c_int()
        self._ck(self.lib.nmfx_comm_get_exchange(self.h, C.byref(m)))
        return m.value

    def mur_slice_info(self, dist, world):
        """(columns of H per rank, f32 elements per rank) of the sliced phase B, or (0, 0) where it is not available."""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.nmfx_mur_slice_info(self.h, int(dist), int(world), C.byref(a), C.byref(b)))
        return a.value, b.value

    def mur_phase_b_slice(self, dist, lambda_h, min_iter, tol1, tol2, j, c0, c1):
        self._ck(self.lib.nmfx_mur_phase_b_slice(self.h, int(dist), float(lambda_h), int(min_iter), float(tol1), float(tol2), int(j),
                                                 int(c0), int(c1)))

    def mur_phase_b_rest(self, dist, c0, c1):
        self._ck(self.lib.nmfx_mur_phase_b_rest(self.h, int(dist), int(c0), int(c1)))

    def comm_graph_replays(self):
        n = C.c_int64()
        self._ck(self.lib.nmfx_comm_graph_replays(self.h, C.byref(n)))
        return n.value

    def mur_run_sharded(self, dist, lambda_w, lambda_h, min_iter, tol1, tol2, first, count):
        self._ck(self.lib.nmfx_mur_run_sharded(self.h, int(dist), float(lambda_w), float(lambda_h), int(min_iter), float(tol1),
                                               float(tol2), int(first), int(count)))

    def mur_finish_sharded(self, dist, min_iter, tol1, tol2, done):
        self._ck(self.lib.nmfx_mur_finish_sharded(self.h, int(dist), int(min_iter), float(tol1), float(tol2), int(done)))

    # -- measurement -------------------------------------------------------
    def profile_enable(self, on=True):
        self._ck(self.lib.nmfx_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._ck(self.lib.nmfx_profile_reset(self.h))

    def profile_repeat(self, which, reps=100, dist=0):
        """ms per launch of the 'wphase' or 'hphase' product kernel over `reps` back-to-back launches."""
        ms = C.c_double()
        self._ck(self.lib.nmfx_profile_repeat(self.h, which.encode(), int(dist), int(reps), C.byref(ms)))
        return ms.value

    def profile_get(self, name):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.lib.nmfx_profile_get(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class _EngineScope:
    def __init__(self, cls, x, k, device, engine):
        self.cls, self.x, self.k, self.device, self.engine = cls, x, k, device, engine
        self.owned = None

    def __enter__(self):
        if self.engine is not None:
            e = self.engine
            if (e.m, e.n, e.k) != (self.x.shape[0], self.x.shape[1], int(self.k)):
                raise ValueError(f"engine holds a {e.m}x{e.n} problem with k={e.k}, not "
                                 f"{self.x.shape[0]}x{self.x.shape[1]} with k={self.k}")
            return e
        self.owned = self.cls(self.x.shape[0], self.x.shape[1], self.k, device=self.device)
        try:
            self.owned.upload_v(self.x)
        except Exception:
            self.owned.close()
            raise
        return self.owned

    def __exit__(self, *exc):
        if self.owned is not None:
            self.owned.close()
