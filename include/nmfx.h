/*
 * nmfx.h -- C ABI of the MI355X-native NMF solver engine (libnmfx.so).
 *
 * The reference (raleng/nmf) is pure Python and has no FFI of its own; its
 * boundary is the Python call surface NMF(data, k).factorize(method=...) and the
 * four solver functions it forwards to.  This header is the native boundary a
 * maintainer would bind (ctypes stub in INTEGRATION.md) to replace the bodies of
 * those solver loops.  Each entry point names the reference lines it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *  - plain C, opaque handle, plain pointers and sizes; no torch / numpy types.
 *  - host matrices are C-contiguous row-major like the reference's ndarrays.
 *    Factors cross the boundary as float64 (the reference's dtype for W/H,
 *    nmf/utils.py:51-52); the engine computes in float32 on MFMA and keeps
 *    objective sums in float64.
 *  - every function returns 0 or a negative NMFX_E_* code; nmfx_last_error()
 *    gives the message.  One handle = one device + one stream; a handle is not
 *    thread-safe, independent handles are.
 *  - the caller owns all host buffers, the library owns all device buffers
 *    (except exchange buffers handed in with nmfx_set_exchange_buffers).
 *  - device-side control flow: a stop flag in device memory is set by the
 *    engine when the reference's convergence_check (nmf/utils.py:4-15) fires;
 *    all later launches become no-ops, so iterations can be queued in batches
 *    without a host round trip per iteration.
 */
#ifndef NMFX_H
#define NMFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nmfx_engine* nmfx_handle_t;

enum {
    NMFX_OK = 0,
    NMFX_E_ARG = -1,     /* bad argument / unsupported request (k > 128 in a row-sharded AO-ADMM / ADMM / ANLS phase) */
    NMFX_E_HIP = -2,     /* HIP runtime error, no device                          */
    NMFX_E_NOTPD = -3,   /* Gram + rho I not positive definite (scipy LinAlgError, nmf/ao_admm.py:55) */
    NMFX_E_STATE = -4,   /* call sequence error (no V uploaded, no factors set)   */
    NMFX_E_NOMEM = -5,
    NMFX_E_RCCL = -6     /* RCCL not found (dlopen) or a collective / communicator call failed */
};

enum { NMFX_F32 = 0, NMFX_F64 = 1 };                 /* host dtype of V          */
enum { NMFX_EU = 0, NMFX_KL = 1 };                   /* distance_type            */
enum { NMFX_PROX_NN = 0, NMFX_PROX_L1N = 1, NMFX_PROX_L2N = 2,      /* reg type   */
       NMFX_PROX_L1INF = 3, NMFX_PROX_L1INF_T = 4 };                /* ADMM only (nmf/admm.py:158-210) */

/* ---- lifecycle ---------------------------------------------------------- */
/* m, n: rows/cols of the LOCAL block of V held by this handle (all of V on one
 * GPU; a row shard when the caller shards rows over ranks).  k <= 4096 (nmf/nmf.py:32-35 accepts any `factors`): up to 128
 * every solver runs its tuned kernels; beyond that (padded to a multiple of 128) the MUR entry points and the whole-loop
 * entry points nmfx_aoadmm_run / nmfx_admm_run / nmfx_anls_run compose their iterations from a generic exact-f32 product
 * kernel (kernels_generic.hip); the row-sharded phase entry points of AO-ADMM / ADMM / ANLS stay at k <= 128. */
int nmfx_create(nmfx_handle_t* out, int device, int64_t m, int64_t n, int k);
int nmfx_destroy(nmfx_handle_t h);
const char* nmfx_last_error(nmfx_handle_t h);        /* h may be NULL            */
int nmfx_version(void);
int nmfx_device_count(void);
/* Run on a caller-provided hipStream_t (e.g. torch's current stream) instead of
 * the handle's own non-blocking stream.  NULL is HIP's default (null) stream --
 * which is what torch.cuda.current_stream() is unless the caller changed it.
 * nmfx_reset_stream goes back to the handle's own stream.                      */
int nmfx_set_stream(nmfx_handle_t h, void* hip_stream);
int nmfx_reset_stream(nmfx_handle_t h);
int nmfx_synchronize(nmfx_handle_t h);
/* Arithmetic of the V-sized products: 0 = f32-input MFMA (exact
 * f32 FMA chains), 1 = split bf16 (each f32 operand as bf16 hi + bf16 lo, three or four
 * bf16 MFMA terms per product -- kernels_bf16.hip, top -- f32 accumulation; available when
 * k pads to 64 or 128, otherwise mode 0 is used; every solver's Euclidean products and
 * MUR-KL's quotient products).  nmfx_get_precision returns the mode in effect.
 * Environment override at create time: NMFX_PRECISION=f32|bf16.                 */
int nmfx_set_precision(nmfx_handle_t h, int mode);
int nmfx_get_precision(nmfx_handle_t h);
/* What nmfx_create decided on its own, in words ("" if nothing): today the fall back to the exact-f32 kernels when the
 * two extra V-sized buffers of the split-bf16 path do not fit into the free device memory.  (In split-bf16 mode the
 * row-major copy of a V of 4 GiB or more is freed once the tile-major copies exist -- NMFX_DROP_V=0/1 overrides -- and
 * rebuilt when a kernel needs it.)                                                                                  */
const char* nmfx_get_note(nmfx_handle_t h);

/* ---- data --------------------------------------------------------------- */
/* Copy rows [row0, row0+rows) of the local V from host memory (row stride `ld`
 * elements, dtype NMFX_F32/F64).  V is borrowed, never modified (the MUR shift
 * of negative data, nmf/mur.py:99-101, is done by the caller on its array).   */
int nmfx_upload_v(nmfx_handle_t h, const void* host, int dtype, int64_t ld,
                  int64_t row0, int64_t rows);
/* The same from DEVICE memory of the handle's GPU (a block the caller produced there, e.g. a
 * torch tensor's data_ptr): device-to-device copy into the engine's padded layout, no PCIe.
 * The caller makes sure the producer of `dev` has finished (the copy runs on the handle's
 * stream); the call returns when the copy is done, `dev` may then be freed.                 */
int nmfx_upload_v_device(nmfx_handle_t h, const void* dev, int dtype, int64_t ld,
                         int64_t row0, int64_t rows);
/* W (m x k) and H (k x n), float64 row-major; either may be NULL to skip.
 * set_factors also zeroes all dual/auxiliary state and the iteration state.   */
int nmfx_set_factors(nmfx_handle_t h, const double* w, const double* hmat);
int nmfx_get_factors(nmfx_handle_t h, double* w, double* hmat);
/* Other state matrices by name: "dual_w" "dual_h" (AO-ADMM / ADMM),
 * "w_aux" "h_aux" (ADMM).  Shapes as W / H.                                   */
int nmfx_get_matrix(nmfx_handle_t h, const char* name, double* out);
/* The inverse: overwrite one of those state matrices (allocates the ADMM state on first use). */
int nmfx_set_matrix(nmfx_handle_t h, const char* name, const double* in);

/* ---- iteration state ---------------------------------------------------- */
/* stop_rule: 0 running, 1 / 2 = which branch of convergence_check fired
 * (nmf/utils.py:8-11); stop_i: the reference's loop index `i` at which it
 * fired; n_obj: number of objective values recorded so far (obj[0] is the
 * objective of the initial factors, nmf/mur.py:115).                           */
int nmfx_get_state(nmfx_handle_t h, int* stop_rule, int64_t* stop_i, int64_t* n_obj);
int nmfx_get_objectives(nmfx_handle_t h, int64_t first, int64_t count, double* out);
/* NNLS diagnostics of ANLS since the last nmfx_set_factors: passive variables dropped because their pivot vanished
 * (a dead or collinear component at lambda = 0; their x stays 0 like in scipy's nnls / nmf/fcnnls.py) and solves that
 * ran into the iteration cap (8 k + 64 exchanges; the reference's FCNNLS prints 'Not converged.' in that case).      */
int nmfx_get_diagnostics(nmfx_handle_t h, int64_t* nnls_evicted, int64_t* nnls_capped);
/* How often the default NNLS path (f64 inverse of the Gram matrix + complement of the passive set) handed work
 * to the elimination kernels since the last nmfx_set_factors: single problems (complement larger than its workspace), and whole
 * half-steps (Gram matrix singular or too ill-conditioned for an explicit inverse).                                         */
int nmfx_get_nnls_fallbacks(nmfx_handle_t h, int64_t* problems, int64_t* half_steps);
/* AO-ADMM, fused inner rounds (nmf/ao_admm.py:58-66 run speculatively, DESIGN.md 4b "hinted speculation"): how many
 * sub-problems since the last nmfx_set_factors had a first leg that [0] stood as it was, [1] was cut back to an earlier round, [2] had to be
 * continued to admm_iter, [3] was continued and then cut back.  Cost diagnostics only: the rounds that count and their
 * arithmetic are the reference's on every path.                                                                          */
int nmfx_get_inner_paths(nmfx_handle_t h, int64_t out[4]);

/* ---- MUR (replaces the loop body nmf/mur.py:119-131) -------------------- */
/* Queue `count` outer iterations starting at iteration `first` (= number of
 * iterations already run on this handle).  Iteration j computes, like
 * mur.py:122-127: W <- w_update (mur.py:20-33), H <- h_update with the new W
 * (mur.py:36-49), objective of the result (utils.py:18-33), and the
 * convergence check for `i = j` when j > min_iter (mur.py:131).  Asynchronous;
 * read results with nmfx_get_state / nmfx_get_objectives (they synchronise).  */
int nmfx_mur_run(nmfx_handle_t h, int distance, double lambda_w, double lambda_h,
                 int64_t min_iter, double tol1, double tol2,
                 int64_t first, int64_t count);
/* Complete the objective/convergence bookkeeping of the last queued iteration
 * (the engine evaluates the objective of iteration j inside the first kernel
 * of iteration j+1; this runs that evaluation alone).                         */
int nmfx_mur_finish(nmfx_handle_t h, int distance, int64_t min_iter, double tol1,
                    double tol2, int64_t iters_done);

/* Row-sharded form: phase A = everything up to the rank-local partial sums
 * [W^T V | W^T W | objective], phase B = H update from the (all-reduced) sums.
 * Between the two the caller sum-all-reduces the exchange buffers over ranks
 * (RCCL via torch.distributed, see nmf_amd/dist.py).                          */
int nmfx_mur_phase_a(nmfx_handle_t h, int distance, double lambda_w, int64_t j);
int nmfx_mur_phase_b(nmfx_handle_t h, int distance, double lambda_h,
                     int64_t min_iter, double tol1, double tol2, int64_t j);
/* Phase A in pieces, so that the exchange can overlap with the H-side product (the all-reduce of one column chunk of V runs
 * while the next chunk is computed).  Euclidean loss on the split-bf16 path only: nmfx_mur_chunk_info returns
 * unit = 128 (0: not available for this handle / distance), the padded n and the padded k.  The f32 exchange buffer is
 * [column][factor]: after nmfx_mur_phase_a_cols(h, d, c0, c1) the range xf32[c0 * k_padded, c1 * k_padded) holds this
 * rank's part of (W^T V)[:, c0:c1] and may be reduced; the call whose c1 is the padded n also packs what lies behind the
 * k n part (W^T W, the objective partial), so its range extends to the end of the buffer.  c0, c1: multiples of `unit`,
 * at least 512 columns per call, ascending and without gaps:
 *     nmfx_mur_phase_a_head(j);  for each chunk: nmfx_mur_phase_a_cols(c0, c1), all-reduce of the range (asynchronously);
 *     wait for the reductions;  nmfx_mur_phase_b(j).
 * Results differ from nmfx_mur_phase_a by the summation order of the product's splits only.                            */
int nmfx_mur_chunk_info(nmfx_handle_t h, int distance, int64_t* unit, int64_t* n_padded, int64_t* k_padded);
int nmfx_mur_phase_a_head(nmfx_handle_t h, int distance, double lambda_w, int64_t j);
int nmfx_mur_phase_a_cols(nmfx_handle_t h, int distance, int64_t c0, int64_t c1);
int nmfx_mur_finish_a(nmfx_handle_t h, int distance, int64_t j);
int nmfx_mur_finish_b(nmfx_handle_t h, int64_t min_iter, double tol1, double tol2, int64_t j);
/* Phase B in two parts, for an exchange by reduce-scatter + all-gather instead of one all-reduce (the sum over the
 * row shards is the one nmf/mur.py:45 `w.T @ x` forms): with `world` ranks, rank r owns the columns [r * cols, (r + 1) * cols) of H,
 *     nmfx_mur_phase_a(j);
 *     reduce-scatter (sum) of xf32[0, world * elems) in place: rank r receives the range [r * elems, (r + 1) * elems);
 *         sum-all-reduce of the rest of the f32 buffer, xf32[world * elems, n_f32)   (W^T W, the objective digits);
 *     nmfx_mur_phase_b_slice(j, r * cols, (r + 1) * cols):   the update of mur.py:45 for those columns, the objective and the stop
 *         rule of iteration j (identical on every rank); the new columns are also left in the rank's range of xf32;
 *     all-gather of the ranges in place;
 *     nmfx_mur_phase_b_rest(r * cols, (r + 1) * cols):       the other ranks' columns into H (and whatever follows the update).
 * nmfx_mur_slice_info returns cols (a multiple of 64) and elems = cols * k_padded, or 0 / 0 where this form is not available:
 * Euclidean loss on the split-bf16 path only (nmfx_set_exchange_rank in force), padded n a multiple of 64 * world.  Every rank ends
 * with bit-identical H: each column is computed by ONE rank and copied.                                                           */
int nmfx_mur_slice_info(nmfx_handle_t h, int distance, int world, int64_t* cols, int64_t* elems);
int nmfx_mur_phase_b_slice(nmfx_handle_t h, int distance, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j,
                           int64_t c0, int64_t c1);
int nmfx_mur_phase_b_rest(nmfx_handle_t h, int distance, int64_t c0, int64_t c1);

/* Exchange buffers: f32 part = [W^T V (kp x n_pad) | W^T W (kp x kp)] (+ KL:
 * column sums of W), f64 part = [objective partial, 4 inner-loop norm sums (sharded AO-ADMM, round by round), 3 spare,
 * 64 x 4 norm sums of the speculative rounds (nmfx_aoadmm_phase_w_fused)].  Sizes in
 * elements.  The caller may supply its own device allocations (e.g. torch
 * tensors, so that torch.distributed can all-reduce them in place).           */
int nmfx_exchange_sizes(nmfx_handle_t h, int64_t* n_f32, int64_t* n_f64);
/* Stream-capture support for the caller-driven loop.  The phase calls only queue launches on the
 * handle's stream, so a caller may capture  phase_a(j=0) . all-reduce . phase_b(j=0) .
 * phase_a(j=1) . all-reduce . phase_b(j=1) . nmfx_shift_iteration_base(+2)  into ONE hipGraph
 * and replay it: the device adds the base to the index each launch carries (objective slot,
 * `i > min_iter` test of nmf/mur.py:131).  Keep the base even (W ping-pong parity), reserve
 * the objective history first (it must not move during replays) and shift the base back to 0
 * before nmfx_mur_finish_* / the run entry points, which take absolute indices. */
int nmfx_reserve_objectives(nmfx_handle_t h, int64_t count);
int nmfx_shift_iteration_base(nmfx_handle_t h, int64_t delta);
/* One collective per outer iteration instead of two (MUR, Euclidean loss, k padded to 64 / 128): with rank / world set, phase A
 * also writes this rank's f64 objective partial into the tail of the f32 exchange buffer -- its four 16-bit digits as exact small
 * floats in the rank's own slot, zeros in the other ranks' slots -- so that the SUM all-reduce of the f32 buffer alone delivers every
 * rank's partial bit for bit; phase B adds them in rank order.  nmfx_exchange_sizes already includes the tail (64 ranks).
 * world = 0 switches back to the separate all-reduce of the f64 buffer.  NMFX_E_STATE when the split-bf16 epilogues are not in use. */
int nmfx_set_exchange_rank(nmfx_handle_t h, int rank, int world);
/* n_f32 / n_f64: the sizes (elements) of the caller's allocations, checked against nmfx_exchange_sizes -- which grew in round 2
 * (f64: 8 -> 8 + 4 * 64 doubles; f32: a tail of 256 floats): a caller that still allocates the old sizes gets NMFX_E_ARG instead of
 * out-of-bounds device writes.  ABI change of version 300 (the two size arguments are new).                                */
int nmfx_set_exchange_buffers(nmfx_handle_t h, void* dev_f32, int64_t n_f32, void* dev_f64, int64_t n_f64);
int nmfx_get_exchange_buffers(nmfx_handle_t h, void** dev_f32, void** dev_f64);

/* ---- the stop rule's referee (r3) ---------------------------------------------------------------------------------------------
 * The objective every iteration records is evaluated in float32 products with float64 sums; near the stop of a long run its
 * iteration-to-iteration jitter (~3e-9 of the objective at 16384 x 8192) is what `new >= old - tol2` (nmf/utils.py:10) sees once
 * tol2 is below ~1e-6 of the objective.  nmfx_objective_f64 evaluates 1/2 ||V - W H||^2 of the CURRENT pair with the product and
 * the sum in float64 (f64 MFMA; ~0.4 ms at 16384 x 8192, k = 64), as the reference's arithmetic would for this iterate.
 * nmfx_set_stop_guard(h, g): the device's rule 2 becomes `new >= old - tol2 - g` -- with g a few times the jitter it fires EARLY, as
 * a candidate; nmfx_resume(h) clears the stop flag so that the caller can walk on one iteration at a time with the f64 objective
 * deciding (nmf_amd._driver.drive, `verify_stop`).  g = 0 (the default) is the plain rule.                                        */
int nmfx_objective_f64(nmfx_handle_t h, double* out);
int nmfx_set_stop_guard(nmfx_handle_t h, double guard);
int nmfx_resume(nmfx_handle_t h);

/* ---- pair mode: two MUR-Euclidean factorizations of the SAME V in one pass over it ------------------------------
 * The reference author's parameter grids (nmf/nmf_old.py:52-66, nmf/nmf.py:38-45) call the solver once per (lambda_w, lambda_h,
 * start); on the GPU two such problems with k <= 64 share the V stream: a handle created with k = 128 holds problem 0 in the
 * factor columns [0, 64) and problem 1 in [64, 128) -- set with ONE nmfx_set_factors call on the stacked, zero-padded m x 128 /
 * 128 x n matrices.  The V-sized products are the k = 128 products; objective, Gram matrices, lambda, stop rule and objective
 * history are per problem (nmf/mur.py:119-136 twice).  A problem whose stop rule fires keeps the iterate the reference returns
 * while the other one continues; the handle's ordinary stop flag is set when both have stopped.  lambda_w / lambda_h: two values
 * each.  Split-bf16 path only (NMFX_E_STATE otherwise); single GPU.                                                            */
int nmfx_mur_pair_run(nmfx_handle_t h, const double* lambda_w, const double* lambda_h, int64_t min_iter, double tol1, double tol2,
                      int64_t first, int64_t count);
int nmfx_mur_pair_finish(nmfx_handle_t h, int64_t min_iter, double tol1, double tol2, int64_t iters_done);
int nmfx_pair_get_state(nmfx_handle_t h, int p, int* stop_rule, int64_t* stop_i, int64_t* n_obj);
int nmfx_pair_get_objectives(nmfx_handle_t h, int p, int64_t first, int64_t count, double* out);
int nmfx_pair_get_factors(nmfx_handle_t h, int p, int k_p, double* w, double* hmat);      /* w: m x k_p, hmat: k_p x n */

/* ---- the exchange step behind the C ABI: RCCL over xGMI, one process per GPU -----------------------------------------------
 * north_star: "shard rows of V and W across the 8 GPUs of one node with an RCCL all-reduce over xGMI of the k x k Gram W^T W and
 * the k x n product W^T V each outer iteration" (nmf/mur.py:45 needs w.T @ x and w.T @ w over ALL rows).  RCCL is bound at run
 * time (dlopen librccl.so.1; NMFX_RCCL_LIB overrides the name): NMFX_E_RCCL when it is missing or a call fails.
 *   rank 0:     nmfx_comm_unique_id(id)            -> 128 bytes, handed to the other ranks by the launcher's own means
 *   every rank: nmfx_comm_init_rank(h, id, rank, world)        (h = the handle of this rank's row shard; collective call)
 *               nmfx_comm_negotiate(h)             -> all ranks agree on what fixes the sequence of collectives: the objective
 *                                                     partial inside the f32 buffer (one all-reduce per MUR-eu iteration), the
 *                                                     chunk unit, the arithmetic mode (NMFX_E_STATE if the modes differ)
 *               nmfx_mur_run_sharded(...)          -> `count` outer iterations, each  phase A . all-reduce . phase B  queued on
 *                                                     the handle's stream in ONE call (nmf/mur.py:119-131 for a row shard)
 *               nmfx_mur_finish_sharded(...)       -> objective of the last pair + final stop test (as nmfx_mur_finish)
 * nmfx_comm_all_reduce: in-place SUM over [first, first + count) of the f32 (which = 0) or f64 (which = 1) exchange buffer on the
 * handle's stream -- the exchange of the other solvers' phase entry points (nmfx_aoadmm_phase_*, nmfx_admm_phase_*,
 * nmfx_anls_phase_*).  nmfx_comm_all_min: MIN over the ranks of up to 64 host integers (blocking).
 * nmfx_comm_set_graph(h, 1): nmfx_mur_run_sharded replays pairs of iterations (collectives included) as ONE hipGraph after the
 * first eager pair; a capture that fails leaves the eager loop in charge.  NMFX_DIST_CHUNKS=n: phase A in n column chunks, each
 * chunk's all-reduce on a side stream behind the next chunk's product.                                                       */
int nmfx_comm_unique_id(void* id128);
int nmfx_comm_init_rank(nmfx_handle_t h, const void* id128, int rank, int world);
int nmfx_comm_destroy(nmfx_handle_t h);
int nmfx_comm_info(nmfx_handle_t h, int* rank, int* world, int* merged, int* rccl_version);
int nmfx_comm_negotiate(nmfx_handle_t h);
int nmfx_comm_all_reduce(nmfx_handle_t h, int which, int64_t first, int64_t count);
int nmfx_comm_all_min(nmfx_handle_t h, int64_t* vals, int n);
int nmfx_comm_barrier(nmfx_handle_t h);     /* the handle's queued work is done and every rank has arrived (one-word all-reduce + stream sync) */
int nmfx_comm_set_graph(nmfx_handle_t h, int enable);
/* The exchange inside nmfx_mur_run_sharded: 0 = one sum-all-reduce of the f32 buffer per iteration (default); 1 = reduce-scatter of
 * its W^T V part (+ all-reduce of the k x k Gram matrix and the objective digits, one RCCL group) . nmfx_mur_phase_b_slice .
 * all-gather . nmfx_mur_phase_b_rest -- see nmfx_mur_slice_info.  Iterations the sliced form is not available for take the
 * all-reduce.  The same value on every rank.                                                                                  */
int nmfx_comm_set_exchange(nmfx_handle_t h, int mode);
int nmfx_comm_get_exchange(nmfx_handle_t h, int* mode);
int nmfx_comm_graph_replays(nmfx_handle_t h, int64_t* replays);
int nmfx_mur_run_sharded(nmfx_handle_t h, int distance, double lambda_w, double lambda_h, int64_t min_iter,
                         double tol1, double tol2, int64_t first, int64_t count);
int nmfx_mur_finish_sharded(nmfx_handle_t h, int distance, int64_t min_iter, double tol1, double tol2, int64_t iters_done);

/* ---- initialisation: leading singular triplets of the uploaded V ------------------------
 * Replaces `numpy.linalg.svd(x, full_matrices=False)` in nmf/utils.py:50 for what NNDSVD uses of
 * it (the first `rank` triplets, utils.py:51-82; the construction is invariant under the sign
 * choice of a triplet).  f64 arithmetic on the device: block subspace iteration with
 * Rayleigh-Ritz on `block` columns (0 = max(k + 16, 1.5 k)), stopped when
 * ||V^T u_i - s_i v_i|| <= tol * s_1 for i < k (tol <= 0: 1e-11) or after max_sweeps (<= 0: 4000).
 * u: m x k, s: k, vt: k x n (row-major, host).  sweeps / resid (may be NULL) report the sweeps
 * run and the largest relative residual reached, so the caller can tell a cap from convergence. */
int nmfx_topk_svd(nmfx_handle_t h, int k, int block, double tol, int max_sweeps, uint64_t seed,
                  double* u, double* s, double* vt, int* sweeps, double* resid);

/* ---- AO-ADMM (replaces nmf/ao_admm.py:259-301) -------------------------- */
/* One call queues `count` outer iterations: H sub-problem then W sub-problem
 * (admm_ls_update, ao_admm.py:46-68: Gram, rho = trace/k, Cholesky, up to
 * admm_iter rounds of solve / prox / dual update with the `terminate` test of
 * ao_admm.py:33-43 evaluated on the device), objective, convergence check.
 * inner counts per outer iteration are readable with nmfx_get_inner_counts.   */
int nmfx_aoadmm_run(nmfx_handle_t h, int distance, int prox_w, double lambda_w,
                    int prox_h, double lambda_h, int admm_iter,
                    int64_t min_iter, double tol1, double tol2,
                    int64_t first, int64_t count);
/* Row-sharded form (Euclidean loss).  Per outer iteration j the caller runs
 *   phase_h_products . all-reduce(f32, f64) . phase_h_solve . phase_w_products .
 *   { phase_w_round(r) . all-reduce(f64) } for r = 0 .. admm_iter-1 . phase_w_close
 * The H sub-problem (ao_admm.py:263) needs sum_p W_p^T V_p and sum_p W_p^T W_p, after which it
 * is replicated work; the W sub-problem (ao_admm.py:265) is rank-local except that `terminate`
 * (ao_admm.py:33-43) takes norms over ALL rows of W: each round leaves this rank's four sums of
 * squares in the f64 exchange buffer [1..4].  After the last iteration: nmfx_objective_partial .
 * all-reduce(f64) . nmfx_mur_finish_b.                                                      */
int nmfx_aoadmm_phase_h_products(nmfx_handle_t h, int64_t j);
int nmfx_aoadmm_phase_h_solve(nmfx_handle_t h, int prox_h, double lambda_h, int admm_iter,
                              int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_aoadmm_phase_w_products(nmfx_handle_t h, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_aoadmm_phase_w_round(nmfx_handle_t h, int prox_w, double lambda_w, int round);
int nmfx_aoadmm_phase_w_close(nmfx_handle_t h, int admm_iter, int64_t j);
/* The W sub-problem with ONE exchange: nmfx_aoadmm_phase_w_fused runs all admm_iter (<= 64) rounds speculatively on
 * this rank's rows and leaves its four norm sums of every round in the f64 exchange buffer [8 + 4 round + c];
 * after the caller's all-reduce of those 4 admm_iter numbers nmfx_aoadmm_phase_w_repair derives the round at which
 * `terminate` (ao_admm.py:33-43) fires -- the same on every rank -- reruns that many rounds from the saved start if it
 * is before the last one, records the inner count and leaves the objective partials of the new pair.  Replaces
 * { phase_w_round . all-reduce } x admm_iter . phase_w_close: 3 collectives per outer iteration instead of 2 + admm_iter. */
int nmfx_aoadmm_phase_w_fused(nmfx_handle_t h, int prox_w, double lambda_w, int admm_iter);
int nmfx_aoadmm_phase_w_repair(nmfx_handle_t h, int prox_w, double lambda_w, int admm_iter, int64_t j);

/* Row-sharded AO-ADMM with the KL loss (nmf/ao_admm.py:71-101, 277-283: admm_kl_update for H, then for W on the
 * transposed data).  Its V-sized products sit INSIDE the inner rounds, so the exchange is per round:
 *   for r in range(admm_iter):  nmfx_aoadmm_kl_phase_h_products(j, r); all-reduce(f32 buffer [, f64[0..7] when r == 0]);
 *                               nmfx_aoadmm_kl_phase_h_round(prox_h, lambda_h, r, min_iter, tol1, tol2, j)
 *   nmfx_aoadmm_kl_phase_h_close(admm_iter, min_iter, tol1, tol2, j)
 *   for r in range(admm_iter):  nmfx_aoadmm_kl_phase_w_round(prox_w, lambda_w, r); all-reduce(f64[1..4])
 *   nmfx_aoadmm_kl_phase_w_close(admm_iter, j)
 * The rounds behind the inner stop (`ADMM break`, ao_admm.py:96-98) are no-ops on every rank. */
int nmfx_aoadmm_kl_phase_h_products(nmfx_handle_t h, int64_t j, int round);
int nmfx_aoadmm_kl_phase_h_round(nmfx_handle_t h, int prox_h, double lambda_h, int round, int64_t min_iter, double tol1,
                                 double tol2, int64_t j);
int nmfx_aoadmm_kl_phase_h_close(nmfx_handle_t h, int admm_iter, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_aoadmm_kl_phase_w_round(nmfx_handle_t h, int prox_w, double lambda_w, int round);
int nmfx_aoadmm_kl_phase_w_close(nmfx_handle_t h, int admm_iter, int64_t j);
/* f64 exchange buffer [0] = this rank's objective partial of the current factor pair.        */
int nmfx_objective_partial(nmfx_handle_t h);
/* Record the objective / convergence test of the last queued iteration.       */
int nmfx_aoadmm_finish(nmfx_handle_t h, int64_t min_iter, double tol1, double tol2, int64_t iters_done);
/* Per outer iteration two int32: (H sub-problem, W sub-problem); low 16 bits =
 * inner rounds run, bit 16 = the inner stop test fired ("ADMM break after").  */
int nmfx_get_inner_counts(nmfx_handle_t h, int64_t first, int64_t count, int32_t* out_pairs);

/* ---- ADMM (replaces nmf/admm.py:292-334) -------------------------------- */
/* prox 'l2n' (nmf/admm.py:141-156) solves (1/rho)(lambda T^T T + rho I) X = aux - dual
 * with the k x k second-difference operator T; since rho and lambda are fixed
 * for a run the caller supplies the k x k inverse P (float64, row-major) once:
 * which = 0 for the W regulariser, 1 for H.  The finish call of AO-ADMM
 * (nmfx_aoadmm_finish) is shared by ADMM.                                     */
int nmfx_set_l2n_operator(nmfx_handle_t h, int which, const double* p_inverse);
/* prox 'l1inf' / 'l1inf_transpose' (nmf/admm.py:158-183, :185-210, upper_bound = 1) as one launch on the current
 * ADMM state: side 0: w = prox(w_aux^T, dual_w^T)^T (admm.py:320), side 1: h = prox(h_aux, dual_h) (admm.py:319);
 * update_dual != 0 also does dual += x - x_aux (admm.py:321-322).  nmfx_admm_run calls the same kernels; this entry
 * exists so that the operator can be checked at function level (state through nmfx_set_matrix / nmfx_get_factors).
 * 'l1inf' sorts vectors of n (side 1) or m (side 0) entries in LDS: at most 32768.                                   */
int nmfx_prox_apply(nmfx_handle_t h, int side, int prox, double rho, double lambda, int update_dual);
int nmfx_admm_run(nmfx_handle_t h, int distance, double rho, int prox_w, double lambda_w,
                  int prox_h, double lambda_h, int64_t min_iter, double tol1,
                  double tol2, int64_t first, int64_t count);
/* Row-sharded form.  Per outer iteration j:  phase_products . all-reduce(f32, f64) . phase_update
 * phase_products leaves this rank's [w_aux^T V | w_aux^T w_aux] (KL loss: w_aux^T (v_aux + dual_v)) and the objective
 * partial of the current pair in the exchange buffers (j = 0: also the start state of admm.py:27-28, 289);
 * phase_update records obj[j] / evaluates the stop rule, solves h_aux from the all-reduced sums (replicated work,
 * admm.py:295), solves this rank's rows of w_aux (admm.py:297), applies both prox operators and dual updates
 * (admm.py:319-322).  ADMM has no inner loop: this is the only exchange.  After the last iteration:
 * nmfx_objective_partial . all-reduce(f64) . nmfx_mur_finish_b.  The row-coupled prox 'l1inf' on the W side and
 * 'l1inf_transpose' on the W side (column 1 of dual_w^T = global row 1) are not available sharded.                   */
int nmfx_admm_phase_products(nmfx_handle_t h, int distance, double rho, int prox_w, int prox_h, int64_t j);
int nmfx_admm_phase_update(nmfx_handle_t h, int distance, double rho, int prox_w, double lambda_w,
                           int prox_h, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j);

/* ---- ANLS (replaces nmf/anls.py:111-122) -------------------------------- */
/* distance_type of ANLS (nmf/anls.py:108,118): NMFX_EU (default) or NMFX_KL.  It only selects the objective that is
 * recorded and tested by the stop rule; the iterates are the least-squares ones either way, as in the reference.   */
int nmfx_anls_set_distance(nmfx_handle_t h, int distance);
int nmfx_anls_run(nmfx_handle_t h, double lambda_w, double lambda_h,
                  int64_t min_iter, double tol1, double tol2,
                  int64_t first, int64_t count);
/* Row-sharded form.  Per outer iteration j:
 *   phase_objective . all-reduce(f64) . phase_w . all-reduce(f32) . phase_h
 * phase_w records obj[j] / evaluates the stop rule, solves the rank's rows of W (anls.py:18-31)
 * and leaves [W^T V | W^T W] partials in the exchange buffer; phase_h solves all columns of H
 * (replicated, anls.py:34-47).  After the last iteration: nmfx_objective_partial .
 * all-reduce(f64) . nmfx_mur_finish_b.                                                      */
int nmfx_anls_phase_objective(nmfx_handle_t h, int64_t j);
int nmfx_anls_phase_w(nmfx_handle_t h, double lambda_w, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_anls_phase_h(nmfx_handle_t h, double lambda_h, int64_t j);

/* ---- measurement -------------------------------------------------------- */
/* Accumulated device time (HIP events on the handle's stream) and launch count
 * of a named kernel since the last reset; names: "wphase" "hphase" ...        */
int nmfx_profile_enable(nmfx_handle_t h, int on);
int nmfx_profile_get(nmfx_handle_t h, const char* name, double* total_ms, int64_t* launches);
int nmfx_profile_reset(nmfx_handle_t h);
/* Mean duration of ONE product kernel of MUR -- which = "wphase" (V H^T + objective, nmf/mur.py:29 + utils.py:29)
 * or "hphase" (W^T V, nmf/mur.py:45), distance NMFX_EU / NMFX_KL -- over `reps` back-to-back launches on the
 * handle's stream between one pair of HIP events (two untimed launches first).  The launches recompute the
 * products of the current factors; no factor is modified.  This is the live per-launch time `bench.py` divides
 * the algorithmic bytes by: a HIP event in front of EVERY launch (nmfx_profile_enable) puts a command-processor
 * barrier there, which the loop of a real factorisation does not have.                                        */
int nmfx_profile_repeat(nmfx_handle_t h, const char* which, int distance, int reps, double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* NMFX_H */
