// Device helpers shared by the solver-specific kernel files.
#pragma once
#include "nmfx_internal.h"

// Record obj[j] (the objective of the factors after j outer iterations) and
// evaluate the reference's stop rule for loop index i = j-1
// (nmf/mur.py:131 `if i > min_iter`, nmf/utils.py:4-15):
//   rule 1: obj < tol1,   rule 2: obj >= obj[j-1] - tol2   (tested in that order).
// NaN compares false on both, exactly like the numpy scalars of the reference.
// Every caller evaluates the same inputs, so all blocks agree; `writer` selects
// the single thread that publishes the result.
__device__ __forceinline__ int nmfx_record_objective(DevState* st, double* obj_hist, double obj,
                                                     long long j, long long min_iter, double tol1,
                                                     double tol2, bool writer)
{
    int rule = 0;
    j += st->j_base;
    if (j >= 1 && (j - 1) > min_iter) {
        const double prev = obj_hist[j - 1];
        if (obj < tol1) rule = 1;
        else if (obj >= prev - tol2 - st->stop_guard) rule = 2;      // (stop_guard = 0 unless nmfx_set_stop_guard)
    }
    if (writer) {
        obj_hist[j] = obj;
        st->n_obj = j + 1;
        if (rule) { st->flag = rule; st->stop_i = j - 1; }
    }
    return rule;
}

// Pair mode: the same test and bookkeeping for problem p of two (objective slots 2 j + p, state in pflag / pstop_i / pn_obj).
__device__ __forceinline__ int nmfx_record_objective_pair(DevState* st, double* obj_hist, double obj, int p, long long j,
                                                          long long min_iter, double tol1, double tol2, bool writer)
{
    int rule = 0;
    if (j >= 1 && (j - 1) > min_iter) {
        const double prev = obj_hist[2 * (j - 1) + p];
        if (obj < tol1) rule = 1;
        else if (obj >= prev - tol2) rule = 2;
    }
    if (writer) {
        obj_hist[2 * j + p] = obj;
        st->pn_obj[p] = j + 1;
        if (rule) { st->pflag[p] = rule; st->pstop_i[p] = j - 1; }
    }
    return rule;
}

// ---- inner stop test -------------------------------------------------------
// `terminate` (ao_admm.py:33-43) on the four sums of squares: ||X - aux|| / ||X|| < 1e-2 and ||X - X_prev|| / ||U|| < 1e-2,
// written without the square roots and divisions (~1500 cycles of f64 per test on the serial path of every round):
// sqrt(a) / sqrt(b) < 1e-2  <=>  a < 1e-4 b for a >= 0, b > 0; b = 0 gives inf or nan on the left (false) and a < 0 on the
// right (false); a nan makes both false.  Only sums within a rounding error of the threshold could be told apart.
__device__ __forceinline__ bool inner_test(double a, double b, double c, double d) {
    return (a < 1e-4 * b) && (c < 1e-4 * d);
}
// Sum the four norm partials of the previous round; identical in every block.
__device__ __forceinline__ bool inner_round_fired(const double* __restrict__ part, int nblk, double* sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;     // wave w sums component w
    double s = 0.0;
    if (wave < 4)
        for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * 4 + wave];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0 && wave < 4) sh[wave] = s;
    __syncthreads();
    const bool hit = inner_test(sh[0], sh[1], sh[2], sh[3]);
    __syncthreads();
    return hit;
}

