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
