// ADMM (single-level) outer iteration, Euclidean loss.
//   reference: nmf/admm.py:292-334; aux_update :216-230 (np.linalg.solve of the
//   shifted Gram system), prox :117-156, initialize :17-35.
//
//   h_aux = (w_aux^T w_aux + rho I)^-1 (w_aux^T V + rho (h + dual_h))
//   w_aux = ((h_aux h_aux^T + rho I)^-1 (h_aux V^T + rho (w^T + dual_w^T)))^T
//   h = prox(h_aux, dual_h);  w = prox(w_aux, dual_w);  duals += x - x_aux
//
// Each half is exactly one round of the AO-ADMM kernels (kernels_aoadmm.hip) with a
// fixed rho and the aux matrix kept; prox 'l2n' (a k x k banded solve in the
// reference, admm.py:141-156) is a second product with the fixed k x k inverse
// P = ((lambda T^T T + rho I) / rho)^-1 supplied by the host (nmfx_set_l2n_operator).
#include "nmfx_internal.h"
#include "kernels_small.h"
#include <vector>

template <typename T>
static int lazy_alloc(nmfx_engine* E, T** p, int64_t count) {
    if (*p) return NMFX_OK;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T)));
    NMFX_HIP(hipMemsetAsync(*p, 0, (size_t)count * sizeof(T), E->stream));
    return NMFX_OK;
}

static int admm_alloc(nmfx_engine* E) {
    int rc;
    if ((rc = nmfx_aoadmm_alloc(E))) return rc;
    if ((rc = lazy_alloc(E, &E->auxH, (int64_t)E->kp * E->np))) return rc;
    if ((rc = lazy_alloc(E, &E->Asum, E->mp * E->kp))) return rc;
    return NMFX_OK;
}

int nmfx_admm_state_alloc(nmfx_engine* E) { return admm_alloc(E); }

static bool admm_bf16(const nmfx_engine* E) { return E->precision == 1 && nmfx_bf16_supported(E); }

// objective partials of (W, H) (admm.py:324); split-bf16: one pass of the product kernel with
// Y = H, Z = W (its A output is not used)
static int admm_objective(nmfx_engine* E) {
    int rc;
    if (!admm_bf16(E)) return nmfx_launch_wphase(E, E->W[0], false, true);
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if ((rc = nmfx_bf16_images_w(E, E->W[0], 1))) return rc;
    if ((rc = nmfx_bf16_images_h(E, false))) return rc;
    return nmfx_bf16_objective(E, 1, "objective");
}

// first half of an iteration: this rank's [w_aux^T V | w_aux^T w_aux | objective partial] into the exchange buffers
// (row-sharded runs all-reduce them here)
static int admm_eu_products(nmfx_engine* E) {
    int rc;
    const bool bf = admm_bf16(E);
    if (bf) {       // w_aux^T V and w_aux^T w_aux on the split-bf16 kernel (kp = 64: the Gram is its by-product)
        const int64_t nobj = E->obj_count;             // the objective pass that ended the previous iteration
        if ((rc = nmfx_bf16_images_w(E, E->auxW, 0))) return rc;
        if ((rc = nmfx_bf16_vtw(E, false, "hphase"))) return rc;
        if (E->kp == 64) rc = nmfx_bf16_pack_t(E, E->G_part, nmfx_bf16_g_slabs(E), nobj);
        else {
            if ((rc = nmfx_launch_gram_tn(E, E->auxW, E->mp, E->G_part, E->gsplit))) return rc;
            rc = nmfx_bf16_pack_t(E, E->G_part, E->gsplit, nobj);
        }
        if (rc) return rc;
    } else {
    const bool fuse_g = nmfx_hphase_can_fuse_gram(E);
    if (!fuse_g && (rc = nmfx_launch_gram_tn(E, E->auxW, E->mp, E->G_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_hphase(E, E->auxW, fuse_g))) return rc;
    if ((rc = nmfx_launch_pack(E))) return rc;
    }
    return NMFX_OK;
}

static int admm_eu_update(nmfx_engine* E, double rho, int prox_w, double lam_w, int prox_h, double lam_h,
                          int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    const int64_t kk = (int64_t)E->kp * E->kp;
    const bool bf = admm_bf16(E);
    // ---- h_aux and the H half ----
    if ((rc = nmfx_launch_prepare(E, E->xf32 + (int64_t)E->kp * E->np, 1, j, min_iter, tol1, tol2, rho))) return rc;
    { ProfScope ps(E, "inner_h");
      if (prox_h == NMFX_PROX_L2N) {
          if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 1, prox_h, (float)lam_h, 0))) return rc;
          if ((rc = nmfx_inner_cols(E, E->Ph, E->auxH, 2, prox_h, (float)lam_h, 0))) return rc;
      } else if (prox_h == NMFX_PROX_L1INF || prox_h == NMFX_PROX_L1INF_T) {   // couples whole rows / columns: its own launch
          if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 1, prox_h, (float)lam_h, 0))) return rc;
          if ((rc = nmfx_launch_prox_l1inf(E, true, prox_h == NMFX_PROX_L1INF_T, rho, lam_h, 1.0, true))) return rc;
      } else if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 0, prox_h, (float)lam_h, 0))) return rc; }
    // ---- w_aux (from the NEW h_aux) and the W half ----
    if (bf) {
        const bool byprod = E->kp == 64;               // h_aux h_aux^T as a by-product
        if ((rc = nmfx_bf16_images_h(E, false, E->auxH))) return rc;
        if ((rc = nmfx_bf16_vht(E, false, 0, "wphase_noobj"))) return rc;
        if (!byprod && (rc = nmfx_launch_gram_nt(E, E->auxH, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
        { ProfScope ps(E, "sums");
          if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, byprod ? nmfx_bf16_hht_slabs(E) : E->gsplit, kk, E->HHt))) return rc;
          if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->bf_wsplit, E->mp * E->kp, E->Asum))) return rc; }
    } else {
    if ((rc = nmfx_launch_gram_nt(E, E->auxH, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_wphase(E, W, true, false, false, E->auxH))) return rc;
    { ProfScope ps(E, "sums");
      if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
      if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->wsplit, E->mp * E->kp, E->Asum))) return rc; }
    }
    if ((rc = nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, rho))) return rc;
    { ProfScope ps(E, "inner_w");
      if (prox_w == NMFX_PROX_L2N) {
          if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 1, prox_w, (float)lam_w, 0))) return rc;
          if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Pw, E->auxW, 2, prox_w, (float)lam_w, 0))) return rc;
      } else if (prox_w == NMFX_PROX_L1INF || prox_w == NMFX_PROX_L1INF_T) {
          if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 1, prox_w, (float)lam_w, 0))) return rc;
          if ((rc = nmfx_launch_prox_l1inf(E, false, prox_w == NMFX_PROX_L1INF_T, rho, lam_w, 1.0, true))) return rc;
      } else if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 0, prox_w, (float)lam_w, 0))) return rc; }
    // ---- objective of (w, h) (admm.py:324) ----
    return admm_objective(E);
}

static int admm_eu_iteration(nmfx_engine* E, double rho, int prox_w, double lam_w, int prox_h, double lam_h,
                             int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    if ((rc = admm_eu_products(E))) return rc;
    return admm_eu_update(E, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j);
}

// KL loss (admm.py:303-315): the Gram right-hand sides multiply S = v_aux + dual_v, and
// v_aux / dual_v are refreshed from w_aux @ h_aux at the end of the iteration.
static int admm_kl_products(nmfx_engine* E) {
    int rc;
    const bool fuse_g = nmfx_hphase_can_fuse_gram(E);
    if (!fuse_g && (rc = nmfx_launch_gram_tn(E, E->auxW, E->mp, E->G_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_hphase(E, E->auxW, fuse_g, E->S))) return rc;
    return nmfx_launch_pack(E);
}

static int admm_kl_update(nmfx_engine* E, double rho, int prox_w, double lam_w, int prox_h, double lam_h,
                          int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    const int64_t kk = (int64_t)E->kp * E->kp;
    if ((rc = nmfx_launch_prepare(E, E->xf32 + (int64_t)E->kp * E->np, 1, j, min_iter, tol1, tol2, rho))) return rc;
    if (prox_h == NMFX_PROX_L2N) {
        if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 1, prox_h, (float)lam_h, 0))) return rc;
        if ((rc = nmfx_inner_cols(E, E->Ph, E->auxH, 2, prox_h, (float)lam_h, 0))) return rc;
    } else if (prox_h == NMFX_PROX_L1INF || prox_h == NMFX_PROX_L1INF_T) {
        if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 1, prox_h, (float)lam_h, 0))) return rc;
        if ((rc = nmfx_launch_prox_l1inf(E, true, prox_h == NMFX_PROX_L1INF_T, rho, lam_h, 1.0, true))) return rc;
    } else if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 0, prox_h, (float)lam_h, 0))) return rc;
    if ((rc = nmfx_launch_gram_nt(E, E->auxH, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_wphase(E, W, true, false, false, E->auxH, E->S))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->wsplit, E->mp * E->kp, E->Asum))) return rc;
    if ((rc = nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, rho))) return rc;
    if (prox_w == NMFX_PROX_L2N) {
        if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 1, prox_w, (float)lam_w, 0))) return rc;
        if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Pw, E->auxW, 2, prox_w, (float)lam_w, 0))) return rc;
    } else if (prox_w == NMFX_PROX_L1INF || prox_w == NMFX_PROX_L1INF_T) {
        if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 1, prox_w, (float)lam_w, 0))) return rc;
        if ((rc = nmfx_launch_prox_l1inf(E, false, prox_w == NMFX_PROX_L1INF_T, rho, lam_w, 1.0, true))) return rc;
    } else if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 0, prox_w, (float)lam_w, 0))) return rc;
    if ((rc = nmfx_launch_kl_vaux(E, E->auxW, E->auxH))) return rc;
    return nmfx_launch_wphase(E, W, false, true, true);
}

static int admm_kl_iteration(nmfx_engine* E, double rho, int prox_w, double lam_w, int prox_h, double lam_h,
                             int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    if ((rc = admm_kl_products(E))) return rc;
    return admm_kl_update(E, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j);
}

// The KL-loss iteration on the split-bf16 kernels (r4; one GPU, k padded to 64 or 128; NMFX_KL_BF16=0 keeps the exact-f32 launches):
// both right-hand sides from the SAME S = v_aux + dual_v (admm.py:303-305) -- w_aux^T S on the copy of S in the orientation of V^T
// (kl_S[0]), S h_aux^T on the one the auxiliaries write (kl_S[1], rows m) --, the two halves as in the exact path, then the m x n
// auxiliaries where the product w_aux h_aux stands in the accumulators (admm.py:311-314) and the transposed copy of the new S.
static bool admm_kl_bf16(const nmfx_engine* E) {
    static const bool on = !(getenv("NMFX_KL_BF16") && atoi(getenv("NMFX_KL_BF16")) == 0);
    return on && admm_bf16(E);
}

static int admm_kl_objective_bf16(nmfx_engine* E) {     // KL(V, w h) partials from the images of (w, h): one pass over the tile-major V
    int rc;
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if ((rc = nmfx_bf16_images_w(E, E->W[0], 0))) return rc;
    if ((rc = nmfx_bf16_images_h(E, false))) return rc;
    return nmfx_bf16_kl_objective(E);
}

static int admm_kl_iteration_bf16(nmfx_engine* E, double rho, int prox_w, double lam_w, int prox_h, double lam_h,
                                  int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    const int64_t kk = (int64_t)E->kp * E->kp;
    const int64_t nobj32 = E->obj_count;                               // partials of the KL objective pass that closed the iteration before
    // r5 (NMFX_KL_GATHER, default on): no transposed copy of S -- a product reads it from the buffer of the other orientation through the
    // product kernel's transposing requests.  The auxiliaries then run in the orientation of V^T (side 0: X = V^T, Y = w_aux^T images,
    // Z = h_aux^T images), where their launch can also form w_aux^T S, the first product of the NEXT iteration, from S in registers
    // (NMFX_KL_FUSE, default on; w_aux does not change in between).  NMFX_KL_GATHER=0: the r4 sequence (auxiliaries in the orientation
    // of V, S transposed at the end of every iteration).
    static const bool gather = !(getenv("NMFX_KL_GATHER") && atoi(getenv("NMFX_KL_GATHER")) == 0);
    static const bool fuse = gather && !(getenv("NMFX_KL_FUSE") && atoi(getenv("NMFX_KL_FUSE")) == 0);
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if ((rc = nmfx_bf16_kl_state(E, false))) return rc;
    // ---- h_aux = (w_aux^T w_aux + rho I)^-1 (w_aux^T S + rho (h + dual_h)) ----
    if ((rc = nmfx_bf16_images_w(E, E->auxW, 0))) return rc;           // (w_aux)^T images: Y
    const bool have_bt = fuse && E->kl_bt_ready;                       // (left by the auxiliaries launch of the iteration before)
    E->kl_bt_ready = false;
    if (!have_bt && (rc = nmfx_bf16_kl_product(E, 0, 4, nullptr, E->kl_s_side != 0))) return rc;
    if (E->kp == 64 && !gather) rc = nmfx_bf16_pack_t(E, E->G_part, nmfx_bf16_g_slabs(E), nobj32);    // (the product's Gram by-product)
    else {
        if ((rc = nmfx_launch_gram_tn(E, E->auxW, E->mp, E->G_part, E->gsplit))) return rc;
        rc = nmfx_bf16_pack_t(E, E->G_part, E->gsplit, nobj32);
    }
    if (rc) return rc;
    if ((rc = nmfx_launch_prepare(E, E->xf32 + (int64_t)E->kp * E->np, 1, j, min_iter, tol1, tol2, rho))) return rc;
    if (prox_h == NMFX_PROX_L2N) {
        if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 1, prox_h, (float)lam_h, 0))) return rc;
        if ((rc = nmfx_inner_cols(E, E->Ph, E->auxH, 2, prox_h, (float)lam_h, 0))) return rc;
    } else if (prox_h == NMFX_PROX_L1INF || prox_h == NMFX_PROX_L1INF_T) {
        if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 1, prox_h, (float)lam_h, 0))) return rc;
        if ((rc = nmfx_launch_prox_l1inf(E, true, prox_h == NMFX_PROX_L1INF_T, rho, lam_h, 1.0, true))) return rc;
    } else if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 0, prox_h, (float)lam_h, 0))) return rc;
    // ---- w_aux from the NEW h_aux ----
    if ((rc = nmfx_bf16_images_h(E, gather, E->auxH))) return rc;      // h_aux images: Y of the product (and, transposed, Z of the auxiliaries in the orientation of V^T)
    if ((rc = nmfx_bf16_kl_product(E, 1, 4, nullptr, E->kl_s_side != 1))) return rc;
    if ((rc = nmfx_launch_gram_nt(E, E->auxH, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->bf_wsplit, E->mp * E->kp, E->Asum))) return rc;
    if ((rc = nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, rho))) return rc;
    if (prox_w == NMFX_PROX_L2N) {
        if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 1, prox_w, (float)lam_w, 0))) return rc;
        if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Pw, E->auxW, 2, prox_w, (float)lam_w, 0))) return rc;
    } else if (prox_w == NMFX_PROX_L1INF || prox_w == NMFX_PROX_L1INF_T) {
        if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 1, prox_w, (float)lam_w, 0))) return rc;
        if ((rc = nmfx_launch_prox_l1inf(E, false, prox_w == NMFX_PROX_L1INF_T, rho, lam_w, 1.0, true))) return rc;
    } else if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 0, prox_w, (float)lam_w, 0))) return rc;
    // ---- v_aux, dual_v from w_aux h_aux; the new S ----
    if ((rc = nmfx_bf16_images_w(E, E->auxW, 0))) return rc;           // the new w_aux: Z (orientation of V) / Y (orientation of V^T)
    if (!gather) {
        if ((rc = nmfx_bf16_vaux(E, 1))) return rc;
        E->kl_s_side = 1;
        if ((rc = nmfx_bf16_kl_orient(E, 0, false, true))) return rc;  // (kl_S[1] stays valid: the W-side product of the next iteration reads it)
    } else {
        if (fuse) { if ((rc = nmfx_bf16_vaux_fused(E, 0, nullptr, nullptr, 0, true))) return rc; E->kl_bt_ready = true; }
        else if ((rc = nmfx_bf16_vaux(E, 0))) return rc;
        E->kl_s_side = 0;
    }
    E->wimg_ok = false; E->himg_both = false;
    return admm_kl_objective_bf16(E);                                  // KL objective of (w, h) (admm.py:324)
}

extern "C" int nmfx_set_l2n_operator(nmfx_handle_t E, int which, const double* p) {
    if (!E || !p || (which != 0 && which != 1)) { if (E) E->err = "set_l2n_operator: bad argument"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    float** dst = which == 0 ? &E->Pw : &E->Ph;
    int rc;
    if ((rc = lazy_alloc(E, dst, (int64_t)E->kp * E->kp))) return rc;
    std::vector<float> tmp((size_t)E->kp * E->kp, 0.f);
    for (int i = 0; i < E->k; ++i)
        for (int c = 0; c < E->k; ++c) tmp[(size_t)i * E->kp + c] = (float)p[(size_t)i * E->k + c];
    NMFX_HIP(hipMemcpyAsync(*dst, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

// Function-level entry (tests, and callers that drive their own ADMM loop): X = prox(X_aux, dual) on one side of the
// current ADMM state; update_dual != 0 also performs dual += X - X_aux (admm.py:321-322).
extern "C" int nmfx_prox_apply(nmfx_handle_t E, int side, int prox, double rho, double lambda, int update_dual) {
    if (!E || (side != 0 && side != 1)) { if (E) E->err = "prox_apply: side must be 0 (W) or 1 (H)"; return NMFX_E_ARG; }
    if (prox != NMFX_PROX_L1INF && prox != NMFX_PROX_L1INF_T) { E->err = "prox_apply: only the l1inf operators have a launch of their own"; return NMFX_E_ARG; }
    if (!(rho != 0.0)) { E->err = "prox_apply: rho must not be zero"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if ((rc = nmfx_small_k_only(E, "prox_apply"))) return rc;
    if ((rc = admm_alloc(E))) return rc;
    E->wsel = 0; E->w_in_place = true;
    return nmfx_launch_prox_l1inf(E, side == 1, prox == NMFX_PROX_L1INF_T, rho, lambda, 1.0, update_dual != 0);
}

// argument checks, allocations and (for the first iteration) the start state w_aux = w, h_aux = h (admm.py:27-28)
// with the objective partials of the initial pair (admm.py:289)
static int admm_begin(nmfx_engine* E, int distance, double rho, int prox_w, int prox_h, int64_t first, int64_t count, bool whole_run = false) {
    if (!E) return NMFX_E_ARG;
    E->anls_a_ready = false; E->kl_h_iter = -2;
    E->himg_both = false;
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    if (distance != NMFX_EU && distance != NMFX_KL) { E->err = "Unknown loss type."; return NMFX_E_ARG; }
    auto bad = [](int p) { return p != NMFX_PROX_NN && p != NMFX_PROX_L1N && p != NMFX_PROX_L2N && p != NMFX_PROX_L1INF && p != NMFX_PROX_L1INF_T; };
    if (bad(prox_w) || bad(prox_h)) { E->err = "Unknown prox_type."; return NMFX_E_ARG; }
    // rho = 0 is a plain Gram solve in the reference (np.linalg.solve, admm.py:230); negative values make the shifted
    // system indefinite -- the elimination then reports "not positive definite" like any singular system
    if (first < 0 || count < 0 || !(rho >= 0.0)) { E->err = "negative iteration range or rho"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if ((rc = nmfx_enter_family(E, 3))) return rc;
    if ((rc = admm_alloc(E))) return rc;
    if (distance == NMFX_KL && (rc = nmfx_kl_state_alloc(E))) return rc;
    if ((prox_w == NMFX_PROX_L2N && !E->Pw) || (prox_h == NMFX_PROX_L2N && !E->Ph)) {
        E->err = "l2n prox needs nmfx_set_l2n_operator first"; return NMFX_E_STATE; }
    if ((rc = nmfx_ensure_obj_capacity(E, first + count + 2))) return rc;
    E->wsel = 0;
    E->w_in_place = true;
    if (first == 0 && count > 0) {
        NMFX_HIP(hipMemcpyAsync(E->auxW, E->W[0], (size_t)E->mp * E->kp * 4, hipMemcpyDeviceToDevice, E->stream));
        NMFX_HIP(hipMemcpyAsync(E->auxH, E->H, (size_t)E->kp * E->np * 4, hipMemcpyDeviceToDevice, E->stream));
        if (E->kp > 128) return NMFX_OK;               // (nmfx_generic_admm_run evaluates the initial objective itself)
        if (distance == NMFX_EU) rc = admm_objective(E);
        else if (admm_kl_bf16(E) && whole_run) rc = admm_kl_objective_bf16(E);      // (nmfx_admm_run; the row-sharded phases keep the exact-f32 pass their pack counts on)
        else rc = nmfx_launch_wphase(E, E->W[0], false, true, true);
        if (rc) return rc;
    }
    if (distance == NMFX_EU && admm_bf16(E) && (rc = nmfx_bf16_prepare(E))) return rc;
    return NMFX_OK;
}

// ---- row-sharded form: phase_products -> [all-reduce f32 + f64] -> phase_update (the H half is replicated work on the
// all-reduced sums, the W half is local to the rank's rows; ADMM has no inner loop, so this is the only exchange) ----
extern "C" int nmfx_admm_phase_products(nmfx_handle_t E, int distance, double rho, int prox_w, int prox_h, int64_t j) {
    int rc = admm_begin(E, distance, rho, prox_w, prox_h, j, 1); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_admm_phase(E, 0, distance, rho, prox_w, 0.0, prox_h, 0.0, 0, 0.0, 0.0, j);      // (r4)
    return distance == NMFX_EU ? admm_eu_products(E) : admm_kl_products(E);
}

extern "C" int nmfx_admm_phase_update(nmfx_handle_t E, int distance, double rho, int prox_w, double lambda_w, int prox_h,
                                      double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j) {
    if (!E) return NMFX_E_ARG;
    if (distance != NMFX_EU && distance != NMFX_KL) { E->err = "Unknown loss type."; return NMFX_E_ARG; }
    if (!E->auxH || j < 0) { E->err = "admm_phase_update: call nmfx_admm_phase_products first"; return NMFX_E_STATE; }
    NMFX_HIP(hipSetDevice(E->device));
    if (E->kp > 128) return nmfx_generic_admm_phase(E, 1, distance, rho, prox_w, lambda_w, prox_h, lambda_h, min_iter, tol1, tol2, j);
    return distance == NMFX_EU ? admm_eu_update(E, rho, prox_w, lambda_w, prox_h, lambda_h, min_iter, tol1, tol2, j)
                               : admm_kl_update(E, rho, prox_w, lambda_w, prox_h, lambda_h, min_iter, tol1, tol2, j);
}

extern "C" int nmfx_admm_run(nmfx_handle_t E, int distance, double rho, int prox_w, double lambda_w, int prox_h,
                             double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t first,
                             int64_t count) {
    int rc = admm_begin(E, distance, rho, prox_w, prox_h, first, count, true); if (rc) return rc;
    E->kl_bt_ready = false;    // (slabs a fused auxiliaries launch left for this iteration are not trusted across calls)
    if (E->kp > 128)           // composed from the generic product kernel (kernels_generic.hip)
        return nmfx_generic_admm_run(E, distance, rho, prox_w, lambda_w, prox_h, lambda_h, min_iter, tol1, tol2, first, count);
    for (int64_t j = first; j < first + count; ++j) {
        rc = distance == NMFX_EU
            ? admm_eu_iteration(E, rho, prox_w, lambda_w, prox_h, lambda_h, min_iter, tol1, tol2, j)
            : admm_kl_bf16(E) ? admm_kl_iteration_bf16(E, rho, prox_w, lambda_w, prox_h, lambda_h, min_iter, tol1, tol2, j)
                              : admm_kl_iteration(E, rho, prox_w, lambda_w, prox_h, lambda_h, min_iter, tol1, tol2, j);
        if (rc) return rc;
    }
    return NMFX_OK;
}
