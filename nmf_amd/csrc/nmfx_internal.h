// Internal declarations shared by the translation units of libnmfx.so.
// gfx950 only: 64-lane wavefronts, f32-input MFMA (v_mfma_f32_16x16x4_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <vector>
#include "../../include/nmfx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Geometry of the padded device layout.  All big kernels are written without
// bounds checks: V is stored zero-padded to multiples of 64 in both dimensions
// and the factor rank is padded to KP in {16, 32, 64, 128}.  Zero padding is a
// fixed point of every update rule implemented here ("padding").
#define NMFX_TILE 64
#define NMFX_MAX_FUSED_ROUNDS 64     // rows of the norm table in the f64 exchange buffer (sharded fused W sub-problem)
// Tail of the f32 exchange buffer, behind [k n | k k | k]: per rank (up to 64) the four 16-bit digits of its f64 objective
// partial as exact small floats.  A rank writes its own slot and zeros in the others, so the SUM all-reduce of the buffer
// hands every rank every rank's partial bit for bit (x + 0 = x); the consumer adds them in rank order in f64.  One
// collective per outer iteration instead of two (MUR, Euclidean loss, split-bf16 epilogues).
#define NMFX_XTAIL_RANKS 64
#define NMFX_XTAIL (NMFX_XTAIL_RANKS * 4)     // floats of that tail

struct DevState {          // lives in device memory, written by kernels
    int flag;              // 0 running, 1/2 = convergence_check branch (utils.py:8-11)
    int pad0;
    long long stop_i;      // reference loop index at which the check fired
    long long n_obj;       // objectives recorded
    double obj_prev;       // obj[j-1]
    // AO-ADMM inner loop
    int inner_stop;        // set when `terminate` (ao_admm.py:33-43) fires
    int inner_count;       // inner iterations executed in the current sub-problem
    int notpd;             // Cholesky hit a non-positive pivot
    int notpd_pending;     // ... in an inversion that ran BESIDE the product whose objective may still stop the run (r4: the side job)
    double rho;            // trace(G)/k of the current sub-problem
    // added to the iteration index a launch carries: lets a captured hipGraph of two outer
    // iterations (indices 0 and 1 baked into its kernel arguments) be replayed for any pair
    long long j_base;
    // NNLS diagnostics (ANLS): passive variables dropped because their pivot vanished (a dead or collinear
    // component at lambda = 0: its x stays 0, as in Lawson-Hanson / FCNNLS), solves that hit the iteration cap
    int nnls_evicted;
    int nnls_capped;
    int nnls_fallback;     // problems the inverse + complement pass left to the elimination kernel (complement too large)
    int nnls_noinv;        // half-steps whose Gram matrix had no usable explicit inverse (all problems to the elimination kernel)
    // AO-ADMM, hinted speculation of the fused inner rounds: [H side, W side][parity of the outer iteration] = the count of
    // rounds of the side's previous sub-problem (0: none yet).  Only ever changes the cost of a sub-problem, not its result.
    int ao_hint[2][2];
    int ao_continued;      // the decide launch of the current sub-problem found no stop among the hinted rounds and went on
    int pad2;
    int ao_paths[4];       // sub-problems whose first leg stood / was cut back / was continued / was continued and cut back
    // pair mode (two MUR-eu problems stacked into the k = 128 layouts, kernels_bf16.hip): per-problem stop state; the objective
    // of problem p after iteration j sits at obj_hist[2 j + p]; `flag` is set once BOTH have stopped
    int pflag[2];
    long long pstop_i[2];
    long long pn_obj[2];
    // nmfx_set_stop_guard: rule 2 fires at `new >= old - tol2 - stop_guard` (a CANDIDATE stop that the f64 objective referees)
    double stop_guard;
};

struct ProfSlot { double ms = 0; int64_t n = 0; };

struct nmfx_engine {
    int device = 0;
    int64_t m = 0, n = 0;          // logical local shape
    int k = 0, kp = 0;             // logical / padded rank
    int64_t mp = 0, np = 0;        // padded shape (multiples of 64)
    hipStream_t own_stream = nullptr, stream = nullptr;
    // device buffers
    float* V = nullptr;            // [mp][np]
    float* W[2] = {nullptr, nullptr};   // [mp][kp], double buffered
    float* H = nullptr;            // [kp][np]
    float* HHt = nullptr;          // [kp][kp]
    float* HHt_part = nullptr;     // [gram_splits][kp][kp]
    float* G_part = nullptr;       // [gram_splits][kp][kp]
    float* G_big = nullptr;        // [<= 256][kp][kp]: row-split slabs of the image-based Gram kernel (k padded to 128), folded into G_part
    float* Bt_chunk = nullptr;     // chunked exchange (nmfx_mur_phase_a_cols): the slabs of ONE column chunk of the H-side product
    int64_t Bt_chunk_cap = 0;
    int chunk_gslabs = 0;          // ... and the number of W^T W slabs its first chunk / the Gram kernel left
    float* kl_part = nullptr;      // MUR-KL, split-bf16: [np/64][kp] row-sum partials of H | [mp/64][kp] column-sum partials of W (the fused epilogues)
    int64_t kl_h_iter = -2;        // outer iteration whose H epilogue wrote the H partials and the H / H^T images (valid for iteration + 1 only)
    float* A_part = nullptr;       // [wsplit][mp][kp]
    float* B_part = nullptr;       // [hsplit][kp][np]
    double* obj_part = nullptr;    // [max blocks]
    int64_t obj_part_cap = 0;
    float* xf32 = nullptr;         // exchange: [kp*np | kp*kp | kp]
    double* xf64 = nullptr;        // exchange: [8] = objective partial, 4 inner-loop norm sums, 3 spare
    bool own_x = true;
    // k > 128 (kernels_generic.hip): objective partials per 128 x 128 tile, an m x kp / kp x n scratch, split-product slabs
    float* prox_keys = nullptr;    // l1inf prox on vectors beyond 32768 entries: the sort's global work area [k][Lpad]
    int64_t prox_keys_cap = 0;
    double* gx_part = nullptr;
    float* gx_d = nullptr;
    float* gx_s = nullptr;
    unsigned short* gx_gimg = nullptr;   // MUR-eu for k > 128 (r4): bf16 hi / lo images of the Gram matrix of a denominator product ([2][kp][kp], tiled)
    float* gx_r = nullptr;         // AO-ADMM for k > 128: right-hand side of a round
    double* gx_w64 = nullptr;      // ... the f64 work matrix of the Gauss-Jordan inversion
    double* gx_nrm = nullptr;      // ... norm partials of a round
    double* gx_nnls_work = nullptr; int64_t gx_nnls_cap = 0;   // ANLS for k > 128: per-block f64 systems of the passive-set solves
    // k > 128, split-bf16 products (kernels_generic.hip, gxb_* / gxt_*): bf16 hi / lo planes of V (rows m) and of V^T (rows n), TILED for the LDS-DMA kernel
    unsigned short* gxb_v[4] = {nullptr, nullptr, nullptr, nullptr};
    bool gxb_v_ready = false;      // the planes are those of the current V
    bool gxb_disabled = false;     // the planes did not fit beside V: this handle keeps the exact-f32 product kernel beyond k = 128 (noted)
    float* gxb_vt = nullptr;       // MUR-KL beyond k = 128: V^T [np][mp] in f32 (the quotient of the H side is formed transposed)
    bool gxb_vt_ready = false;
    unsigned short* gxb_q[2] = {nullptr, nullptr};   // ... bf16 hi / lo planes of the quotient V / (W H + 1e-9): [mp][np], then [np][mp]
    bool gxb_img_ready = false;    // Whi/Wlo[0], WThi/WTlo, Hhi/Hlo, HThi/HTlo are the images of the current (W, H) of the k > 128 MUR loop
    struct nmfx_comm* comm = nullptr;      // RCCL communicator of a row-sharded run (comm.hip), or none
    double* obj_hist = nullptr;    // device, capacity obj_cap
    int64_t obj_cap = 0;
    DevState* state = nullptr;
    // AO-ADMM / ADMM state
    float* dualW = nullptr; float* dualH = nullptr;     // like W / H
    float* auxW = nullptr;  float* auxH = nullptr;
    float* Minv = nullptr;         // [kp][kp] (G + rho I)^-1
    float* Pw = nullptr; float* Ph = nullptr;   // [kp][kp] fixed l2n prox operators (ADMM)
    float* Asum = nullptr;         // [mp][kp] summed V H^T (ADMM)
    float* S = nullptr; float* DV = nullptr;    // [mp][np] KL-ADMM: v_aux + dual_v, dual_v
    float* kl_S[2] = {nullptr, nullptr}; float* kl_DV[2] = {nullptr, nullptr};   // ... split-bf16 form (r4): tile-major, [0] rows n (like Vt), [1] rows m (like Vtile)
    int kl_side = 0;               // the orientation that holds the current dual_v
    int kl_s_side = 0;             // ... and the one whose buffer holds the current S = v_aux + dual_v
    bool kl_bt_ready = false;      // ADMM-KL: Bt_part holds w_aux^T S of the coming iteration (left by the fused auxiliaries launch)
    // split-bf16 mode (kernels_bf16.hip): V^T and bf16 hi/lo images of the factors
    int precision = 0;             // 0 = f32 MFMA, 1 = split bf16 (k padded to 64 only)
    bool bf_ready = false;
    bool fused_pack = false;       // nmfx_mur_run (single GPU): no pack launch, h_update reads the slabs
    int xrank = -1, xworld = 0;    // nmfx_set_exchange_rank: the objective partial travels INSIDE the f32 exchange buffer (see NMFX_XTAIL)
    int ncu = 256, bt_split = 1, bf_wsplit = 1;
    int gram_ng_w = 1, gram_ng_h = 1;   // row blocks sharing the Gram by-product of the W / H phase (kp = 64)
    int64_t obj_count = 0;         // entries of obj_part the last objective-producing launch wrote
    const int* xyt_flag2 = nullptr;  // a second skip flag for the next 32-row product launches (the inner stop of a KL-ADMM sub-problem)
    int xyt_xpriv = 0;                // the next 32-row product launches read an X that lies in the KL auxiliaries' register order (kl_dv_pos)
    int xyt_nw = 8;                // waves per block of the next 32-row product launch (4: 64-row blocks, two per CU; set and reset by the caller)
    int ao_a_slabs = 0;            // AO-ADMM W side: slabs of A_part the fused inner kernel adds itself (0: auxW holds the sum)
    const float* ao_b_src = nullptr; const int* ao_b_cnt = nullptr;   // AO-ADMM H side, behind a stream-K product: B^T slabs the fused rounds sum themselves (+ what their first launch records)
    int ao_rec_nobj = 0; int64_t ao_rec_j = 0, ao_rec_min_iter = 0; double ao_rec_tol1 = 0.0, ao_rec_tol2 = 0.0;
    const float* ao_a_src = nullptr; const int* ao_a_cnt = nullptr;   // ... the slab buffer (default A_part) and, behind a stream-K product, the slabs per 128-row block
    bool wimg_ok = false;          // Whi/Wlo[0] and WThi/WTlo are the images of the current W[0] (AO-ADMM: left by the fused W-side launches)
    bool ao_images = false;        // the fused round kernels being launched write the images of the factor they update
    bool himg_both = false;        // Hhi/Hlo AND HThi/HTlo are the images of the current H (AO-ADMM skips a rebuild)
    bool lazy_objective = false;   // AO-ADMM split-bf16: the objective of the current pair rides on the next H-side product
    bool drop_v = false;           // split-bf16 mode: free the row-major V once Vtile / Vt exist (rebuilt on demand, nmfx_need_v)
    std::string note;              // what nmfx_create decided on its own (precision fallback, dropped V): nmfx_get_note
    float* Vtile = nullptr;        // V, tile-major: [mp/128][np/64] tiles of [128][64] (bf16-path W phase)
    float* Vt = nullptr;           // V^T, tile-major: [np/128][mp/64] tiles of [128][64] (bf16-path H phase)
    float* Bt_part = nullptr;      // [bt_split][np][kp]
    unsigned short *Whi[2] = {nullptr, nullptr}, *Wlo[2] = {nullptr, nullptr};   // [mp][kp]
    unsigned short *WThi = nullptr, *WTlo = nullptr;                              // [kp][mp]
    unsigned short *Hhi = nullptr, *Hlo = nullptr;   // [kp][np]
    unsigned short* minv_img = nullptr;             // r4: the three bf16 images of M^-1 ([3][kp][kp]) of the any-rank rounds beyond k = 128
    unsigned short *HThi = nullptr, *HTlo = nullptr; // [np][kp], only where H^T is the Z operand (AO-ADMM's fused objective)
    double* nrm_part = nullptr;    // [blocks][4]
    double* nrm_rounds = nullptr;  // [admm_iter][blocks][4]: norm partials of the fused inner rounds
    double* nnls_ginv = nullptr;   // ANLS: f64 inverse of the half-step's Gram matrix [kp][kp], then { int inv_bad } behind it
    int* nnls_todo = nullptr;      // ANLS: per problem, 1 = left to the elimination kernel
    int64_t nrm_rounds_cap = 0;
    float *bkX = nullptr, *bkU = nullptr;   // initial X, U of a fused sub-problem (restart point of the repair launch)
    int32_t* inner_hist = nullptr; int64_t inner_cap = 0;   // device [cap][2]
    // r4, stream-K form of the k = 128 Euclidean products (kernels_bf16.hip, xyt32_bf16_kernel<..., SK>): [0] H side (X = V^T), [1] W side (X = V)
    struct SkPlan {
        int4* seg = nullptr;       // device [nseg]: row block, first group, end group, slab
        int* first = nullptr;      // device [workers + 1]: a worker's segments
        int* cnt = nullptr;        // device [R / 128]: slabs of a row block (what the consumers sum)
        float* slabs = nullptr;    // device [maxslab][R][kp]
        int workers = 0, nseg = 0, maxslab = 0;
    } sk[2];
    // split configuration
    int wsplit = 1, hsplit = 1, gsplit = 1;
    bool have_v = false, have_f = false;
    int anls_dist = NMFX_EU;       // objective ANLS reports (anls.py:108,118): the iterates are least-squares either way
    bool anls_a_ready = false;    // ANLS: A_part / H H^T slabs of the CURRENT (W, H) are valid (produced by the fused objective pass)
    int wsel = 0;                  // W buffer holding the current iterate
    bool family_started = false;   // a MUR run has begun since nmfx_set_factors (pair mode must be chosen at its start)
    bool pair = false;             // nmfx_mur_pair_*: factor columns [0, 64) and [64, 128) are two independent problems
    int family = 0;                // solver family that has run since nmfx_set_factors (0 none, 1 MUR eu/kl, 2 AO-ADMM, 3 ADMM, 4 ANLS): nmfx_enter_family
    bool w_in_place = false;       // solver updates W[0] in place (all but MUR, which ping-pongs)
    // profiling
    bool prof = false;
    std::map<std::string, ProfSlot> prof_slots;
    std::vector<std::tuple<std::string, hipEvent_t, hipEvent_t>> prof_pending;
    std::string err;
};

#define NMFX_STR2(x) #x
#define NMFX_STR(x) NMFX_STR2(x)
#define NMFX_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    E->err = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" NMFX_STR(__LINE__) ")"; return NMFX_E_HIP; } } while (0)

// ---- launch helpers implemented in the kernel translation units ---------
// A_part[sp] = V(rows, cols of split sp) * H^T ; optionally the residual
// objective 0.5*sum (V - W H)^2 into obj_part (one double per block).
// Hsrc / Vsrc: use these instead of E->H / E->V; flag2: optional second "skip" flag
int nmfx_launch_wphase(nmfx_engine* E, const float* W, bool with_a, bool with_obj, bool kl = false,
                       const float* Hsrc = nullptr, const float* Vsrc = nullptr, const int* flag2 = nullptr);
int nmfx_bf16_gram_tn(nmfx_engine* E, int* slabs);   // W^T W from the transposed bf16 images of W (fresh after a W epilogue): G_part[*slabs][kp][kp]
int nmfx_bf16_gram_h(nmfx_engine* E, int* slabs);
int nmfx_launch_inverse64(nmfx_engine* E, const float* src, double diag_add, double* out64, int* soft_bad);   // f64 (src + diag_add I)^-1, kp 64 / 128
int nmfx_launch_inverse64_block(nmfx_engine* E, const double* src64, int64_t ld, double* out64);   // 128 x 128 f64 block -> its f64 inverse
int nmfx_launch_kl_vaux(nmfx_engine* E, const float* Wsrc, const float* Hsrc, const int* flag2 = nullptr);
// B_part[sr] = W^T V over the rows of split sr.
int nmfx_launch_hphase(nmfx_engine* E, const float* W, bool with_g, const float* Vsrc = nullptr,
                       const int* flag2 = nullptr);
bool nmfx_hphase_can_fuse_gram(const nmfx_engine* E);
void nmfx_phase_occupancy(int kp, int* wocc, int* hocc);
// number of W^T W partial slabs the H phase leaves in G_part
inline int nmfx_g_slabs(const nmfx_engine* E) { return nmfx_hphase_can_fuse_gram(E) ? E->hsplit : E->gsplit; }
// out_part[s] = X^T X  (X [rows][kp])  /  X X^T (X [kp][cols])
int nmfx_launch_gram_tn(nmfx_engine* E, const float* X, int64_t rows, float* out_part, int splits);
int nmfx_launch_gram_nt(nmfx_engine* E, const float* X, int64_t cols, int64_t ld, float* out_part, int splits);

// shared small launchers (kernels_mur.hip / engine.hip)
int nmfx_launch_sum_partials(nmfx_engine* E, const float* part, int splits, int64_t count, float* out);
int nmfx_launch_w_update(nmfx_engine* E, const float* Wold, float* Wnew, float lam, int splits);
int nmfx_launch_h_update(nmfx_engine* E, float lam, int64_t j, int64_t min_iter, double tol1, double tol2);
int nmfx_launch_pack(nmfx_engine* E, const int* flag2 = nullptr);   // xf32 = [sum B_part | sum G_part], xf64[0] = sum obj_part
int nmfx_launch_pack_from(nmfx_engine* E, const float* Bpart, int bsplit, const float* Gpart, int gsplit,
                          int64_t nobj);
int nmfx_launch_obj_reduce(nmfx_engine* E, int64_t nobj = 0, const double* src = nullptr);    // xf64[0] = sum obj_part (nobj 0: the f32 W phase's count)
bool nmfx_bf16_supported(const nmfx_engine* E);
inline int nmfx_bf16_hht_slabs(const nmfx_engine* E) { return E->gram_ng_w * E->bf_wsplit; }   // H H^T by-product slabs
inline int nmfx_bf16_g_slabs(const nmfx_engine* E) { return E->gram_ng_h * E->bt_split; }      // W^T W by-product slabs
int nmfx_bf16_prepare(nmfx_engine* E);
int nmfx_bf16_images_w(nmfx_engine* E, const float* W, int buf);
int nmfx_bf16_images_h(nmfx_engine* E, bool transposed, const float* src = nullptr);
int nmfx_mur_eu_phase_a_head_bf16(nmfx_engine* E, double lambda_w, int64_t j);
int nmfx_mur_eu_phase_a_cols_bf16(nmfx_engine* E, int64_t c0, int64_t c1);
int nmfx_bf16_kl_w_epilogue(nmfx_engine* E, const float* Wold, float* Wnew, int nxt, float lam, const float* rowsum);
int nmfx_bf16_kl_h_epilogue(nmfx_engine* E, float lam, int64_t j, int64_t min_iter, double tol1, double tol2);
int nmfx_bf16_vht(nmfx_engine* E, bool obj, int zbuf, const char* name, bool kl = false, int terms = 4);   // terms: kernels_bf16.hip, top
int nmfx_bf16_vtw(nmfx_engine* E, bool obj, const char* name, bool kl = false, int terms = 4);
int nmfx_bf16_objective(nmfx_engine* E, int zbuf, const char* name);
int nmfx_mur_kl_phase_a_bf16(nmfx_engine* E, double lambda_w, int64_t j);
int nmfx_mur_kl_phase_b_bf16(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_bf16_pack_t(nmfx_engine* E, const float* Gpart, int gsplit, int64_t nobj);
// r4: the stream-K product of side 0 (B^T slabs = V^T W, objective with Z = H^T images if obj) / 1 (A slabs = V H^T) with the
// inversion of (sum of gslabs slabs of gsrc) + rho I as its side job (gsrc = nullptr: none); consumers read E->sk[side]
int nmfx_bf16_sk_product(nmfx_engine* E, int side, bool obj, const float* gsrc, int gslabs, double fixed_rho, const char* name);
// the pack behind it: xf32 = [B^T sums transposed], xf64[0] = objective, recorded as obj[j] with the stop rule (and a pending "not
// positive definite" of the side job promoted unless the rule fired)
int nmfx_bf16_pack_sk(nmfx_engine* E, int64_t j, int64_t min_iter, double tol1, double tol2);
bool nmfx_sk_enabled(const nmfx_engine* E);
// KL-loss ADMM variants on the split-bf16 kernels (kernels_bf16.hip, r4)
int nmfx_bf16_kl_state(nmfx_engine* E, bool reset);
int nmfx_bf16_kl_orient(nmfx_engine* E, int side, bool with_dv, bool with_s = true);
int nmfx_bf16_vaux(nmfx_engine* E, int side, const int* flag2 = nullptr);
int nmfx_bf16_vaux_fused(nmfx_engine* E, int side, const int* flag2, const double* nrm, int nblk, bool last);   // + the next round's product (r5)
int nmfx_bf16_kl_objective(nmfx_engine* E);      // obj_part <- KL(V, W H) from the images of W[0] (buffer 0) and H; E->obj_count entries
int nmfx_bf16_kl_product(nmfx_engine* E, int side, int terms, const int* flag2 = nullptr, bool gather = false);
int nmfx_mur_eu_phase_a_bf16(nmfx_engine* E, double lambda_w, int64_t j);
int nmfx_mur_eu_phase_b_bf16(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_mur_eu_phase_b_slice_bf16(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j, int cb0, int nblk);
int nmfx_mur_eu_phase_b_rest_bf16(nmfx_engine* E, int cb0, int nblk);
int nmfx_finish_b(nmfx_engine* E, int64_t min_iter, double tol1, double tol2, int64_t j);
// One solver family per set of factors: the families keep different device state next to W and H (MUR: W ping-pong and bf16 images of
// both factors; AO-ADMM / ADMM: duals and auxiliaries; ANLS: warm-start supports), and a family that starts in the middle of another's
// run would read leftovers.  A second family on the same handle needs nmfx_get_factors -> nmfx_set_factors first (NMFX_E_STATE otherwise).
inline int nmfx_enter_family(nmfx_engine* E, int fam) {
    if (E->family && E->family != fam) {
        E->err = "another solver has run on this handle since nmfx_set_factors: read the factors back and set them again "
                 "(nmfx_get_factors, nmfx_set_factors) before a different solver continues from them";
        return NMFX_E_STATE;
    }
    E->family = fam;
    return NMFX_OK;
}
int nmfx_ensure_obj_capacity(nmfx_engine* E, int64_t need);
void nmfx_comm_free(nmfx_engine* E);      // comm.hip
void nmfx_comm_invalidate(nmfx_engine* E);   // nmfx_set_precision: renegotiate, drop the captured graph, objective back to the f64 buffer
// kernels_generic.hip: MUR for k > 128 (kp a multiple of 128), same phase protocol
int nmfx_generic_mur_phase_a(nmfx_engine* E, int distance, double lambda, int64_t j);
int nmfx_generic_mur_phase_b(nmfx_engine* E, int distance, double lambda, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_generic_mur_finish_a(nmfx_engine* E, int distance, int64_t j);
int nmfx_generic_aoadmm_run(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h, int admm_iter, int64_t min_iter,
                            double tol1, double tol2, int64_t first, int64_t count);
int nmfx_generic_aoadmm_kl_run(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h, int admm_iter, int64_t min_iter,
                               double tol1, double tol2, int64_t first, int64_t count);
int nmfx_generic_admm_run(nmfx_engine* E, int distance, double rho, int prox_w, double lam_w, int prox_h, double lam_h, int64_t min_iter,
                          double tol1, double tol2, int64_t first, int64_t count);
int nmfx_generic_anls_run(nmfx_engine* E, double lam_w, double lam_h, int64_t min_iter, double tol1, double tol2, int64_t first, int64_t count);
// row-sharded forms beyond 128 components (r4): ADMM phase 0 products / 1 update; ANLS phase 0 objective / 1 W half + products / 2 H half
int nmfx_generic_admm_phase(nmfx_engine* E, int phase, int distance, double rho, int prox_w, double lam_w, int prox_h, double lam_h,
                            int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_generic_anls_phase(nmfx_engine* E, int phase, double lam, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_preload_generic();
// the tuned kernels keep k x k matrices and k-wide panels on chip: everything but MUR ends at k = 128
inline int nmfx_small_k_only(nmfx_engine* E, const char* what) {
    if (E->kp <= 128) return NMFX_OK;
    E->err = std::string(what) + ": not available with more than 128 components in this build";
    return NMFX_E_ARG;
}
int nmfx_ensure_inner_capacity(nmfx_engine* E, int64_t need);

// AO-ADMM / ADMM building blocks (kernels_aoadmm.hip)
int nmfx_aoadmm_alloc(nmfx_engine* E);
int nmfx_admm_state_alloc(nmfx_engine* E);       // + the aux matrices of ADMM (kernels_admm.hip)
int nmfx_kl_state_alloc(nmfx_engine* E);
int nmfx_launch_prepare(nmfx_engine* E, const float* src, int record_obj, int64_t j, int64_t min_iter,
                        double tol1, double tol2, double fixed_rho);
int nmfx_inner_cols(nmfx_engine* E, const float* M, float* aux, int mode, int prox, float lam, int round);
int nmfx_inner_rows(nmfx_engine* E, const float* Asum, float* W, const float* M, float* aux, int mode, int prox,
                    float lam, int round, const double* nrm_global = nullptr);
int nmfx_inner_finish(nmfx_engine* E, int nblk, int admm_iter, int32_t* slot, const double* nrm_global = nullptr);
int nmfx_round_any(nmfx_engine* E, bool cols, const float* B, float* X, float* U, int prox, float lam, int round,
                   const double* nrm_global = nullptr);   // k padded to <= 512: one launch per round
int nmfx_gather_round_norms(nmfx_engine* E, int nblk, int round);
// row-sharded AO-ADMM (least-squares loss) beyond 128 components, phase 0 .. 4 = h_products, h_solve, w_products, w_round, w_close
int nmfx_generic_aoadmm_kl_phase(nmfx_engine* E, int phase, int prox, double lam, int admm_iter, int64_t min_iter, double tol1, double tol2,
                                 int64_t j, int round);
int nmfx_generic_aoadmm_phase(nmfx_engine* E, int phase, int prox, double lam, int admm_iter, int64_t min_iter, double tol1, double tol2,
                              int64_t j, int round);

// Row-major V for the kernels that read it (exact-f32 products, KL auxiliaries, the SVD): in split-bf16 mode it may
// have been freed after the tile-major copies were built (drop_v) and is then rebuilt from Vtile.
int nmfx_need_v(nmfx_engine* E);

// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (device, kernel):
// the attribute belongs to the function ON A DEVICE, so the bookkeeping is keyed by both and guarded by a
// mutex -- independent handles may launch from different threads and on different devices.
int nmfx_allow_lds(nmfx_engine* E, const void* kernel, int bytes);

// prox 'l1inf' / 'l1inf_transpose' of ADMM (kernels_prox.hip): X = prox(X_aux, dual) on the W or the H side
int nmfx_launch_prox_l1inf(nmfx_engine* E, bool h_side, bool transpose, double rho, double lam, double ub, bool update_dual, bool ao = false);

int nmfx_preload_bf16();
int nmfx_preload_products();
int nmfx_preload_mur();
int nmfx_preload_kl();
int nmfx_preload_aoadmm();
int nmfx_preload_anls();
int nmfx_preload_svd();
int nmfx_preload_prox();

struct ProfScope {
    nmfx_engine* E; hipEvent_t a = nullptr, b = nullptr; const char* name;
    ProfScope(nmfx_engine* e, const char* nm) : E(e), name(nm) {
        if (E->prof) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, E->stream); }
    }
    ~ProfScope() {
        if (E->prof) { (void)hipEventRecord(b, E->stream); E->prof_pending.emplace_back(name, a, b); }
    }
};
