// AO-ADMM (Huang, Sidiropoulos, Liavas) sub-problem kernels.
//   reference: admm_ls_update nmf/ao_admm.py:46-68, prox nn/l1n :113-124,
//   terminate :33-43, driver :259-301.
//
// Per sub-problem:  G = F^T F (k x k), rho = trace(G)/k, then up to admm_iter
// rounds of   aux = (G + rho I)^-1 (B + rho (X + U));  X = prox(aux - U);
// U += X - aux;  stop when ||X-aux||/||X|| < 1e-2 and ||X-X_prev||/||U|| < 1e-2.
//
// The reference factors G + rho I once (Cholesky) and back-substitutes every
// round.  Here the k x k inverse is formed once per sub-problem in float64 by a
// single workgroup (in-place Gauss-Jordan in LDS; the pivots are the Cholesky
// pivots squared, so "not positive definite" is detected on the same
// condition), rounded to f32, and every round is then ONE fused kernel: an MFMA
// product with that inverse + prox + dual update + the four norm partials.
// cond(G + rho I) <= k + 1 because rho >= lambda_max / k, so the explicit
// inverse is benign.
//
// The inner stop test runs on the device: round i first re-derives the decision
// of round i-1 from the per-block norm partials (every block sums the same
// numbers in the same order, so all agree) and turns into a no-op once it fired.
#include "nmfx_internal.h"
#include "kernels_small.h"
#include <cstdlib>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- k x k inverse -------------------------------------------------------
// src: summed Gram matrix (f32, [KP][KP], zero beyond the logical k).
// record_obj: also publish obj[j] and run the outer convergence test first.
// The matrix lives in registers: thread (ti, tj) of a (KP/4) x (KP/4) grid owns the 4 x 4
// patch rows 4ti.., columns 4tj...  Per pivot p the owners publish row p, column p and
// 1/piv through double-buffered LDS (one barrier per pivot) and every thread applies
// ONE rank-1 FMA update to its patch.  The in-place Gauss-Jordan step
//     a[p][p] <- 1/piv,  a[p][c] <- a[p][c]/piv,  a[i][p] <- -a[i][p]/piv,
//     a[i][c] <- a[i][c] - a[i][p] a[p][c]/piv
// is exactly  a[i][c] <- a[i][c] - cv[i] * rv[c]  with the published values adjusted to
//     cv[p] = piv - 1   and   rv[p] = 1 + 1/piv      (rv[c] = a[p][c]/piv otherwise),
// so there are no special cases in the update (the kernel is f64 issue bound).
template <int KP>
__global__ __launch_bounds__((KP / 4) * (KP / 4) < 64 ? 64 : (KP / 4) * (KP / 4)) void ao_prepare_kernel(
    const float* __restrict__ src, int k, float* __restrict__ Minv, DevState* __restrict__ st,
    int record_obj, const double* __restrict__ xf64, long long j, long long min_iter,
    double tol1, double tol2, double* __restrict__ obj_hist, double fixed_rho)
{
    if (st->flag) return;
    if (record_obj) {
        const int rule = nmfx_record_objective(st, obj_hist, xf64[0], j, min_iter, tol1, tol2,
                                               threadIdx.x == 0);
        if (rule) return;
    }
    constexpr int T = KP / 4;
    __shared__ double rowr[2][KP], colp[2][KP], pivinv[2][2], shrho;
    const int tid = threadIdx.x;
    const bool active = tid < T * T;
    const int ti = active ? tid / T : -1, tj = active ? tid % T : -1;
    double a[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 v = active ? *reinterpret_cast<const float4*>(src + (int64_t)(4 * ti + r) * KP + 4 * tj)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        a[r][0] = v.x; a[r][1] = v.y; a[r][2] = v.z; a[r][3] = v.w;
    }
    if (tid == 0) {
        double rho = fixed_rho;
        if (!(fixed_rho >= 0.0)) {                 // AO-ADMM: rho = trace(G) / k  (ao_admm.py:54)
            double tr = 0.0;
            for (int i = 0; i < k; ++i) tr += (double)src[(int64_t)i * KP + i];
            rho = tr / (double)k;
        }
        shrho = rho;
        st->rho = rho; st->inner_stop = 0; st->inner_count = 0;
    }
    __syncthreads();
    const double rho = shrho;
    if (active && ti == tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r][r] += rho;
    }
    for (int p = 0; p < KP; ++p) {
        const int buf = p & 1, pg = p >> 2, pr = p & 3;
        if (ti == pg) {                            // owners of row p
            double rowv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                rowv[c] = pr == 0 ? a[0][c] : pr == 1 ? a[1][c] : pr == 2 ? a[2][c] : a[3][c];
                rowr[buf][4 * tj + c] = rowv[c];
            }
            if (tj == pg) {                        // the pivot owner divides once for everybody
                const double piv = pr == 0 ? rowv[0] : pr == 1 ? rowv[1] : pr == 2 ? rowv[2] : rowv[3];
                pivinv[buf][0] = piv;
                pivinv[buf][1] = 1.0 / piv;
            }
        }
        if (tj == pg) {                            // owners of column p
#pragma unroll
            for (int r = 0; r < 4; ++r)
                colp[buf][4 * ti + r] = pr == 0 ? a[r][0] : pr == 1 ? a[r][1] : pr == 2 ? a[r][2] : a[r][3];
        }
        __syncthreads();
        const double piv = pivinv[buf][0];
        if (!(piv > 0.0)) {                        // scipy cholesky would raise LinAlgError
            if (tid == 0) { st->notpd = 1; st->flag = 3; }
            return;
        }
        const double inv = pivinv[buf][1];
        double rv[4], cv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rv[c] = (4 * tj + c == p) ? 1.0 + inv : rowr[buf][(4 * tj + c) & (KP - 1)] * inv;
            cv[c] = (4 * ti + c == p) ? piv - 1.0 : colp[buf][(4 * ti + c) & (KP - 1)];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[r][c] = fma(-cv[r], rv[c], a[r][c]);
    }
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(Minv + (int64_t)(4 * ti + r) * KP + 4 * tj) =
                make_float4((float)a[r][0], (float)a[r][1], (float)a[r][2], (float)a[r][3]);
    }
}

#define NMFX_PREP_STAMPS_HERE
#include "prepare_body.h"

// the stand-alone launch: one workgroup of KP / 16 row waves + the helper wave
template <int KP>
__global__ __launch_bounds__(KP * 4 + 64) void ao_prepare_mfma_kernel(
    const float* __restrict__ src, int k, float* __restrict__ Minv, DevState* __restrict__ st,
    int record_obj, const double* __restrict__ xf64, long long j, long long min_iter,
    double tol1, double tol2, double* __restrict__ obj_hist, double fixed_rho,
    double* __restrict__ out64 = nullptr, int* __restrict__ soft_bad = nullptr,
    const double* __restrict__ src64 = nullptr, int64_t ld64 = 0)
{
    if (st->flag) return;
    if (record_obj) {
        const int rule = nmfx_record_objective(st, obj_hist, xf64[0], j, min_iter, tol1, tol2,
                                               threadIdx.x == 0);
        if (rule) return;
    }
    extern __shared__ __attribute__((aligned(16))) double prep_lds[];
    ao_prepare_body<KP, true>(prep_lds, src, 1, k, Minv, st, fixed_rho, false, out64, soft_bad, src64, ld64);
}

// (inner_test / inner_round_fired: kernels_small.h -- the fused auxiliaries launch of the KL loss applies the same test, r5)
__device__ __forceinline__ float prox_apply(float aux, float dual, float shift) {
    const float d = (aux - dual) - shift;              // nn: shift = 0; l1n: lambda / rho
    return (d < 0.f) ? 0.f : d;                        // np.where(d < 0, 0, d): NaN stays NaN
}

// wave64 sum of an f32 on the DPP network (quad swaps, row mirrors, row broadcasts: ~6 VALU moves instead of
// 12 dependent LDS-crossbar shuffles for an f64); the total lands in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define NMFX_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xf, false))
    NMFX_DPP_ADD(0xB1, 0xf);       // quad_perm [1,0,3,2]
    NMFX_DPP_ADD(0x4E, 0xf);       // quad_perm [2,3,0,1]
    NMFX_DPP_ADD(0x141, 0xf);      // row_half_mirror
    NMFX_DPP_ADD(0x140, 0xf);      // row_mirror: every lane of a row of 16 holds the row sum
    NMFX_DPP_ADD(0x142, 0xa);      // row_bcast15 into rows 1 and 3
    NMFX_DPP_ADD(0x143, 0xc);      // row_bcast31 into rows 2 and 3: lane 63 holds the wave sum
#undef NMFX_DPP_ADD
    return v;
}

// The four norm sums of a block: per-thread f32 partials -> f32 wave sums -> f64 across the waves.  (The
// per-round norm reduction is on the serial path of every inner-round kernel: with f64 shuffles it was
// ~1.3 us of each round.)  Used by every inner-round kernel, so all of them see the same numbers.
template <int NW>
__device__ __forceinline__ void block_store_norms(float n0, float n1, float n2, float n3,
                                                  double* __restrict__ out, double* sh)
{
    const float w0 = wave_sum_to_lane63(n0), w1 = wave_sum_to_lane63(n1);
    const float w2 = wave_sum_to_lane63(n2), w3 = wave_sum_to_lane63(n3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) { sh[wave * 4 + 0] = (double)w0; sh[wave * 4 + 1] = (double)w1; sh[wave * 4 + 2] = (double)w2; sh[wave * 4 + 3] = (double)w3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += sh[w * 4 + threadIdx.x];
        out[threadIdx.x] = t;
    }
}

// ---- H-side round: X is [KP][np] (columns independent), aux = Minv * RHS ---
// block = 64 columns; wave w owns row tiles {w, w+4, ...}; RHS tile through LDS.
// mode 0: the round described above (aux additionally stored when AUX != null);
// mode 1: aux = Minv * (B + rho (X + U)) stored to AUX only            (ADMM, l2n step 1)
// mode 2: X = max(P * (AUX - U), 0), U += X - AUX with P in `Minv`     (ADMM, l2n step 2:
//         prox 'l2n' of nmf/admm.py:141-156 is a k x k solve = product with a fixed inverse)
template <int KP>
__global__ __launch_bounds__(256) void ao_inner_cols_kernel(
    const float* __restrict__ Bsum, float* __restrict__ X, float* __restrict__ U,
    const float* __restrict__ Minv, float* __restrict__ AUX, int mode, int64_t np, int prox, float lam,
    int round, DevState* __restrict__ st, double* __restrict__ nrm)       // nrm: [2][nblk][4]
{
    if (st->flag || st->inner_stop) return;
    constexpr int JT = KP / 16;
    constexpr int ITW = (JT + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];    // [KP][64] RHS + 16 doubles
    double* sh = reinterpret_cast<double*>(lds + KP * 64);
    const int nblk = gridDim.x;
    if (round > 0 && inner_round_fired(nrm + (int64_t)((round - 1) & 1) * nblk * 4, nblk, sh)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_stop = 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_count = round + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const float rho = (float)st->rho;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;

    {   // RHS = B + rho (X + U)
        const int srow = tid >> 4, sc = tid & 15;
#pragma unroll
        for (int p = 0; p < JT; ++p) {
            const int64_t g = (int64_t)(p * 16 + srow) * np + c0 + 4 * sc;
            const float4 u = *reinterpret_cast<const float4*>(U + g);
            float4 r;
            if (mode == 2) {
                const float4 a = *reinterpret_cast<const float4*>(AUX + g);
                r.x = a.x - u.x; r.y = a.y - u.y; r.z = a.z - u.z; r.w = a.w - u.w;
            } else {
                const float4 b = *reinterpret_cast<const float4*>(Bsum + g);
                const float4 h = *reinterpret_cast<const float4*>(X + g);
                r.x = b.x + rho * (h.x + u.x); r.y = b.y + rho * (h.y + u.y);
                r.z = b.z + rho * (h.z + u.z); r.w = b.w + rho * (h.w + u.w);
            }
            *reinterpret_cast<float4*>(lds + (p * 16 + srow) * 64 + 4 * sc) = r;
        }
    }
    __syncthreads();

    f32x4 acc[ITW][4];
#pragma unroll
    for (int r = 0; r < ITW; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
    for (int r = 0; r < ITW; ++r) {
        const int it = wave + 4 * r;
        if (it < JT) {
            float4 mf[JT];
#pragma unroll
            for (int u = 0; u < JT; ++u)
                mf[u] = *reinterpret_cast<const float4*>(Minv + (int64_t)(16 * it + x) * KP + 16 * u + 4 * q);
#pragma unroll
            for (int u = 0; u < JT; ++u) {
                const float ma[4] = {mf[u].x, mf[u].y, mf[u].z, mf[u].w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 rb = *reinterpret_cast<const float4*>(lds + (16 * u + 4 * q + s) * 64 + 4 * x);
                    acc[r][0] = MFMA(ma[s], rb.x, acc[r][0]);
                    acc[r][1] = MFMA(ma[s], rb.y, acc[r][1]);
                    acc[r][2] = MFMA(ma[s], rb.z, acc[r][2]);
                    acc[r][3] = MFMA(ma[s], rb.w, acc[r][3]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t idx = (int64_t)(16 * it + 4 * q + g) * np + c0 + 4 * x;
                float ax[4] = {acc[r][0][g], acc[r][1][g], acc[r][2][g], acc[r][3][g]};
                if (mode == 1) {
                    *reinterpret_cast<float4*>(AUX + idx) = make_float4(ax[0], ax[1], ax[2], ax[3]);
                    continue;
                }
                const float4 h = *reinterpret_cast<const float4*>(X + idx);
                const float4 u = *reinterpret_cast<const float4*>(U + idx);
                const float hx[4] = {h.x, h.y, h.z, h.w}, ux[4] = {u.x, u.y, u.z, u.w};
                float hn[4], un[4];
                float zx[4] = {ax[0], ax[1], ax[2], ax[3]};          // mode 2: P (aux - U)
                if (mode == 2) {
                    const float4 a = *reinterpret_cast<const float4*>(AUX + idx);
                    ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w;
                } else if (AUX) {
                    *reinterpret_cast<float4*>(AUX + idx) = make_float4(ax[0], ax[1], ax[2], ax[3]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hn[e] = (mode == 2) ? ((zx[e] < 0.f) ? 0.f : zx[e]) : prox_apply(ax[e], ux[e], shift);
                    un[e] = ux[e] + hn[e] - ax[e];
                    const float d0 = hn[e] - ax[e], d2 = hn[e] - hx[e];
                    n0 += d0 * d0; n1 += hn[e] * hn[e]; n2 += d2 * d2; n3 += un[e] * un[e];
                }
                *reinterpret_cast<float4*>(X + idx) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                *reinterpret_cast<float4*>(U + idx) = make_float4(un[0], un[1], un[2], un[3]);
            }
        }
    }
    block_store_norms<4>(n0, n1, n2, n3, nrm + ((int64_t)(round & 1) * nblk + blockIdx.x) * 4, sh);
}

// ---- W-side round: X is [mp][KP] (rows independent), aux = RHS * Minv ------
// block = 64 rows (4 waves x 16); Minv through LDS (row stride KP + 4).
template <int KP>
__global__ __launch_bounds__(256) void ao_inner_rows_kernel(
    const float* __restrict__ Asum, float* __restrict__ X, float* __restrict__ U,
    const float* __restrict__ Minv, float* __restrict__ AUX, int mode, int prox, float lam, int round,
    DevState* __restrict__ st, double* __restrict__ nrm, const double* __restrict__ nrm_global)
{
    // nrm: [2][nblk][4] partials of this launch's blocks.  The previous round's norms are their
    // sum, or -- row-sharded runs, where the rows of other ranks count too (ao_admm.py:33-43
    // takes norms of the whole factor) -- the all-reduced sums in nrm_global[4].
    if (st->flag || st->inner_stop) return;
    constexpr int JT = KP / 16;
    constexpr int LDM = KP + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];    // [KP][LDM] + 16 doubles
    double* sh = reinterpret_cast<double*>(lds + KP * LDM);
    const int nblk = gridDim.x;
    if (round > 0 && (nrm_global ? inner_round_fired(nrm_global, 1, sh)
                                 : inner_round_fired(nrm + (int64_t)((round - 1) & 1) * nblk * 4, nblk, sh))) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_stop = 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_count = round + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const float rho = (float)st->rho;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;
    for (int i = tid; i < KP * (KP / 4); i += 256) {
        const int r = i / (KP / 4), c4 = i % (KP / 4);
        *reinterpret_cast<float4*>(lds + r * LDM + 4 * c4) =
            *reinterpret_cast<const float4*>(Minv + (int64_t)r * KP + 4 * c4);
    }
    const int64_t r0 = (int64_t)blockIdx.x * 64 + wave * 16;
    float4 xf[JT];
#pragma unroll
    for (int u = 0; u < JT; ++u) {
        const int64_t g = (r0 + x) * KP + 16 * u + 4 * q;
        const float4 d = *reinterpret_cast<const float4*>(U + g);
        if (mode == 2) {
            const float4 a = *reinterpret_cast<const float4*>(AUX + g);
            xf[u].x = a.x - d.x; xf[u].y = a.y - d.y; xf[u].z = a.z - d.z; xf[u].w = a.w - d.w;
        } else {
            const float4 a = *reinterpret_cast<const float4*>(Asum + g);
            const float4 w = *reinterpret_cast<const float4*>(X + g);
            xf[u].x = a.x + rho * (w.x + d.x); xf[u].y = a.y + rho * (w.y + d.y);
            xf[u].z = a.z + rho * (w.z + d.z); xf[u].w = a.w + rho * (w.w + d.w);
        }
    }
    __syncthreads();
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
    for (int it = 0; it < JT; ++it) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < JT; ++u) {
            const float4 mb = *reinterpret_cast<const float4*>(lds + (16 * it + x) * LDM + 16 * u + 4 * q);
            acc = MFMA(xf[u].x, mb.x, acc);
            acc = MFMA(xf[u].y, mb.y, acc);
            acc = MFMA(xf[u].z, mb.z, acc);
            acc = MFMA(xf[u].w, mb.w, acc);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t idx = (r0 + 4 * q + g) * KP + 16 * it + x;
            if (mode == 1) { AUX[idx] = acc[g]; continue; }
            const float w = X[idx], d = U[idx];
            float ax = acc[g];
            const float z = ax;
            if (mode == 2) ax = AUX[idx];
            else if (AUX) AUX[idx] = ax;
            const float wn = (mode == 2) ? ((z < 0.f) ? 0.f : z) : prox_apply(ax, d, shift);
            const float dn = d + wn - ax;
            const float d0 = wn - ax, d2 = wn - w;
            n0 += d0 * d0; n1 += wn * wn; n2 += d2 * d2; n3 += dn * dn;
            X[idx] = wn; U[idx] = dn;
        }
    }
    block_store_norms<4>(n0, n1, n2, n3, nrm + ((int64_t)(round & 1) * nblk + blockIdx.x) * 4, sh);
}

// ---- the same two rounds for ANY padded rank (r4: k > 128, kernels_generic.hip) -------------------------------------------------
// One launch per round beyond 128 components as well (r3: right-hand side, product, prox / dual / norms and `terminate` were four
// launches per round, 100 us where the arithmetic is 20).  Same protocol as the kernels above -- round r first re-derives the
// decision of round r - 1 from the norm partials, the norm partials of round r go to nrm[r & 1] --, same MFMA product on
// v_mfma_f32_16x16x4_f32 with a runtime contraction length: the right-hand side tile lives in LDS (kp x 64 floats on the H side,
// 64 x (kp + 4) on the W side: kp <= 512), the rows of M^-1 (symmetric) are the other operand straight from L2.
__global__ __launch_bounds__(256) void ao_round_cols_any_kernel(
    const float* __restrict__ Bsum, float* __restrict__ X, float* __restrict__ U, const float* __restrict__ Minv, int kp, int64_t np,
    int prox, float lam, int round, DevState* __restrict__ st, double* __restrict__ nrm)
{
    if (st->flag || st->inner_stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];    // [kp][64] RHS + 16 doubles
    double* sh = reinterpret_cast<double*>(lds + (int64_t)kp * 64);
    const int nblk = gridDim.x, JT = kp / 16;
    if (round > 0 && inner_round_fired(nrm + (int64_t)((round - 1) & 1) * nblk * 4, nblk, sh)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_stop = 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_count = round + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const float rho = (float)st->rho;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;
    {   // RHS = B + rho (X + U)
        const int srow = tid >> 4, sc = tid & 15;
        for (int p = 0; p < JT; ++p) {
            const int64_t g = (int64_t)(p * 16 + srow) * np + c0 + 4 * sc;
            const float4 u = *reinterpret_cast<const float4*>(U + g), b = *reinterpret_cast<const float4*>(Bsum + g);
            const float4 h = *reinterpret_cast<const float4*>(X + g);
            *reinterpret_cast<float4*>(lds + (p * 16 + srow) * 64 + 4 * sc) =
                make_float4(b.x + rho * (h.x + u.x), b.y + rho * (h.y + u.y), b.z + rho * (h.z + u.z), b.w + rho * (h.w + u.w));
        }
    }
    __syncthreads();
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
    for (int it = wave; it < JT; it += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* mrow = Minv + (int64_t)(16 * it + x) * kp + 4 * q;
        for (int u0 = 0; u0 < JT; u0 += 4) {           // (kp is a multiple of 128: JT of 8) four k-steps' operand loads in flight together
            float4 mf[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) mf[uu] = *reinterpret_cast<const float4*>(mrow + 16 * (u0 + uu));
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const float ma[4] = {mf[uu].x, mf[uu].y, mf[uu].z, mf[uu].w};
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const float4 rb = *reinterpret_cast<const float4*>(lds + (16 * (u0 + uu) + 4 * q + s2) * 64 + 4 * x);
                    acc[0] = MFMA(ma[s2], rb.x, acc[0]);
                    acc[1] = MFMA(ma[s2], rb.y, acc[1]);
                    acc[2] = MFMA(ma[s2], rb.z, acc[2]);
                    acc[3] = MFMA(ma[s2], rb.w, acc[3]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t idx = (int64_t)(16 * it + 4 * q + g) * np + c0 + 4 * x;
            const float ax[4] = {acc[0][g], acc[1][g], acc[2][g], acc[3][g]};
            const float4 h = *reinterpret_cast<const float4*>(X + idx), u = *reinterpret_cast<const float4*>(U + idx);
            const float hx[4] = {h.x, h.y, h.z, h.w}, ux[4] = {u.x, u.y, u.z, u.w};
            float hn[4], un[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hn[e] = prox_apply(ax[e], ux[e], shift);
                un[e] = ux[e] + hn[e] - ax[e];
                const float d0 = hn[e] - ax[e], d2 = hn[e] - hx[e];
                n0 += d0 * d0; n1 += hn[e] * hn[e]; n2 += d2 * d2; n3 += un[e] * un[e];
            }
            *reinterpret_cast<float4*>(X + idx) = make_float4(hn[0], hn[1], hn[2], hn[3]);
            *reinterpret_cast<float4*>(U + idx) = make_float4(un[0], un[1], un[2], un[3]);
        }
    }
    block_store_norms<4>(n0, n1, n2, n3, nrm + ((int64_t)(round & 1) * nblk + blockIdx.x) * 4, sh);
}

__global__ __launch_bounds__(256) void ao_round_rows_any_kernel(
    const float* __restrict__ Asum, float* __restrict__ X, float* __restrict__ U, const float* __restrict__ Minv, int kp,
    int prox, float lam, int round, DevState* __restrict__ st, double* __restrict__ nrm, const double* __restrict__ nrm_global)
{
    // (nrm_global: row-sharded runs -- the previous round's norm sums over ALL ranks' rows, as in ao_inner_rows_kernel)
    if (st->flag || st->inner_stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];    // RHS [64][kp + 4] + 16 doubles
    const int LD = kp + 4, JT = kp / 16;
    double* sh = reinterpret_cast<double*>(lds + (int64_t)64 * LD);
    const int nblk = gridDim.x;
    if (round > 0 && (nrm_global ? inner_round_fired(nrm_global, 1, sh)
                                 : inner_round_fired(nrm + (int64_t)((round - 1) & 1) * nblk * 4, nblk, sh))) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_stop = 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_count = round + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const float rho = (float)st->rho;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;
    const int64_t rb0 = (int64_t)blockIdx.x * 64;
    for (int i = tid; i < 64 * (kp / 4); i += 256) {   // RHS = A + rho (X + U)
        const int r = i / (kp / 4), c4 = i % (kp / 4);
        const int64_t g = (rb0 + r) * kp + 4 * c4;
        const float4 a = *reinterpret_cast<const float4*>(Asum + g), w = *reinterpret_cast<const float4*>(X + g);
        const float4 d = *reinterpret_cast<const float4*>(U + g);
        *reinterpret_cast<float4*>(lds + r * LD + 4 * c4) =
            make_float4(a.x + rho * (w.x + d.x), a.y + rho * (w.y + d.y), a.z + rho * (w.z + d.z), a.w + rho * (w.w + d.w));
    }
    __syncthreads();
    // wave w takes the column tiles w, w + 4, ... of all 64 rows: the rows of M^-1 it multiplies with come from L2 ONCE per block
    // (with a wave per 16 rows every wave read all of M^-1: 52 us per round at k = 256 against 42 for the four launches it replaced)
    const float* xrow = lds + x * LD + 4 * q;           // + 16 rt rows, + 16 u
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
    for (int it = wave; it < JT; it += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* mrow = Minv + (int64_t)(16 * it + x) * kp + 4 * q;
        for (int u0 = 0; u0 < JT; u0 += 4) {
            float4 mb[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) mb[uu] = *reinterpret_cast<const float4*>(mrow + 16 * (u0 + uu));
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const float4 xf = *reinterpret_cast<const float4*>(xrow + (16 * rt) * LD + 16 * (u0 + uu));
                    acc[rt] = MFMA(xf.x, mb[uu].x, acc[rt]);
                    acc[rt] = MFMA(xf.y, mb[uu].y, acc[rt]);
                    acc[rt] = MFMA(xf.z, mb[uu].z, acc[rt]);
                    acc[rt] = MFMA(xf.w, mb[uu].w, acc[rt]);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t idx = (rb0 + 16 * rt + 4 * q + g) * kp + 16 * it + x;
                const float w = X[idx], d = U[idx], ax = acc[rt][g];
                const float wn = prox_apply(ax, d, shift);
                const float dn = d + wn - ax;
                const float d0 = wn - ax, d2 = wn - w;
                n0 += d0 * d0; n1 += wn * wn; n2 += d2 * d2; n3 += dn * dn;
                X[idx] = wn; U[idx] = dn;
            }
    }
    block_store_norms<4>(n0, n1, n2, n3, nrm + ((int64_t)(round & 1) * nblk + blockIdx.x) * 4, sh);
}

// After the last round: record (rounds run | fired << 16) for this sub-problem.
__global__ __launch_bounds__(256) void ao_inner_finish_kernel(
    DevState* __restrict__ st, const double* __restrict__ nrm, int nblk, int admm_iter,
    int32_t* __restrict__ slot, const double* __restrict__ nrm_global)
{
    if (st->flag) return;
    __shared__ double sh[16];
    int fired = st->inner_stop;
    const int count = st->inner_count;
    if (!fired && count == admm_iter && admm_iter > 0)
        fired = (nrm_global ? inner_round_fired(nrm_global, 1, sh)
                            : inner_round_fired(nrm + (int64_t)((admm_iter - 1) & 1) * nblk * 4, nblk, sh)) ? 1 : 0;
    if (threadIdx.x == 0) { *slot = count | (fired << 16); st->inner_stop = 0; }
}

// The decision of a speculative launch, derived identically by EVERY block of the launch that follows it:
// the first round whose four norm sums satisfy `terminate` (ao_admm.py:33-43), i.e. how many rounds count.
// Wave w sums component w of up to 8 rounds at a time (their loads are in flight together); per (round,
// component) the per-lane strides and the shuffle tree are those of inner_round_fired, so the sums are
// the same numbers.  sh: 64 doubles.  256 or 512 threads.
__device__ __forceinline__ int fused_decide(const double* __restrict__ nrm_rounds, int nblk, int admm_iter,
                                            double* sh, int* fired_out)
{
    // Up to 16 rounds at a time with ALL their loads in flight together (4 block strides x 16 rounds per lane): the no-op
    // REPAIR launch of the common case (no early stop) is nothing but this function, and as 2 x 4 dependent round trips
    // plus the f64 square roots / divisions of the tests it took 17 us per launch -- 6 % of a config-3 iteration.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int count = admm_iter, fired = 0;
    for (int r0 = 0; r0 < admm_iter && !fired; r0 += 16) {
        double sums[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) sums[u] = 0.0;
        if (wave < 4) {                                // (blocks of 512 threads: waves 4..7 only wait)
            for (int b0 = lane; b0 < nblk; b0 += 256) {
                double t[4][16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        t[i][u] = (b0 + 64 * i < nblk && r0 + u < admm_iter)
                                      ? nrm_rounds[((int64_t)(r0 + u) * nblk + b0 + 64 * i) * 4 + wave] : 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i)            // (block order per lane as in inner_round_fired: the same sums)
#pragma unroll
                    for (int u = 0; u < 16; ++u) sums[u] += t[i][u];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sums[u] += __shfl_down(sums[u], off, 64);
                if (lane == 0) sh[u * 4 + wave] = sums[u];
            }
        }
        __syncthreads();
        bool hit = false;
        if (lane < 16 && r0 + lane < admm_iter)        // lane u tests round r0 + u (inner_test: no square roots, no divisions)
            hit = inner_test(sh[lane * 4 + 0], sh[lane * 4 + 1], sh[lane * 4 + 2], sh[lane * 4 + 3]);
        const unsigned long long mask = __ballot(hit);                          // the same in every wave
        if (mask) { count = r0 + __ffsll((long long)mask); fired = 1; }
        __syncthreads();
    }
    *fired_out = fired;
    return count;
}

// What one launch of a fused sub-problem kernel has to do.  Hinted speculation (r2): the round at which `terminate` fires
// hardly moves from one outer iteration to the next (config 3: 4, 4, 4, ... on the H side, 3, 3, 3, ... on the W side, of
// admm_iter = 10), so the first launch runs only r1 = the previous sub-problem's count of rounds (DevState::ao_hint, written
// by the launch of that sub-problem that DECIDED, into the slot of the other parity -- no launch reads a slot it writes).
//   phase 0: rounds [0, r1) from the live X, U; start saved; norm partials stored.
//   phase 1: decide over [0, r1).  Fired at the last of them (the usual case) or r1 = admm_iter without a stop: the result
//            stands.  Fired earlier: rerun from the saved start.  Not fired and r1 < admm_iter: CONTINUE -- save the state at
//            r1 and run [r1, admm_iter) speculatively.
//   phase 2: nothing (one flag read) unless phase 1 continued: then decide over all rounds and rerun [r1, count) from the
//            state saved at r1.  The launch that decides leaves the next hint.
// Every path performs the reference's rounds 0 .. count-1 in order with the same arithmetic (states pass through memory as
// the f32 values they are), so the hint changes the cost only.  hint_rd = nullptr: r1 = admm_iter, phases 0 and 1 are the
// speculate / repair pair of round 1 (the row-sharded W sub-problem, whose decision comes from the all-reduced table).
struct FusedPlan { int first, count; bool from_backup, save, norms; };
__device__ __forceinline__ bool fused_plan(int phase, int admm_iter, const int* __restrict__ hint_rd, int* __restrict__ hint_wr,
                                           const double* __restrict__ tab, int nblk, double* sh, DevState* __restrict__ st,
                                           int32_t* __restrict__ slot, FusedPlan& p)
{
    if (phase == 2 && !st->ao_continued) return false;                 // (written by the lead thread of phase 1: a launch ago)
    int r1 = admm_iter;
    if (hint_rd) { const int h = *hint_rd; if (h > 0 && h < admm_iter) r1 = h; }
    if (phase == 0) { p = FusedPlan{0, r1, false, true, true}; return true; }
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    int fired;
    if (phase == 1) {
        const int cnt = fused_decide(tab, nblk, r1, sh, &fired);
        if (!fired && r1 < admm_iter) {                // no stop among the hinted rounds: go on to the end, phase 2 decides
            if (lead) st->ao_continued = 1;
            p = FusedPlan{r1, admm_iter - r1, false, true, true};
            return true;
        }
        if (lead) {
            st->inner_count = cnt; st->inner_stop = 0; *slot = cnt | (fired << 16);
            st->ao_continued = 0; st->ao_paths[cnt >= r1 ? 0 : 1] += 1;
            if (hint_wr) *hint_wr = fired ? cnt : admm_iter;
        }
        if (cnt >= r1) return false;                   // the speculative result stands
        p = FusedPlan{0, cnt, true, false, false};
        return true;
    }
    const int cnt = fused_decide(tab, nblk, admm_iter, sh, &fired);    // (rounds below r1 did not fire: the first hit is >= r1)
    if (lead) {
        st->inner_count = cnt; st->inner_stop = 0; *slot = cnt | (fired << 16);
        st->ao_paths[cnt >= admm_iter ? 2 : 3] += 1;
        if (hint_wr) *hint_wr = fired ? cnt : admm_iter;
    }
    if (cnt >= admm_iter) return false;
    p = FusedPlan{r1, cnt - r1, true, false, false};
    return true;
}

// ---- split-bf16 pieces for the inner product of the fused W-side rounds (k padded to 64 / 128) ----
typedef __bf16 ao_bf16x8 __attribute__((ext_vector_type(8)));
union AoFrag8 { uint4 u; ao_bf16x8 v; };
#define AO_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a).v, (b).v, (c), 0, 0, 0)
__device__ __forceinline__ void ao_split2(float a, float b, unsigned& hi, unsigned& lo) {      // x = hi + lo, both bf16 (round to nearest even)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(a - ah), "v"(b - bh));
}
__device__ __forceinline__ void ao_split8(const float4& p, const float4& q, AoFrag8& hi, AoFrag8& lo) {
    ao_split2(p.x, p.y, hi.u.x, lo.u.x); ao_split2(p.z, p.w, hi.u.y, lo.u.y);
    ao_split2(q.x, q.y, hi.u.z, lo.u.z); ao_split2(q.z, q.w, hi.u.w, lo.u.w);
}
// Three images: x = hi + mid + lo EXACTLY (3 x 8 significant bits = the 24 of an f32; every remainder is exact in f32).
// The inner product of a round takes them with the six terms hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid (what is dropped
// is below 2^-24 of |x||y|): r3 -- with two images (16 bits, four terms) the error of aux = M^-1 rhs, amplified by the
// cancellation in that product (cond(G + rho I) up to k + 1), put ||WH - W_ref H_ref|| / ||V|| at 1.1e-4 on a k = 100
// problem whose H sub-problem is unregularised (tools/lab/ao_f32_state.py models it: 5.4e-5 -> 4.3e-6 with these six terms).
__device__ __forceinline__ void ao_split2x3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(mid) : "v"(ra), "v"(rb));
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(sa), "v"(sb));
}
__device__ __forceinline__ void ao_split8x3(const float4& p, const float4& q, AoFrag8& hi, AoFrag8& mid, AoFrag8& lo) {
    ao_split2x3(p.x, p.y, hi.u.x, mid.u.x, lo.u.x); ao_split2x3(p.z, p.w, hi.u.y, mid.u.y, lo.u.y);
    ao_split2x3(q.x, q.y, hi.u.z, mid.u.z, lo.u.z); ao_split2x3(q.z, q.w, hi.u.w, mid.u.w, lo.u.w);
}

// ---- r4: the any-rank rounds with the product aux = rhs M^-1 on the bf16 matrix cores (k padded to 256 / 384) ------------------------
// ao_round_*_any_kernel multiply in exact f32 (v_mfma_f32_16x16x4_f32): 64 x kp x kp MACs per block are 15 us of a 30-38 us round at
// kp = 256.  Here both operands are the three bf16 images of above (hi + mid + lo = the f32 value exactly), six terms, on
// v_mfma_f32_16x16x32_bf16: 0.375 of the matrix time.  M^-1's images are made once per sub-problem (ao_minv_images_kernel: [3][kp][kp],
// L2-resident, the operand fragments are 16-byte global loads, the next k-step's in flight under the MFMAs of the current one); the
// right-hand side tile of the block's 64 rows (W) / columns (H) is split once, cooperatively, into LDS ([3][64][kp + 8] bf16: rows 16
// bytes apart from a multiple of 128, so the sixteen rows of a fragment read fall on distinct bank groups).  One kernel for both
// sides: the accumulator tile is (entity, factor) either way, only the global addressing differs (COLS: X is [kp][np]).
__global__ __launch_bounds__(256) void ao_minv_images_kernel(const float* __restrict__ Minv, int kp, unsigned short* __restrict__ img, const DevState* __restrict__ st)
{
    if (st->flag) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, kk = (int64_t)kp * kp;
    if (8 * i >= kk) return;
    const float4 a = *reinterpret_cast<const float4*>(Minv + 8 * i), b = *reinterpret_cast<const float4*>(Minv + 8 * i + 4);
    AoFrag8 h, m, l;
    ao_split8x3(a, b, h, m, l);
    *reinterpret_cast<uint4*>(img + 8 * i) = h.u;
    *reinterpret_cast<uint4*>(img + kk + 8 * i) = m.u;
    *reinterpret_cast<uint4*>(img + 2 * kk + 8 * i) = l.u;
}

template <bool COLS, int NWV>
__global__ __launch_bounds__(64 * NWV) void ao_round_any_bf16_kernel(
    const float* __restrict__ Bsum, float* __restrict__ X, float* __restrict__ U, const unsigned short* __restrict__ Mimg, int kp, int64_t ld,
    int prox, float lam, int round, DevState* __restrict__ st, double* __restrict__ nrm, const double* __restrict__ nrm_global)
{
    // ld: COLS -- np (X, U, Bsum are [kp][np], the block's entities are 64 columns); else kp (they are [mp][kp], 64 rows)
    if (st->flag || st->inner_stop) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    const int LDR = kp + 8;                            // bf16 elements per entity row of an image
    unsigned short* rimg = reinterpret_cast<unsigned short*>(smem_r);              // [3][64][LDR]
    double* sh = reinterpret_cast<double*>(smem_r + (size_t)3 * 64 * LDR * 2);
    const int nblk = gridDim.x, JT = kp / 16, KS = kp / 32;
    if (round > 0 && (nrm_global ? inner_round_fired(nrm_global, 1, sh)
                                 : inner_round_fired(nrm + (int64_t)((round - 1) & 1) * nblk * 4, nblk, sh))) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_stop = 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->inner_count = round + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const float rho = (float)st->rho;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;
    const int64_t e0 = (int64_t)blockIdx.x * 64;
    const int64_t kk = (int64_t)kp * kp;
    // RHS = B + rho (X + U), split into its three images: entity r, factors 8 c8 .. + 7
    if (!COLS) {
#pragma unroll 4
        for (int i = tid; i < 64 * (kp / 8); i += 64 * NWV) {   // (several units' 6 loads each in flight together)
            const int r = i / (kp / 8), c8 = i % (kp / 8);
            const int64_t g = (e0 + r) * kp + 8 * c8;
            float4 t[2];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const float4 a = *reinterpret_cast<const float4*>(Bsum + g + 4 * hlf), w = *reinterpret_cast<const float4*>(X + g + 4 * hlf);
                const float4 d = *reinterpret_cast<const float4*>(U + g + 4 * hlf);
                t[hlf] = make_float4(a.x + rho * (w.x + d.x), a.y + rho * (w.y + d.y), a.z + rho * (w.z + d.z), a.w + rho * (w.w + d.w));
            }
            AoFrag8 h, m, l;
            ao_split8x3(t[0], t[1], h, m, l);
            *reinterpret_cast<uint4*>(rimg + r * LDR + 8 * c8) = h.u;
            *reinterpret_cast<uint4*>(rimg + (64 + r) * LDR + 8 * c8) = m.u;
            *reinterpret_cast<uint4*>(rimg + (128 + r) * LDR + 8 * c8) = l.u;
        }
    } else {
        for (int i = tid; i < 16 * (kp / 8); i += 64 * NWV) {                     // four columns x eight factors per unit
            const int cg = i & 15, c8 = i >> 4;
            float v[8][4];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int64_t g = (int64_t)(8 * c8 + t) * ld + e0 + 4 * cg;
                const float4 a = *reinterpret_cast<const float4*>(Bsum + g), w = *reinterpret_cast<const float4*>(X + g), d = *reinterpret_cast<const float4*>(U + g);
                v[t][0] = a.x + rho * (w.x + d.x); v[t][1] = a.y + rho * (w.y + d.y); v[t][2] = a.z + rho * (w.z + d.z); v[t][3] = a.w + rho * (w.w + d.w);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                AoFrag8 h, m, l;
                ao_split8x3(make_float4(v[0][c], v[1][c], v[2][c], v[3][c]), make_float4(v[4][c], v[5][c], v[6][c], v[7][c]), h, m, l);
                const int r = 4 * cg + c;
                *reinterpret_cast<uint4*>(rimg + r * LDR + 8 * c8) = h.u;
                *reinterpret_cast<uint4*>(rimg + (64 + r) * LDR + 8 * c8) = m.u;
                *reinterpret_cast<uint4*>(rimg + (128 + r) * LDR + 8 * c8) = l.u;
            }
        }
    }
    __syncthreads();
    // wave w takes the factor tiles w, w + NWV, .. of all 64 entities, ONE TILE AT A TIME: the tile's X / U values are requested first
    // and land under its 6 x 4 x kp / 32 MFMAs; the epilogue of tile r then runs beside the other waves' matrix work (all tiles'
    // products first and all epilogues behind them left every wave waiting for its 64 scattered loads at the same time)
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
    for (int it = wave; it < JT; it += NWV) {
        float xo[4][4], uo[4][4];                      // [rt][g]
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (COLS) {                                // four consecutive columns of factor row 16 it + x
                const int64_t idx = (int64_t)(16 * it + x) * ld + e0 + 16 * rt + 4 * q;
                const float4 h4 = *reinterpret_cast<const float4*>(X + idx), u4 = *reinterpret_cast<const float4*>(U + idx);
                xo[rt][0] = h4.x; xo[rt][1] = h4.y; xo[rt][2] = h4.z; xo[rt][3] = h4.w;
                uo[rt][0] = u4.x; uo[rt][1] = u4.y; uo[rt][2] = u4.z; uo[rt][3] = u4.w;
            } else {                                   // four consecutive factors of entity row 16 rt + x (the product is formed transposed, below)
                const int64_t idx = (e0 + 16 * rt + x) * kp + 16 * it + 4 * q;
                const float4 h4 = *reinterpret_cast<const float4*>(X + idx), u4 = *reinterpret_cast<const float4*>(U + idx);
                xo[rt][0] = h4.x; xo[rt][1] = h4.y; xo[rt][2] = h4.z; xo[rt][3] = h4.w;
                uo[rt][0] = u4.x; uo[rt][1] = u4.y; uo[rt][2] = u4.z; uo[rt][3] = u4.w;
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // M^-1 operand: row 16 it + x, factors 32 u + 8 q .. + 7 of each image; the next k-step's in flight under this one's MFMAs
        const unsigned short* mrow = Mimg + (int64_t)(16 * it + x) * kp + 8 * q;
        AoFrag8 mh, mm, ml, nh, nm, nl;
        mh.u = *reinterpret_cast<const uint4*>(mrow); mm.u = *reinterpret_cast<const uint4*>(mrow + kk); ml.u = *reinterpret_cast<const uint4*>(mrow + 2 * kk);
        for (int u = 0; u < KS; ++u) {
            const int un = u + 1 < KS ? u + 1 : u;
            nh.u = *reinterpret_cast<const uint4*>(mrow + 32 * un); nm.u = *reinterpret_cast<const uint4*>(mrow + kk + 32 * un);
            nl.u = *reinterpret_cast<const uint4*>(mrow + 2 * kk + 32 * un);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                // COLS: acc[rt][g] = entity (column) 16 rt + 4 q + g, factor 16 it + x; rows (W): the operands change places, so that
                // acc[rt][g] = factor 16 it + 4 q + g of entity (row) 16 rt + x -- four consecutive floats of X / U per lane either way
                const int off = (16 * rt + x) * LDR + 32 * u + 8 * q;
                AoFrag8 rh, rm, rl;
                rh.u = *reinterpret_cast<const uint4*>(rimg + off);
                rm.u = *reinterpret_cast<const uint4*>(rimg + 64 * LDR + off);
                rl.u = *reinterpret_cast<const uint4*>(rimg + 128 * LDR + off);
                if (COLS) {
                    acc[rt] = AO_MFMA_BF16(rh, mh, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(rm, mh, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(rh, mm, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(rl, mh, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(rh, ml, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(rm, mm, acc[rt]);
                } else {
                    acc[rt] = AO_MFMA_BF16(mh, rh, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(mh, rm, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(mm, rh, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(mh, rl, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(ml, rh, acc[rt]);
                    acc[rt] = AO_MFMA_BF16(mm, rm, acc[rt]);
                }
            }
            mh = nh; mm = nm; ml = nl;
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            float hn[4], un2[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float ax = acc[rt][g];
                hn[g] = prox_apply(ax, uo[rt][g], shift);
                un2[g] = uo[rt][g] + hn[g] - ax;
                const float d0 = hn[g] - ax, d2 = hn[g] - xo[rt][g];
                n0 += d0 * d0; n1 += hn[g] * hn[g]; n2 += d2 * d2; n3 += un2[g] * un2[g];
            }
            if (COLS) {
                const int64_t idx = (int64_t)(16 * it + x) * ld + e0 + 16 * rt + 4 * q;
                *reinterpret_cast<float4*>(X + idx) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                *reinterpret_cast<float4*>(U + idx) = make_float4(un2[0], un2[1], un2[2], un2[3]);
            } else {
                const int64_t idx = (e0 + 16 * rt + x) * kp + 16 * it + 4 * q;
                *reinterpret_cast<float4*>(X + idx) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                *reinterpret_cast<float4*>(U + idx) = make_float4(un2[0], un2[1], un2[2], un2[3]);
            }
        }
    }
    block_store_norms<NWV>(n0, n1, n2, n3, nrm + ((int64_t)(round & 1) * nblk + blockIdx.x) * 4, sh);
}

// ---- all rounds of a sub-problem in ONE launch ------------------------------
// A round only couples the blocks through `terminate` (four global norms, ao_admm.py:33-43).
// The fused kernels therefore run ALL admm_iter rounds speculatively with X, U (and the
// right-hand side) resident in registers / LDS, leave the per-round norm partials in
// nrm_rounds[round][block][4] and a copy of the initial X, U in the backup buffers.
// The same kernel is then launched again in REPAIR mode: every block first finds, from those partials,
// the round at which the reference would have stopped (fused_decide; block 0 records it); only if that is
// before the last round does it restart from the backup and run exactly that many rounds (the common
// case -- no early stop -- is a launch that reads 80 KiB per block and leaves).  Same arithmetic and
// inner counts as the round-by-round kernels above, 2 launches instead of admm_iter + 1.
//
// H side: block = CB columns (64, or 32 when 64-column blocks would leave CUs idle: the rounds are bound
// by the f32 MFMA rate of the CU a block runs on), wave w owns the factor tiles w, w + 4, ...; M^-1 staged
// in LDS once.  Lane (x, q) holds columns NE x .. NE x + NE - 1 (NE = CB / 16) of rows 16 it + 4 q + g.
template <int NE> struct VecN;
template <> struct VecN<4> { typedef float4 T; };
template <> struct VecN<2> { typedef float2 T; };
template <int NE> __device__ __forceinline__ void ldv(float* d, const float* p) {
    const typename VecN<NE>::T v = *reinterpret_cast<const typename VecN<NE>::T*>(p);
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int e = 0; e < NE; ++e) d[e] = f[e];
}
template <int NE> __device__ __forceinline__ void stv(float* p, const float* s) {
    typename VecN<NE>::T v;
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int e = 0; e < NE; ++e) f[e] = s[e];
    *reinterpret_cast<typename VecN<NE>::T*>(p) = v;
}

template <int KP, int CB>
__global__ __launch_bounds__(256) void ao_fused_cols_kernel(
    const float* __restrict__ Bsum, float* __restrict__ X, float* __restrict__ U, float* __restrict__ Xb,
    float* __restrict__ Ub, const float* __restrict__ Minv, int64_t np, int prox, float lam, int admm_iter,
    DevState* __restrict__ st, double* __restrict__ nrm_rounds, int phase, int32_t* __restrict__ slot,
    const int* __restrict__ hint_rd, int* __restrict__ hint_wr,
    unsigned short* __restrict__ ihi = nullptr, unsigned short* __restrict__ ilo = nullptr,       // r3: the bf16 images of the new H
    unsigned short* __restrict__ ithi = nullptr, unsigned short* __restrict__ itlo = nullptr,     // ([KP][np]) and of H^T ([np][KP])
    // r4, behind a stream-K product (no pack launch): the right-hand side is the sum of the B^T slabs Bt[p][np][KP], p < bcnt[column
    // block of 128], and the FIRST launch of the sub-problem records the objective (sum of nobj partials) as obj[j] and applies the
    // stop rule -- every block evaluates the same sum and the same rule (block 0 publishes), so all of them leave together
    const float* __restrict__ Bt = nullptr, const int* __restrict__ bcnt = nullptr, int64_t bstride = 0,
    const double* __restrict__ objpart = nullptr, int nobj = 0, double* __restrict__ obj_hist = nullptr, double* __restrict__ xf64 = nullptr,
    long long j = 0, long long min_iter = 0, double tol1 = 0.0, double tol2 = 0.0)
{
    if (st->flag) return;
    constexpr int JT = KP / 16;
    constexpr int ITW = (JT + 3) / 4;
    constexpr int LDM = KP + 4;
    constexpr int NE = CB / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];    // RHS [KP][CB] | M^-1 [KP][LDM] | 32 doubles
    float* ms = lds + KP * CB;
    double* sh = reinterpret_cast<double*>(ms + ((KP >= 64 && CB == 32) ? (3 * KP * KP) / 2 : KP * LDM));    // (split form: three bf16 images)
    const int nblk = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    if (objpart && phase == 0) {
        double t = 0.0;                                // (the pack kernel's order of additions)
        for (int i = tid; i < nobj; i += 256) t += objpart[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0) sh[wave] = t;
        __syncthreads();
        const double obj = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        const bool writer = blockIdx.x == 0 && tid == 0;
        const int rule = nmfx_record_objective(st, obj_hist, obj, j, min_iter, tol1, tol2, writer);
        const int bad = st->notpd_pending;             // (written by the side job of the launch before this one, never here)
        if (writer) { xf64[0] = obj; if (!rule && bad) { st->notpd = 1; st->flag = 3; } }
        if (rule || bad) return;
        __syncthreads();
    }
    FusedPlan plan;
    if (!fused_plan(phase, admm_iter, hint_rd, hint_wr, nrm_rounds, nblk, sh, st, slot, plan)) return;
    const int64_t c0 = (int64_t)blockIdx.x * CB;
    const float rho = (float)st->rho;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;
    // k padded to 64 / 128, 32-column blocks (r2): the product aux = M^-1 rhs on the bf16 matrix cores, both operands split,
    // four terms -- as on the W side.  M^-1 staged once as two swizzled bf16 images (A operand: its rows); the right-hand side
    // tile is kept TRANSPOSED in LDS ([column][factor] f32, 16-byte chunk c of the row of lane x at c ^ x), so that the B
    // operand -- 8 consecutive factors of one column -- is two 16-byte reads + a split.
    constexpr bool SPLIT = KP >= 64 && CB == 32;
    constexpr int KS = KP / 32;
    unsigned short* mhi = reinterpret_cast<unsigned short*>(ms);
    unsigned short* mmd = mhi + KP * KP;
    unsigned short* mlo = mmd + KP * KP;
    auto swzm = [](int r) { return KP == 128 ? (r & 15) : ((r & 15) >> 1); };
    if (SPLIT) {
        for (int i = tid; i < KP * (KP / 8); i += 256) {
            const int r = i / (KP / 8), c8 = i % (KP / 8);
            const float4 a4 = *reinterpret_cast<const float4*>(Minv + (int64_t)r * KP + 8 * c8);
            const float4 b4 = *reinterpret_cast<const float4*>(Minv + (int64_t)r * KP + 8 * c8 + 4);
            AoFrag8 h, md, l;
            ao_split8x3(a4, b4, h, md, l);
            const int pos = c8 ^ swzm(r);
            *reinterpret_cast<uint4*>(mhi + r * KP + 8 * pos) = h.u;
            *reinterpret_cast<uint4*>(mmd + r * KP + 8 * pos) = md.u;
            *reinterpret_cast<uint4*>(mlo + r * KP + 8 * pos) = l.u;
        }
    } else {
    for (int i = tid; i < KP * (KP / 4); i += 256) {
        const int r = i / (KP / 4), c4 = i % (KP / 4);
        *reinterpret_cast<float4*>(ms + r * LDM + 4 * c4) = *reinterpret_cast<const float4*>(Minv + (int64_t)r * KP + 4 * c4);
    }
    }
    float hx[ITW][4][NE], ux[ITW][4][NE], bx[ITW][4][NE];
    const float* srcX = plan.from_backup ? Xb : X;
    const float* srcU = plan.from_backup ? Ub : U;
#pragma unroll
    for (int r = 0; r < ITW; ++r) {
        const int it = wave + 4 * r;
        if (it < JT) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t idx = (int64_t)(16 * it + 4 * q + g) * np + c0 + NE * x;
                ldv<NE>(hx[r][g], srcX + idx);
                ldv<NE>(ux[r][g], srcU + idx);
                if (!Bt) ldv<NE>(bx[r][g], Bsum + idx);
                if (plan.save) {
                    stv<NE>(Xb + idx, hx[r][g]);
                    stv<NE>(Ub + idx, ux[r][g]);
                }
            }
        }
    }
    if (Bt) {                                          // slabs [column][factor]: the lane's four factors of a column are one 16-byte load
        const int np_ = bcnt[c0 >> 7];
#pragma unroll
        for (int r = 0; r < ITW; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < NE; ++e) bx[r][g][e] = 0.f;
        for (int p = 0; p < np_; ++p) {                // slab order; a slab's loads all in flight together
            float4 v[ITW][NE];
#pragma unroll
            for (int r = 0; r < ITW; ++r) {
                const int it = wave + 4 * r;
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    v[r][e] = it < JT ? *reinterpret_cast<const float4*>(Bt + (int64_t)p * bstride + (c0 + NE * x + e) * KP + 16 * it + 4 * q)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int r = 0; r < ITW; ++r)
#pragma unroll
                for (int e = 0; e < NE; ++e) { bx[r][0][e] += v[r][e].x; bx[r][1][e] += v[r][e].y; bx[r][2][e] += v[r][e].z; bx[r][3][e] += v[r][e].w; }
        }
    }
    for (int rnd = plan.first; rnd < plan.first + plan.count; ++rnd) {
        // RHS = B + rho (X + U), every wave its own factor rows
#pragma unroll
        for (int r = 0; r < ITW; ++r) {
            const int it = wave + 4 * r;
            if (it < JT) {
                if (SPLIT) {                           // transposed: column NE x + e, factors 16 it + 4 q .. + 3 = chunk 4 it + q
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        float4 t4;
                        t4.x = bx[r][0][e] + rho * (hx[r][0][e] + ux[r][0][e]);
                        t4.y = bx[r][1][e] + rho * (hx[r][1][e] + ux[r][1][e]);
                        t4.z = bx[r][2][e] + rho * (hx[r][2][e] + ux[r][2][e]);
                        t4.w = bx[r][3][e] + rho * (hx[r][3][e] + ux[r][3][e]);
                        *reinterpret_cast<float4*>(lds + (NE * x + e) * KP + 4 * ((4 * it + q) ^ x)) = t4;
                    }
                } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float t[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) t[e] = bx[r][g][e] + rho * (hx[r][g][e] + ux[r][g][e]);
                    stv<NE>(lds + (16 * it + 4 * q + g) * CB + NE * x, t);
                }
                }
            }
        }
        __syncthreads();
        float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
        // CB = 32: the wave's B fragments of the whole RHS tile (KP x NE floats per lane) are read ONCE per
        // round and shared by its row tiles; read inside the MFMA loop they cost one LDS latency per k-step
        constexpr bool PRE = (CB == 32) && !SPLIT;
        float rball[PRE ? JT : 1][4][NE];
        AoFrag8 bh[SPLIT ? KS : 1][NE], bm[SPLIT ? KS : 1][NE], bl[SPLIT ? KS : 1][NE];
        if (SPLIT) {                                   // B operands of the whole tile, once per round: column NE x + e, k block q of k-step u
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const float* row = lds + (NE * x + e) * KP;
                    const float4 p0 = *reinterpret_cast<const float4*>(row + 4 * ((8 * u + 2 * q) ^ x));
                    const float4 p1 = *reinterpret_cast<const float4*>(row + 4 * ((8 * u + 2 * q + 1) ^ x));
                    ao_split8x3(p0, p1, bh[SPLIT ? u : 0][e], bm[SPLIT ? u : 0][e], bl[SPLIT ? u : 0][e]);
                }
        }
        if (PRE) {
#pragma unroll
            for (int u = 0; u < JT; ++u)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) ldv<NE>(rball[PRE ? u : 0][s2], lds + (16 * u + 4 * q + s2) * CB + NE * x);
        }
#pragma unroll
        for (int r = 0; r < ITW; ++r) {
            const int it = wave + 4 * r;
            if (it < JT) {
                f32x4 acc[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (SPLIT) {
#pragma unroll
                    for (int u = 0; u < KS; ++u) {     // A operand: M^-1 row 16 it + x, k block q of k-step u
                        const int off = (16 * it + x) * KP + 8 * ((4 * u + q) ^ swzm(x));
                        AoFrag8 ah, am, al;
                        ah.u = *reinterpret_cast<const uint4*>(mhi + off);
                        am.u = *reinterpret_cast<const uint4*>(mmd + off);
                        al.u = *reinterpret_cast<const uint4*>(mlo + off);
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[e] = AO_MFMA_BF16(ah, bh[SPLIT ? u : 0][e], acc[e]);
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[e] = AO_MFMA_BF16(am, bh[SPLIT ? u : 0][e], acc[e]);
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[e] = AO_MFMA_BF16(ah, bm[SPLIT ? u : 0][e], acc[e]);
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[e] = AO_MFMA_BF16(al, bh[SPLIT ? u : 0][e], acc[e]);
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[e] = AO_MFMA_BF16(ah, bl[SPLIT ? u : 0][e], acc[e]);
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[e] = AO_MFMA_BF16(am, bm[SPLIT ? u : 0][e], acc[e]);
                    }
                }
                float4 mfa[SPLIT ? 1 : JT];
#pragma unroll
                for (int u = 0; u < (SPLIT ? 0 : JT); ++u)
                    mfa[u] = *reinterpret_cast<const float4*>(ms + (16 * it + x) * LDM + 16 * u + 4 * q);
#pragma unroll
                for (int u = 0; u < (SPLIT ? 0 : JT); ++u) {
                    const float ma[4] = {mfa[u].x, mfa[u].y, mfa[u].z, mfa[u].w};
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        float rb[NE];
                        if (PRE) {
#pragma unroll
                            for (int e = 0; e < NE; ++e) rb[e] = rball[PRE ? u : 0][s2][e];
                        } else {
                            ldv<NE>(rb, lds + (16 * u + 4 * q + s2) * CB + NE * x);
                        }
#pragma unroll
                        for (int e = 0; e < NE; ++e) acc[e] = MFMA(ma[s2], rb[e], acc[e]);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const float ax = acc[e][g], ho = hx[r][g][e], uo = ux[r][g][e];
                        const float hn = prox_apply(ax, uo, shift);
                        const float un = uo + hn - ax;
                        const float d0 = hn - ax, d2 = hn - ho;
                        n0 += d0 * d0; n1 += hn * hn; n2 += d2 * d2; n3 += un * un;
                        hx[r][g][e] = hn; ux[r][g][e] = un;
                    }
                }
            }
        }
        if (plan.norms) block_store_norms<4>(n0, n1, n2, n3, nrm_rounds + ((int64_t)rnd * nblk + blockIdx.x) * 4, sh);
        __syncthreads();                               // the RHS tile is rewritten next round
    }
#pragma unroll
    for (int r = 0; r < ITW; ++r) {
        const int it = wave + 4 * r;
        if (it < JT) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t idx = (int64_t)(16 * it + 4 * q + g) * np + c0 + NE * x;
                stv<NE>(X + idx, hx[r][g]);
                stv<NE>(U + idx, ux[r][g]);
            }
        }
    }
    // r3: every launch that leaves an X behind also leaves its bf16 hi / lo images in both layouts -- the values, the split and
    // hence the bits of split_images_kernel, without its launch (config 3: 2 x 11 us per outer iteration)
    if (ihi) {
#pragma unroll
        for (int r = 0; r < ITW; ++r) {
            const int it = wave + 4 * r;
            if (it < JT) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {          // [factor][column]: NE consecutive columns per lane
                    const int64_t idx = (int64_t)(16 * it + 4 * q + g) * np + c0 + NE * x;
                    unsigned h[NE / 2], l[NE / 2];
#pragma unroll
                    for (int e = 0; e < NE / 2; ++e) ao_split2(hx[r][g][2 * e], hx[r][g][2 * e + 1], h[e], l[e]);
                    if (NE == 2) {
                        *reinterpret_cast<unsigned*>(ihi + idx) = h[0];
                        *reinterpret_cast<unsigned*>(ilo + idx) = l[0];
                    } else {
                        *reinterpret_cast<uint2*>(ihi + idx) = make_uint2(h[0], h[NE / 2 - 1]);
                        *reinterpret_cast<uint2*>(ilo + idx) = make_uint2(l[0], l[NE / 2 - 1]);
                    }
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {         // [column][factor]: the lane's four consecutive factors of one column
                    const int64_t idx = (c0 + NE * x + e) * KP + 16 * it + 4 * q;
                    unsigned h0, l0, h1, l1;
                    ao_split2(hx[r][0][e], hx[r][1][e], h0, l0);
                    ao_split2(hx[r][2][e], hx[r][3][e], h1, l1);
                    *reinterpret_cast<uint2*>(ithi + idx) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(itlo + idx) = make_uint2(l0, l1);
                }
            }
        }
    }
}

// W side: block = 64 rows (4 waves x 16), M^-1 in LDS; the right-hand side of a wave's 16 rows is
// turned from the accumulator layout to the A-operand layout through a wave-private LDS tile.
// RB = 64 rows per block (4 waves), or 128 (8 waves = two per SIMD, which overlap each other's serial parts;
// used when 128-row blocks still fill the CUs).
template <int KP, int RB>
__global__ __launch_bounds__(RB * 4) void ao_fused_rows_kernel(
    const float* __restrict__ Asum, int asplit, int64_t astride,     // right-hand side = sum of asplit slabs (slab order)
    float* __restrict__ X, float* __restrict__ U, float* __restrict__ Xb,
    float* __restrict__ Ub, const float* __restrict__ Minv, int prox, float lam, int admm_iter,
    DevState* __restrict__ st, double* __restrict__ nrm_rounds, int phase, int32_t* __restrict__ slot,
    const double* __restrict__ decide_tab,             // row-sharded runs: the ALL-REDUCED norm sums [admm_iter][4]
    const int* __restrict__ hint_rd, int* __restrict__ hint_wr,
    unsigned short* __restrict__ ihi = nullptr, unsigned short* __restrict__ ilo = nullptr,       // r3: the bf16 images of the new W
    unsigned short* __restrict__ ithi = nullptr, unsigned short* __restrict__ itlo = nullptr, int64_t mp = 0,   // ([mp][KP]), W^T ([KP][mp])
    const int* __restrict__ acnt = nullptr)            // r4, behind a stream-K product: slabs per 128-row block instead of asplit
{
    if (st->flag) return;
    if (acnt) asplit = acnt[((int64_t)blockIdx.x * RB) >> 7];
    constexpr int JT = KP / 16;
    constexpr int LDM = KP + 4;
    constexpr int LDR = KP + 4;
    // f32 form: M^-1 [KP][LDM] | RHS RB/16 x [16][LDR] | 32 doubles;  split form (KP >= 64): three bf16 images of M^-1
    // [3][KP][KP] | RHS RB/16 x [16][KP] | 32 doubles
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rs = lds + (KP >= 64 ? (3 * KP * KP) / 2 : KP * LDM);
    double* sh = reinterpret_cast<double*>(rs + RB * (KP >= 64 ? KP : LDR));
    const int nblk = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    FusedPlan plan;
    if (!fused_plan(phase, admm_iter, hint_rd, hint_wr, decide_tab ? decide_tab : nrm_rounds, decide_tab ? 1 : nblk, sh, st, slot, plan))
        return;
    const float rho = (float)st->rho;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;
    // k padded to 64 / 128: the product of a round, aux = rhs M^-1 (16 x KP x KP per wave), runs on the bf16 matrix cores with
    // BOTH operands split hi + lo and all four terms (f32-grade, like the V-sized products of ADMM / ANLS): 128 MFMAs of 16
    // cycles per wave and round at k = 128 instead of 256 f32 MFMAs of 32 -- the rounds were bound by the f32 MFMA rate of
    // the CU a block runs on.  M^-1 is staged ONCE as two bf16 images [KP][KP] (16-byte chunk c of row r at c ^ swz(r)), the
    // right-hand side tile of a wave [16][KP] f32 with chunk c of row r at c ^ r: conflict-free for the fragment reads
    // (tools/lab/swizzle_search.py's bank model).
    constexpr bool SPLIT = KP >= 64;
    constexpr int KS = KP / 32;                        // k-steps of 32
    unsigned short* mhi = reinterpret_cast<unsigned short*>(lds);
    unsigned short* mmd = mhi + KP * KP;
    unsigned short* mlo = mmd + KP * KP;
    auto swzm = [](int r) { return KP == 128 ? (r & 15) : ((r & 15) >> 1); };
    if (SPLIT) {
        for (int i = tid; i < KP * (KP / 8); i += RB * 4) {
            const int r = i / (KP / 8), c8 = i % (KP / 8);
            const float4 a = *reinterpret_cast<const float4*>(Minv + (int64_t)r * KP + 8 * c8);
            const float4 b = *reinterpret_cast<const float4*>(Minv + (int64_t)r * KP + 8 * c8 + 4);
            AoFrag8 h, md, l;
            ao_split8x3(a, b, h, md, l);
            const int pos = c8 ^ swzm(r);
            *reinterpret_cast<uint4*>(mhi + r * KP + 8 * pos) = h.u;
            *reinterpret_cast<uint4*>(mmd + r * KP + 8 * pos) = md.u;
            *reinterpret_cast<uint4*>(mlo + r * KP + 8 * pos) = l.u;
        }
    } else {
    for (int i = tid; i < KP * (KP / 4); i += RB * 4) {
        const int r = i / (KP / 4), c4 = i % (KP / 4);
        *reinterpret_cast<float4*>(lds + r * LDM + 4 * c4) = *reinterpret_cast<const float4*>(Minv + (int64_t)r * KP + 4 * c4);
    }
    }
    const int64_t r0 = (int64_t)blockIdx.x * RB + wave * 16;
    float* myrs = rs + wave * 16 * (KP >= 64 ? KP : LDR);
    // accumulator layout: [it][g] = row 4 q + g, column 16 it + x
    float wx[JT][4], dx[JT][4], ax0[JT][4];
    const float* srcX = plan.from_backup ? Xb : X;
    const float* srcU = plan.from_backup ? Ub : U;
#pragma unroll
    for (int it = 0; it < JT; ++it)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t idx = (r0 + 4 * q + g) * KP + 16 * it + x;
            wx[it][g] = srcX[idx]; dx[it][g] = srcU[idx];
            ax0[it][g] = Asum[idx];
            if (plan.save) { Xb[idx] = wx[it][g]; Ub[idx] = dx[it][g]; }
        }
    for (int p = 1; p < asplit; ++p) {                 // the other slabs, in slab order; a slab's loads all in flight together
        float av[JT][4];
#pragma unroll
        for (int it = 0; it < JT; ++it)
#pragma unroll
            for (int g = 0; g < 4; ++g) av[it][g] = Asum[(int64_t)p * astride + (r0 + 4 * q + g) * KP + 16 * it + x];
#pragma unroll
        for (int it = 0; it < JT; ++it)
#pragma unroll
            for (int g = 0; g < 4; ++g) ax0[it][g] += av[it][g];
    }
    __syncthreads();                                   // M^-1 is in place
    for (int rnd = plan.first; rnd < plan.first + plan.count; ++rnd) {
#pragma unroll
        for (int it = 0; it < JT; ++it)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v = ax0[it][g] + rho * (wx[it][g] + dx[it][g]);
                if (SPLIT) { const int row = 4 * q + g, col = 16 * it + x; myrs[row * KP + 4 * ((col >> 2) ^ row) + (col & 3)] = v; }
                else myrs[(4 * q + g) * LDR + 16 * it + x] = v;
            }
        __syncthreads();
        float4 xf[SPLIT ? 1 : JT];
        AoFrag8 ah[SPLIT ? KS : 1], am[SPLIT ? KS : 1], al[SPLIT ? KS : 1];
        if (SPLIT) {                                   // A operand: row x of the tile, k block q of k-step u (8 consecutive factors)
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const float4 p0 = *reinterpret_cast<const float4*>(myrs + x * KP + 4 * ((8 * u + 2 * q) ^ x));
                const float4 p1 = *reinterpret_cast<const float4*>(myrs + x * KP + 4 * ((8 * u + 2 * q + 1) ^ x));
                ao_split8x3(p0, p1, ah[u], am[u], al[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < JT; ++u) xf[u] = *reinterpret_cast<const float4*>(myrs + x * LDR + 16 * u + 4 * q);
        }
        float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
        for (int it = 0; it < JT; ++it) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (SPLIT) {
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};     // two chains: consecutive MFMAs never wait for each other
#pragma unroll
                for (int u = 0; u < KS; ++u) {         // B operand: M^-1 row 16 it + x (= column, by symmetry of the convention), k block q
                    const int off = (16 * it + x) * KP + 8 * ((4 * u + q) ^ swzm(x));
                    AoFrag8 bh, bm, bl;
                    bh.u = *reinterpret_cast<const uint4*>(mhi + off);
                    bm.u = *reinterpret_cast<const uint4*>(mmd + off);
                    bl.u = *reinterpret_cast<const uint4*>(mlo + off);
                    acc = AO_MFMA_BF16(ah[u], bh, acc);
                    acc2 = AO_MFMA_BF16(am[u], bh, acc2);
                    acc = AO_MFMA_BF16(ah[u], bm, acc);
                    acc2 = AO_MFMA_BF16(al[u], bh, acc2);
                    acc = AO_MFMA_BF16(ah[u], bl, acc);
                    acc2 = AO_MFMA_BF16(am[u], bm, acc2);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] += acc2[g];
            } else {
#pragma unroll
            for (int u = 0; u < JT; ++u) {
                const float4 mb = *reinterpret_cast<const float4*>(lds + (16 * it + x) * LDM + 16 * u + 4 * q);
                acc = MFMA(xf[u].x, mb.x, acc);
                acc = MFMA(xf[u].y, mb.y, acc);
                acc = MFMA(xf[u].z, mb.z, acc);
                acc = MFMA(xf[u].w, mb.w, acc);
            }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float w = wx[it][g], d = dx[it][g], ax = acc[g];
                const float wn = prox_apply(ax, d, shift);
                const float dn = d + wn - ax;
                const float d0 = wn - ax, d2 = wn - w;
                n0 += d0 * d0; n1 += wn * wn; n2 += d2 * d2; n3 += dn * dn;
                wx[it][g] = wn; dx[it][g] = dn;
            }
        }
        if (plan.norms) block_store_norms<RB / 16>(n0, n1, n2, n3, nrm_rounds + ((int64_t)rnd * nblk + blockIdx.x) * 4, sh);
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < JT; ++it)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t idx = (r0 + 4 * q + g) * KP + 16 * it + x;
            X[idx] = wx[it][g]; U[idx] = dx[it][g];
        }
    // r3: the bf16 hi / lo images of the new W in both layouts (see ao_fused_cols_kernel).  [factor][row]: the lane's four
    // consecutive rows of one factor; [row][factor]: the tile is turned through the wave's LDS tile once more, so that a lane
    // stores eight consecutive factors (16 bytes per image).
    if (SPLIT && ihi) {
#pragma unroll
        for (int it = 0; it < JT; ++it) {
            unsigned h0, l0, h1, l1;
            ao_split2(wx[it][0], wx[it][1], h0, l0);
            ao_split2(wx[it][2], wx[it][3], h1, l1);
            const int64_t idx = (int64_t)(16 * it + x) * mp + r0 + 4 * q;
            *reinterpret_cast<uint2*>(ithi + idx) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(itlo + idx) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int it = 0; it < JT; ++it)
#pragma unroll
            for (int g = 0; g < 4; ++g) { const int row = 4 * q + g, col = 16 * it + x; myrs[row * KP + 4 * ((col >> 2) ^ row) + (col & 3)] = wx[it][g]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const float4 p0 = *reinterpret_cast<const float4*>(myrs + x * KP + 4 * ((8 * u + 2 * q) ^ x));
            const float4 p1 = *reinterpret_cast<const float4*>(myrs + x * KP + 4 * ((8 * u + 2 * q + 1) ^ x));
            AoFrag8 h, l;
            ao_split8(p0, p1, h, l);
            const int64_t idx = (r0 + x) * KP + 32 * u + 8 * q;
            *reinterpret_cast<uint4*>(ihi + idx) = h.u;
            *reinterpret_cast<uint4*>(ilo + idx) = l.u;
        }
    }
}

// Row-sharded runs: this rank's four norm sums of round `round` -> out[0..3] (fixed order), to be
// all-reduced by the caller before the next round looks at them.
__global__ __launch_bounds__(256) void ao_norm_gather_kernel(
    const DevState* __restrict__ st, const double* __restrict__ nrm, int nblk, int round, double* __restrict__ out)
{
    if (st->flag) return;
    const double* part = nrm + (int64_t)(round & 1) * nblk * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    // rounds that did not run (inner stop already set) leave stale partials: contribute their
    // value anyway, every rank skips the same rounds and nobody reads the sums afterwards
    for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * 4 + wave];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[wave] = s;
}

// ---- host sequencing -----------------------------------------------------
template <typename T>
static int lazy_alloc(nmfx_engine* E, T** p, int64_t count) {
    if (*p) return NMFX_OK;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T)));
    NMFX_HIP(hipMemsetAsync(*p, 0, (size_t)count * sizeof(T), E->stream));
    return NMFX_OK;
}


int nmfx_kl_state_alloc(nmfx_engine* E) {      // m x n auxiliaries of the KL-loss ADMM variants
    int rc;
    if ((rc = lazy_alloc(E, &E->S, E->mp * E->np))) return rc;
    if ((rc = lazy_alloc(E, &E->DV, E->mp * E->np))) return rc;
    if ((rc = lazy_alloc(E, &E->auxH, (int64_t)E->kp * E->np))) return rc;
    if ((rc = lazy_alloc(E, &E->Asum, E->mp * E->kp))) return rc;
    return NMFX_OK;
}

int nmfx_aoadmm_alloc(nmfx_engine* E) {
    int rc;
    if ((rc = lazy_alloc(E, &E->dualW, E->mp * E->kp))) return rc;
    if ((rc = lazy_alloc(E, &E->dualH, (int64_t)E->kp * E->np))) return rc;
    if ((rc = lazy_alloc(E, &E->auxW, E->mp * E->kp))) return rc;      // A = V H^T summed over splits
    if ((rc = lazy_alloc(E, &E->Minv, (int64_t)E->kp * E->kp))) return rc;
    if ((rc = lazy_alloc(E, &E->nrm_part, 2 * std::max(E->mp, E->np) / 64 * 4 + 64))) return rc;
    return NMFX_OK;
}

template <int KP>
static int launch_prepare(nmfx_engine* E, const float* src, int record_obj, int64_t j, int64_t min_iter,
                          double tol1, double tol2, double fixed_rho) {
    static const bool scalar = getenv("NMFX_PREPARE_SCALAR") != nullptr;      // the one-barrier-per-pivot kernel
    if constexpr (KP >= 64) {
        if (!scalar) {
            constexpr int NB = KP / 16;
            constexpr size_t shm = (size_t)(2 * 16 * 17 + 3 * 16 * (KP + 2) + NB * 16 * 17 + 4) * sizeof(double);
            { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(ao_prepare_mfma_kernel<KP>), (int)shm); if (rc_) return rc_; }
            hipLaunchKernelGGL((ao_prepare_mfma_kernel<KP>), dim3(1), dim3(KP * 4 + 64), shm, E->stream, src, E->k, E->Minv,
                               E->state, record_obj, E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                               E->obj_hist, fixed_rho);
            NMFX_HIP(hipGetLastError());
            return NMFX_OK;
        }
    }
    constexpr int NT = (KP / 4) * (KP / 4) < 64 ? 64 : (KP / 4) * (KP / 4);
    hipLaunchKernelGGL((ao_prepare_kernel<KP>), dim3(1), dim3(NT), 0, E->stream, src, E->k, E->Minv, E->state,
                       record_obj, E->xf64, (long long)j, (long long)min_iter, tol1, tol2, E->obj_hist, fixed_rho);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// f64 inverse of the 128 x 128 f64 block at src64 (row stride ld) -> out64 [128][128]: a diagonal block of the blocked Gauss-Jordan
// inversion beyond k = 128 (kernels_generic.hip: gx_prepare); a pivot <= 0 sets the run's notpd / flag like the k <= 128 prepare
int nmfx_launch_inverse64_block(nmfx_engine* E, const double* src64, int64_t ld, double* out64) {
    constexpr int KP = 128, NB = KP / 16;
    constexpr size_t shm = (size_t)(2 * 16 * 17 + 3 * 16 * (KP + 2) + NB * 16 * 17 + 4) * sizeof(double);
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(ao_prepare_mfma_kernel<KP>), (int)shm); if (rc_) return rc_; }
    hipLaunchKernelGGL((ao_prepare_mfma_kernel<KP>), dim3(1), dim3(KP * 4 + 64), shm, E->stream, (const float*)nullptr, KP, (float*)nullptr,
                       E->state, 0, (const double*)nullptr, 0ll, 0ll, 0.0, 0.0, (double*)nullptr, 0.0, out64, (int*)nullptr, src64, ld);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// f64 inverse of src + diag_add I (k >= 64: the blocked f64-MFMA Gauss-Jordan above) for the ANLS complement solver
int nmfx_launch_inverse64(nmfx_engine* E, const float* src, double diag_add, double* out64, int* soft_bad) {
    auto go = [&](auto kp_t) -> int {
        constexpr int KP = decltype(kp_t)::value;
        constexpr int NB = KP / 16;
        constexpr size_t shm = (size_t)(2 * 16 * 17 + 3 * 16 * (KP + 2) + NB * 16 * 17 + 4) * sizeof(double);
        { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(ao_prepare_mfma_kernel<KP>), (int)shm); if (rc_) return rc_; }
        hipLaunchKernelGGL((ao_prepare_mfma_kernel<KP>), dim3(1), dim3(KP * 4 + 64), shm, E->stream, src, E->k, (float*)nullptr,
                           E->state, 0, (const double*)nullptr, 0ll, 0ll, 0.0, 0.0, (double*)nullptr, diag_add, out64, soft_bad);
        NMFX_HIP(hipGetLastError());
        return NMFX_OK;
    };
    if (E->kp == 64) return go(std::integral_constant<int, 64>{});
    if (E->kp == 128) return go(std::integral_constant<int, 128>{});
    E->err = "inverse64: k padded to 64 or 128 only"; return NMFX_E_ARG;
}

int nmfx_launch_prepare(nmfx_engine* E, const float* src, int record_obj, int64_t j, int64_t min_iter,
                        double tol1, double tol2, double fixed_rho) {
    ProfScope ps(E, "prepare");
    switch (E->kp) {
        case 16: return launch_prepare<16>(E, src, record_obj, j, min_iter, tol1, tol2, fixed_rho);
        case 32: return launch_prepare<32>(E, src, record_obj, j, min_iter, tol1, tol2, fixed_rho);
        case 64: return launch_prepare<64>(E, src, record_obj, j, min_iter, tol1, tol2, fixed_rho);
        default: return launch_prepare<128>(E, src, record_obj, j, min_iter, tol1, tol2, fixed_rho);
    }
}

template <int KP>
static int launch_inner_cols(nmfx_engine* E, const float* M, float* aux, int mode, int prox, float lam, int round) {
    const size_t shm = (size_t)KP * 64 * sizeof(float) + 16 * sizeof(double);
    hipLaunchKernelGGL((ao_inner_cols_kernel<KP>), dim3((unsigned)(E->np / 64)), dim3(256), shm, E->stream,
                       E->xf32, E->H, E->dualH, M, aux, mode, E->np, prox, lam, round, E->state, E->nrm_part);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int KP>
static int launch_inner_rows(nmfx_engine* E, const float* Asum, float* W, const float* M, float* aux, int mode,
                             int prox, float lam, int round, const double* nrm_global) {
    const size_t shm = (size_t)KP * (KP + 4) * sizeof(float) + 16 * sizeof(double);
    auto kern = ao_inner_rows_kernel<KP>;
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(E->mp / 64)), dim3(256), shm, E->stream, Asum, W, E->dualW,
                       M, aux, mode, prox, lam, round, E->state, E->nrm_part, nrm_global);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_inner_cols(nmfx_engine* E, const float* M, float* aux, int mode, int prox, float lam, int round) {
    switch (E->kp) {
        case 16: return launch_inner_cols<16>(E, M, aux, mode, prox, lam, round);
        case 32: return launch_inner_cols<32>(E, M, aux, mode, prox, lam, round);
        case 64: return launch_inner_cols<64>(E, M, aux, mode, prox, lam, round);
        default: return launch_inner_cols<128>(E, M, aux, mode, prox, lam, round);
    }
}

int nmfx_inner_rows(nmfx_engine* E, const float* Asum, float* W, const float* M, float* aux, int mode, int prox,
                    float lam, int round, const double* nrm_global) {
    switch (E->kp) {
        case 16: return launch_inner_rows<16>(E, Asum, W, M, aux, mode, prox, lam, round, nrm_global);
        case 32: return launch_inner_rows<32>(E, Asum, W, M, aux, mode, prox, lam, round, nrm_global);
        case 64: return launch_inner_rows<64>(E, Asum, W, M, aux, mode, prox, lam, round, nrm_global);
        default: return launch_inner_rows<128>(E, Asum, W, M, aux, mode, prox, lam, round, nrm_global);
    }
}

static int inner_cols(nmfx_engine* E, int prox, float lam, int round) {
    return nmfx_inner_cols(E, E->Minv, nullptr, 0, prox, lam, round);
}

static int inner_rows(nmfx_engine* E, float* W, int prox, float lam, int round, const double* nrm_global = nullptr) {
    return nmfx_inner_rows(E, E->auxW, W, E->Minv, nullptr, 0, prox, lam, round, nrm_global);
}

// one round of a sub-problem at any padded rank up to 512 (see ao_round_*_any_kernel): X = H (cols) or W[0] (rows), its dual, B = the
// summed right-hand side product ([kp][np] / [mp][kp]); nmfx_inner_finish(E, np / 64 | mp / 64, ...) closes the sub-problem
int nmfx_round_any(nmfx_engine* E, bool cols, const float* B, float* X, float* U, int prox, float lam, int round, const double* nrm_global) {
    if (E->kp > 512 || E->kp % 64) { E->err = "round_any: k padded to at most 512"; return NMFX_E_ARG; }
    int rc;
    static const bool f32_rounds = getenv("NMFX_GX_ROUNDS_F32") != nullptr;
    if (E->precision == 1 && !f32_rounds && E->kp % 32 == 0 && E->kp <= 384) {      // r4: the product on the bf16 matrix cores, three images, six terms
        const int64_t kk = (int64_t)E->kp * E->kp;
        if (!E->minv_img) NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->minv_img), (size_t)3 * kk * sizeof(unsigned short)));
        if (round == 0) {                              // (the images of this sub-problem's M^-1: every sub-problem starts at round 0)
            hipLaunchKernelGGL(ao_minv_images_kernel, dim3((unsigned)((kk / 8 + 255) / 256)), dim3(256), 0, E->stream, (const float*)E->Minv, (int)E->kp,
                               E->minv_img, (const DevState*)E->state);
            NMFX_HIP(hipGetLastError());
        }
        const size_t shm = (size_t)3 * 64 * (E->kp + 8) * sizeof(unsigned short) + 32 * sizeof(double);
        constexpr int NWV = 8;                         // (two waves per SIMD: with four, a block's phases -- tile build, products, scattered X / U traffic -- ran one after the other)
        if (cols) {
            if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(ao_round_any_bf16_kernel<true, NWV>), (int)shm))) return rc;
            hipLaunchKernelGGL((ao_round_any_bf16_kernel<true, NWV>), dim3((unsigned)(E->np / 64)), dim3(64 * NWV), shm, E->stream, B, X, U, (const unsigned short*)E->minv_img,
                               (int)E->kp, E->np, prox, lam, round, E->state, E->nrm_part, (const double*)nullptr);
        } else {
            if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(ao_round_any_bf16_kernel<false, NWV>), (int)shm))) return rc;
            hipLaunchKernelGGL((ao_round_any_bf16_kernel<false, NWV>), dim3((unsigned)(E->mp / 64)), dim3(64 * NWV), shm, E->stream, B, X, U, (const unsigned short*)E->minv_img,
                               (int)E->kp, E->kp, prox, lam, round, E->state, E->nrm_part, nrm_global);
        }
        NMFX_HIP(hipGetLastError());
        return NMFX_OK;
    }
    if (cols) {
        const size_t shm = (size_t)E->kp * 64 * sizeof(float) + 16 * sizeof(double);
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(ao_round_cols_any_kernel), (int)shm))) return rc;
        hipLaunchKernelGGL(ao_round_cols_any_kernel, dim3((unsigned)(E->np / 64)), dim3(256), shm, E->stream, B, X, U, (const float*)E->Minv, E->kp,
                           E->np, prox, lam, round, E->state, E->nrm_part);
    } else {
        const size_t shm = (size_t)64 * (E->kp + 4) * sizeof(float) + 16 * sizeof(double);
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(ao_round_rows_any_kernel), (int)shm))) return rc;
        hipLaunchKernelGGL(ao_round_rows_any_kernel, dim3((unsigned)(E->mp / 64)), dim3(256), shm, E->stream, B, X, U, (const float*)E->Minv, E->kp,
                           prox, lam, round, E->state, E->nrm_part, nrm_global);
    }
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// this rank's four norm sums of round `round` of the W sub-problem -> xf64[1..4] (row-sharded runs all-reduce them)
int nmfx_gather_round_norms(nmfx_engine* E, int nblk, int round) {
    hipLaunchKernelGGL(ao_norm_gather_kernel, dim3(1), dim3(256), 0, E->stream, E->state, E->nrm_part, nblk, round, E->xf64 + 1);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_inner_finish(nmfx_engine* E, int nblk, int admm_iter, int32_t* slot, const double* nrm_global) {
    hipLaunchKernelGGL(ao_inner_finish_kernel, dim3(1), dim3(256), 0, E->stream, E->state, E->nrm_part, nblk,
                       admm_iter, slot, nrm_global);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// One outer iteration of ao_admm.py:259-292 (Euclidean loss), in the pieces the row-sharded
// form needs (nmfx_aoadmm_phase_*): between h_products and h_solve the caller all-reduces
// [W^T V | W^T W | objective]; between the rounds of the W sub-problem, the four norm sums.
// Split-bf16 products (kp = 64 / 128, Euclidean loss): the H-side product also carries the residual
// objective of the CURRENT pair (X = V^T, Y = W^T images, Z = H^T images), which is exactly the
// obj[j] the following `prepare` records -- so the separate objective pass over V at the end of
// every outer iteration (one third of the V traffic) disappears.
static bool ao_bf16(const nmfx_engine* E) { return E->precision == 1 && nmfx_bf16_supported(E); }

static int ao_bf16_objective_product(nmfx_engine* E) {   // Bt_part, obj_part (and G_part for kp = 64) of the current pair
    int rc;
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if (!E->wimg_ok && (rc = nmfx_bf16_images_w(E, E->W[0], 0))) return rc;       // (r3: the fused W-side launches leave them)
    E->wimg_ok = true;
    if (!E->himg_both && (rc = nmfx_bf16_images_h(E, true))) return rc;   // (the W side of the previous iteration built both)
    return nmfx_bf16_vtw(E, true, "hphase", false, 3);
}

// r4, single GPU, k padded to 128, split bf16, fused rounds: the two one-workgroup inversions run as SIDE JOBS of the V-sized
// products (stream-K partition over ncu - 1 workers, kernels_bf16.hip) instead of as launches of their own, the Gram slabs are summed
// by the side job, the objective is recorded by the pack.  NMFX_AO_OVERLAP=0 keeps the launches of round 3.
static bool ao_overlap(const nmfx_engine* E, int admm_iter) {
    static const bool on = !(getenv("NMFX_AO_OVERLAP") && atoi(getenv("NMFX_AO_OVERLAP")) == 0);
    return on && ao_bf16(E) && nmfx_sk_enabled(E) && admm_iter >= 2 && E->kp >= 16;
}

static int ao_h_products(nmfx_engine* E) {
    int rc;
    float* W = E->W[0];
    if (ao_bf16(E)) {
        if ((rc = ao_bf16_objective_product(E))) return rc;
        const int64_t nobj = (int64_t)(E->np / 128) * E->bt_split;
        if (E->kp == 64) return nmfx_bf16_pack_t(E, E->G_part, nmfx_bf16_g_slabs(E), nobj);     // W^T W: by-product slabs
        int gslabs = 64;                               // W^T W from the transposed images ao_bf16_objective_product has just built (up to 64 slabs: the pack sums them, no fold launch)
        if ((rc = nmfx_bf16_gram_tn(E, &gslabs))) return rc;
        return nmfx_bf16_pack_t(E, E->G_part, gslabs, nobj);
    }
    const bool fuse_g = nmfx_hphase_can_fuse_gram(E);
    if (!fuse_g && (rc = nmfx_launch_gram_tn(E, W, E->mp, E->G_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_hphase(E, W, fuse_g))) return rc;
    return nmfx_launch_pack(E);                                      // xf32 = [W^T V | W^T W], xf64 = obj[j]
}

// objective partials of the pair the iteration has just produced (f32 path: a pass over V now;
// bf16 path: nothing, the next H-side product or ao_final_objective computes it)
static int ao_new_pair_objective(nmfx_engine* E) {
    E->lazy_objective = ao_bf16(E);
    if (E->lazy_objective) return NMFX_OK;
    return nmfx_launch_wphase(E, E->W[0], false, true);
}

// xf64[0] = objective partial of the current pair (end of a run)
static int ao_final_objective(nmfx_engine* E) {
    int rc;
    if (E->lazy_objective) {
        if ((rc = ao_bf16_objective_product(E))) return rc;
        return nmfx_launch_obj_reduce(E, (int64_t)(E->np / 128) * E->bt_split);
    }
    return nmfx_launch_obj_reduce(E);
}

// All rounds of a sub-problem in three launches (see ao_fused_cols_kernel).  NMFX_AO_FUSED=0 keeps
// the round-by-round kernels (they remain the path of the row-sharded W sub-problem, whose norms
// cross ranks every round).
static bool ao_fused_enabled(const nmfx_engine* E, int admm_iter) {
    static const bool on = !(getenv("NMFX_AO_FUSED") && atoi(getenv("NMFX_AO_FUSED")) == 0);
    return on && admm_iter >= 2 && E->kp >= 16;
}

static int ao_fused_alloc(nmfx_engine* E, int admm_iter) {
    int rc;
    const int64_t big = std::max(E->mp, E->np) * E->kp;
    if ((rc = lazy_alloc(E, &E->bkX, big))) return rc;
    if ((rc = lazy_alloc(E, &E->bkU, big))) return rc;
    const int64_t need = (int64_t)admm_iter * std::max(E->mp / 64, E->np / 32) * 4 + 64;
    if (need > E->nrm_rounds_cap) {
        if (E->nrm_rounds) { NMFX_HIP(hipStreamSynchronize(E->stream)); hipFree(E->nrm_rounds); E->nrm_rounds = nullptr; }
        NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->nrm_rounds), (size_t)need * sizeof(double)));
        NMFX_HIP(hipMemsetAsync(E->nrm_rounds, 0, (size_t)need * sizeof(double), E->stream));
        E->nrm_rounds_cap = need;
    }
    return NMFX_OK;
}

// columns per block of the fused H-side kernel: 32 when 64-column blocks would not fill the CUs
static int ao_fused_cols_cb(const nmfx_engine* E) { return (E->np / 64 < (int64_t)E->ncu) ? 32 : 64; }

template <int KP, int CB>
static int launch_fused_cols_cb(nmfx_engine* E, int prox, float lam, int admm_iter, int phase, int32_t* slot,
                                const int* hint_rd, int* hint_wr) {
    constexpr bool split = KP >= 64 && CB == 32;       // three bf16 images of M^-1 instead of the f32 copy
    const size_t shm = (size_t)(KP * CB) * sizeof(float) + (split ? (size_t)3 * KP * KP * 2 : (size_t)KP * (KP + 4) * sizeof(float)) +
                       64 * sizeof(double);
    auto kern = ao_fused_cols_kernel<KP, CB>;
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc_) return rc_; }
    const bool img = E->ao_images;                     // (ao_fused_subproblem: split-bf16 run, images allocated)
    const bool sk = E->ao_b_src != nullptr;            // (the overlap iteration: no pack launch)
    hipLaunchKernelGGL(kern, dim3((unsigned)(E->np / CB)), dim3(256), shm, E->stream, E->xf32, E->H, E->dualH, E->bkX, E->bkU,
                       E->Minv, E->np, prox, lam, admm_iter, E->state, E->nrm_rounds, phase, slot, hint_rd, hint_wr,
                       img ? E->Hhi : nullptr, img ? E->Hlo : nullptr, img ? E->HThi : nullptr, img ? E->HTlo : nullptr,
                       E->ao_b_src, E->ao_b_cnt, (int64_t)E->np * E->kp, sk ? E->obj_part : (const double*)nullptr, E->ao_rec_nobj,
                       E->obj_hist, E->xf64, (long long)E->ao_rec_j, (long long)E->ao_rec_min_iter, E->ao_rec_tol1, E->ao_rec_tol2);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
template <int KP>
static int launch_fused_cols(nmfx_engine* E, int prox, float lam, int admm_iter, int phase, int32_t* slot,
                             const int* hint_rd, int* hint_wr) {
    return ao_fused_cols_cb(E) == 32 ? launch_fused_cols_cb<KP, 32>(E, prox, lam, admm_iter, phase, slot, hint_rd, hint_wr)
                                     : launch_fused_cols_cb<KP, 64>(E, prox, lam, admm_iter, phase, slot, hint_rd, hint_wr);
}

// rows per block of the fused W-side kernel (also the granularity of its norm partials)
static int ao_fused_rows_rb(const nmfx_engine* E) {
    static const int forced = getenv("NMFX_AO_ROWS_RB") ? atoi(getenv("NMFX_AO_ROWS_RB")) : 0;
    if (forced == 64 || (forced == 128 && E->mp % 128 == 0 && E->kp == 64)) return forced;
    // (k padded to 128: three images of M^-1 + the tiles of eight waves would need all 160 KiB and 512 bytes more)
    return (E->kp == 64 && E->mp % 128 == 0 && E->mp / 128 >= (int64_t)E->ncu) ? 128 : 64;
}

template <int KP, int RB>
static int launch_fused_rows_rb(nmfx_engine* E, float* W, int prox, float lam, int admm_iter, int phase, int32_t* slot,
                               const double* decide_tab, const int* hint_rd, int* hint_wr) {
    const size_t shm = (KP >= 64 ? (size_t)3 * KP * KP * 2 + (size_t)RB * KP * sizeof(float)
                                 : (size_t)(KP * (KP + 4) + RB * (KP + 4)) * sizeof(float)) + 64 * sizeof(double);
    auto kern = ao_fused_rows_kernel<KP, RB>;
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc_) return rc_; }
    // ao_a_slabs > 0: the W-side product's slabs are added here instead of by a sum_partials launch
    const bool slabs = E->ao_a_slabs > 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)(E->mp / RB)), dim3(RB * 4), shm, E->stream,
                       slabs ? (E->ao_a_src ? E->ao_a_src : E->A_part) : E->auxW,
                       slabs ? E->ao_a_slabs : 1, (int64_t)E->mp * E->kp, W, E->dualW, E->bkX, E->bkU,
                       E->Minv, prox, lam, admm_iter, E->state, E->nrm_rounds, phase, slot, decide_tab, hint_rd, hint_wr,
                       E->ao_images ? E->Whi[0] : nullptr, E->ao_images ? E->Wlo[0] : nullptr, E->ao_images ? E->WThi : nullptr,
                       E->ao_images ? E->WTlo : nullptr, E->mp, slabs ? E->ao_a_cnt : (const int*)nullptr);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
template <int KP>
static int launch_fused_rows(nmfx_engine* E, float* W, int prox, float lam, int admm_iter, int phase, int32_t* slot,
                             const double* decide_tab, const int* hint_rd, int* hint_wr) {
    if constexpr (KP == 64) {
        if (ao_fused_rows_rb(E) == 128)
            return launch_fused_rows_rb<KP, 128>(E, W, prox, lam, admm_iter, phase, slot, decide_tab, hint_rd, hint_wr);
    }
    return launch_fused_rows_rb<KP, 64>(E, W, prox, lam, admm_iter, phase, slot, decide_tab, hint_rd, hint_wr);
}
static int fused_rows_any(nmfx_engine* E, float* W, int prox, float lam, int admm_iter, int phase, int32_t* slot,
                          const double* decide_tab = nullptr, const int* hint_rd = nullptr, int* hint_wr = nullptr) {
    switch (E->kp) {
        case 16: return launch_fused_rows<16>(E, W, prox, lam, admm_iter, phase, slot, decide_tab, hint_rd, hint_wr);
        case 32: return launch_fused_rows<32>(E, W, prox, lam, admm_iter, phase, slot, decide_tab, hint_rd, hint_wr);
        case 64: return launch_fused_rows<64>(E, W, prox, lam, admm_iter, phase, slot, decide_tab, hint_rd, hint_wr);
        default: return launch_fused_rows<128>(E, W, prox, lam, admm_iter, phase, slot, decide_tab, hint_rd, hint_wr);
    }
}

// this rank's norm sums of every speculative round, nrm_rounds[round][block][4] -> out[round][4] (fixed order)
__global__ __launch_bounds__(256) void nrm_table_kernel(const double* __restrict__ nrm_rounds, int nblk, int admm_iter,
                                                        double* __restrict__ out, const int* __restrict__ flag)
{
    if (*flag) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;       // wave = component
    for (int r = 0; r < admm_iter; ++r) {
        double s = 0.0;
        for (int b = lane; b < nblk; b += 64) s += nrm_rounds[((int64_t)r * nblk + b) * 4 + wave];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) out[r * 4 + wave] = s;
    }
}

// Two launches without the hint (speculate all rounds, decide + repair), three with it (see fused_plan; NMFX_AO_HINT=0
// turns it off).  `parity` alternates between consecutive sub-problems of the same side: the hint slot read / written.
static int ao_fused_subproblem(nmfx_engine* E, bool cols, float* W, int prox, float lam, int admm_iter, int32_t* slot, int parity) {
    static const bool hinted = !(getenv("NMFX_AO_HINT") && atoi(getenv("NMFX_AO_HINT")) == 0);
    int rc;
    if ((rc = ao_fused_alloc(E, admm_iter))) return rc;
    // r3: in a split-bf16 run the launches below leave the bf16 images of the factor they update (NMFX_AO_IMG=0: the separate
    // `images` launches of round 2; a sub-problem of zero rounds has no launch that writes anything)
    static const bool img_on = !(getenv("NMFX_AO_IMG") && atoi(getenv("NMFX_AO_IMG")) == 0);
    E->ao_images = img_on && ao_bf16(E) && admm_iter > 0 && W == (cols ? nullptr : E->W[0]) && E->Whi[0] && E->Hhi;
    if (E->ao_images && cols) {
        if ((rc = lazy_alloc(E, &E->HThi, (int64_t)E->kp * E->np))) return rc;
        if ((rc = lazy_alloc(E, &E->HTlo, (int64_t)E->kp * E->np))) return rc;
    }
    const int* hint_rd = hinted ? &E->state->ao_hint[cols ? 0 : 1][parity & 1] : nullptr;
    int* hint_wr = hinted ? &E->state->ao_hint[cols ? 0 : 1][(parity & 1) ^ 1] : nullptr;
    for (int phase = 0; phase < (hinted ? 3 : 2); ++phase) {
        if (cols) {
            switch (E->kp) {
                case 16: rc = launch_fused_cols<16>(E, prox, lam, admm_iter, phase, slot, hint_rd, hint_wr); break;
                case 32: rc = launch_fused_cols<32>(E, prox, lam, admm_iter, phase, slot, hint_rd, hint_wr); break;
                case 64: rc = launch_fused_cols<64>(E, prox, lam, admm_iter, phase, slot, hint_rd, hint_wr); break;
                default: rc = launch_fused_cols<128>(E, prox, lam, admm_iter, phase, slot, hint_rd, hint_wr); break;
            }
        } else {
            rc = fused_rows_any(E, W, prox, lam, admm_iter, phase, slot, nullptr, hint_rd, hint_wr);
        }
        if (rc) { E->ao_images = false; return rc; }
    }
    if (E->ao_images) { if (cols) E->himg_both = true; else E->wimg_ok = true; }
    E->ao_images = false;                              // (the row-sharded entry points launch the same kernels without images)
    return NMFX_OK;
}

// ---- prox 'l1inf' / 'l1inf_transpose' inside AO-ADMM (nmf/ao_admm.py:143-195 = nmf/admm.py:158-210 word for word; r4) ----
// The operator couples whole rows / columns of the factor, so a round is: aux = M^-1 (B + rho (X + U)) (the round kernel's
// "solve only" mode), X_prev = X, X = prox(aux, U) and U += X - aux (kernels_prox.hip, with the sub-problem's rho read on the
// device), then the four sums of `terminate` (ao_admm.py:33-43) from (X, X_prev, aux, U) in the slots the next round's kernel sums.
// In the reference these runs end within a few outer iterations in scipy's LinAlgError (the operator wipes a factor out, its Gram
// matrix is zero): here the same pivot test stops the run with NMFX_E_NOTPD at the same place.
__global__ __launch_bounds__(256) void ao_l1inf_norms_kernel(
    const float* __restrict__ X, const float* __restrict__ Xprev, const float* __restrict__ AUX, const float* __restrict__ U,
    int64_t count, int round, const DevState* __restrict__ st, double* __restrict__ nrm)     // nrm: [2][gridDim.x][4]
{
    if (st->flag || st->inner_stop) return;
    __shared__ double sh[16];
    const int64_t per = (count + gridDim.x - 1) / gridDim.x, i0 = per * blockIdx.x, i1 = i0 + per < count ? i0 + per : count;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float x = X[i], d0 = x - AUX[i], d2 = x - Xprev[i], u = U[i];
        n0 += d0 * d0; n1 += x * x; n2 += d2 * d2; n3 += u * u;
    }
    block_store_norms<4>(n0, n1, n2, n3, nrm + ((int64_t)(round & 1) * gridDim.x + blockIdx.x) * 4, sh);
}

// One inner round with prox 'l1inf' / 'l1inf_transpose' (nmf/ao_admm.py:143-195): the "solve only" mode of the round kernel (aux), the
// operator on (aux, dual) -> X, dual, and the four norm sums of the round -- for either loss: the right-hand side (xf32 / Asum) is
// whatever the caller's products left (least squares: W^T V, once per sub-problem; KL: W^T (v_aux + dual_v), every round)
static int ao_l1inf_round(nmfx_engine* E, bool cols, int prox, double lam, int r) {
    int rc;
    const int nblk = (int)((cols ? E->np : E->mp) / 64);
    const int64_t count = cols ? (int64_t)E->kp * E->np : E->mp * (int64_t)E->kp;
    float* X = cols ? E->H : E->W[0];
    float* U = cols ? E->dualH : E->dualW;
    float* aux = cols ? E->auxH : E->auxW;
    if (cols) rc = nmfx_inner_cols(E, E->Minv, E->auxH, 1, prox, (float)lam, r);
    else rc = nmfx_inner_rows(E, E->Asum, E->W[0], E->Minv, E->auxW, 1, prox, (float)lam, r);
    if (rc) return rc;
    NMFX_HIP(hipMemcpyAsync(E->bkX, X, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, E->stream));
    if ((rc = nmfx_launch_prox_l1inf(E, cols, prox == NMFX_PROX_L1INF_T, -1.0, lam, 1.0, true, true))) return rc;
    hipLaunchKernelGGL(ao_l1inf_norms_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, X, E->bkX, aux, U, count, r,
                       E->state, E->nrm_part);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

static int ao_l1inf_subproblem(nmfx_engine* E, bool cols, int prox, double lam, int admm_iter, int32_t* slot) {
    int rc;
    if ((rc = nmfx_admm_state_alloc(E))) return rc;     // auxH / Asum: the aux matrices of the "solve only" rounds
    if ((rc = ao_fused_alloc(E, admm_iter))) return rc; // bkX: X_prev
    const int nblk = (int)((cols ? E->np : E->mp) / 64);
    for (int r = 0; r < admm_iter; ++r) if ((rc = ao_l1inf_round(E, cols, prox, lam, r))) return rc;
    return nmfx_inner_finish(E, nblk, admm_iter, slot);
}
static bool ao_is_l1inf(int prox) { return prox == NMFX_PROX_L1INF || prox == NMFX_PROX_L1INF_T; }

static int ao_h_solve(nmfx_engine* E, int prox_h, double lam_h, int admm_iter, int64_t min_iter, double tol1,
                      double tol2, int64_t j) {
    int rc;
    if ((rc = nmfx_launch_prepare(E, E->xf32 + (int64_t)E->kp * E->np, 1, j, min_iter, tol1, tol2, -1.0))) return rc;
    E->himg_both = false;                              // H changes below
    ProfScope ps(E, "inner_h");
    if (ao_is_l1inf(prox_h)) return ao_l1inf_subproblem(E, true, prox_h, lam_h, admm_iter, E->inner_hist + j * 2);
    if (ao_fused_enabled(E, admm_iter))
        return ao_fused_subproblem(E, true, nullptr, prox_h, (float)lam_h, admm_iter, E->inner_hist + j * 2, (int)(j & 1));
    for (int r = 0; r < admm_iter; ++r) if ((rc = inner_cols(E, prox_h, (float)lam_h, r))) return rc;
    return nmfx_inner_finish(E, (int)(E->np / 64), admm_iter, E->inner_hist + j * 2);
}

static int ao_w_products(nmfx_engine* E, int64_t j, int64_t min_iter, double tol1, double tol2, bool sum_a = true) {
    int rc;
    const int64_t kk = (int64_t)E->kp * E->kp;
    E->ao_a_slabs = 0;
    if (ao_bf16(E)) {
        // the images of the H the sub-problem above produced (and of H^T, for the next H-side product): left by its fused launches, or built here
        if (!E->himg_both && (rc = nmfx_bf16_images_h(E, true))) return rc;
        if ((rc = nmfx_bf16_vht(E, false, 0, "wphase_noobj", false, 3))) return rc;      // kp = 64: H H^T slabs as a by-product
        const bool byprod = E->kp == 64;
        int hslabs = E->gsplit;                        // k = 128: H H^T from the images just built (four-term split products, like W^T W)
        if (!byprod && (rc = nmfx_bf16_gram_h(E, &hslabs))) return rc;
        { ProfScope ps(E, "sums");
          if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, byprod ? nmfx_bf16_hht_slabs(E) : hslabs, kk, E->HHt))) return rc;
          if (sum_a) { if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->bf_wsplit, E->mp * E->kp, E->auxW))) return rc; }
          else E->ao_a_slabs = E->bf_wsplit; }          // the fused W-side kernel adds the slabs itself
        return nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, -1.0);
    }
    if ((rc = nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_wphase(E, E->W[0], true, false))) return rc;
    { ProfScope ps(E, "sums");
      if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
      if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->wsplit, E->mp * E->kp, E->auxW))) return rc; }
    return nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, -1.0);
}

// One outer iteration with the inversions beside the products (ao_overlap):
//   gram(W^T W slabs) | product_h [B^T slabs + objective || (W^T W + rho I)^-1] | pack (+ obj[j], stop rule) | rounds of H |
//   gram(H H^T slabs) | product_w [A slabs || (H H^T + rho I)^-1] | rounds of W (they add the A slabs themselves)
static int aoadmm_eu_iteration_overlap(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h,
                                       int admm_iter, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if (!E->wimg_ok && (rc = nmfx_bf16_images_w(E, E->W[0], 0))) return rc;
    E->wimg_ok = true;
    if (!E->himg_both && (rc = nmfx_bf16_images_h(E, true))) return rc;
    int gslabs = 64;
    if ((rc = nmfx_bf16_gram_tn(E, &gslabs))) return rc;
    if ((rc = nmfx_bf16_sk_product(E, 0, true, E->G_part, gslabs, -1.0, "hphase"))) return rc;
    static const bool packed = getenv("NMFX_AO_PACK") && atoi(getenv("NMFX_AO_PACK")) == 1;      // (A/B: the pack launch of the first form)
    if (packed) { if ((rc = nmfx_bf16_pack_sk(E, j, min_iter, tol1, tol2))) return rc; }
    else {         // the rounds of H sum the B^T slabs themselves; their first launch records obj[j] and applies the stop rule
        E->ao_b_src = E->sk[0].slabs; E->ao_b_cnt = E->sk[0].cnt; E->ao_rec_nobj = E->sk[0].nseg;
        E->ao_rec_j = j; E->ao_rec_min_iter = min_iter; E->ao_rec_tol1 = tol1; E->ao_rec_tol2 = tol2;
    }
    E->himg_both = false;                              // H changes below
    { ProfScope ps(E, "inner_h");
      rc = ao_fused_subproblem(E, true, nullptr, prox_h, (float)lam_h, admm_iter, E->inner_hist + j * 2, (int)(j & 1)); }
    E->ao_b_src = nullptr; E->ao_b_cnt = nullptr;
    if (rc) return rc;
    if (!E->himg_both && (rc = nmfx_bf16_images_h(E, true))) return rc;
    int hslabs = 32;                                   // (HHt_part holds at least 40 slabs)
    if ((rc = nmfx_bf16_gram_h(E, &hslabs))) return rc;
    if ((rc = nmfx_bf16_sk_product(E, 1, false, E->HHt_part, hslabs, -1.0, "wphase_noobj"))) return rc;
    E->ao_a_slabs = E->sk[1].maxslab; E->ao_a_src = E->sk[1].slabs; E->ao_a_cnt = E->sk[1].cnt;
    E->wimg_ok = false;                                // W changes below
    { ProfScope ps(E, "inner_w");
      rc = ao_fused_subproblem(E, false, W, prox_w, (float)lam_w, admm_iter, E->inner_hist + j * 2 + 1, (int)(j & 1)); }
    E->ao_a_slabs = 0; E->ao_a_src = nullptr; E->ao_a_cnt = nullptr;
    if (rc) return rc;
    return ao_new_pair_objective(E);
}

static int aoadmm_eu_iteration(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h,
                               int admm_iter, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    const bool any_l1inf = ao_is_l1inf(prox_w) || ao_is_l1inf(prox_h);
    if (!any_l1inf && ao_overlap(E, admm_iter) && ao_fused_enabled(E, admm_iter))
        return aoadmm_eu_iteration_overlap(E, prox_w, lam_w, prox_h, lam_h, admm_iter, min_iter, tol1, tol2, j);
    // ---- H sub-problem: admm_ls_update(v, w, h, dual_h) ----
    if ((rc = ao_h_products(E))) return rc;
    if ((rc = ao_h_solve(E, prox_h, lam_h, admm_iter, min_iter, tol1, tol2, j))) return rc;
    // ---- W sub-problem: admm_ls_update(v.T, h.T, w.T, dual_w.T) ----
    if ((rc = ao_w_products(E, j, min_iter, tol1, tol2, ao_is_l1inf(prox_w) || !ao_fused_enabled(E, admm_iter)))) return rc;
    E->wimg_ok = false;                                // W changes below
    if (ao_is_l1inf(prox_w)) {                         // (the summed right-hand side moves to Asum: auxW receives the aux matrix)
        ProfScope ps(E, "inner_w");
        if ((rc = nmfx_admm_state_alloc(E))) return rc;
        NMFX_HIP(hipMemcpyAsync(E->Asum, E->auxW, (size_t)E->mp * E->kp * sizeof(float), hipMemcpyDeviceToDevice, E->stream));
        if ((rc = ao_l1inf_subproblem(E, false, prox_w, lam_w, admm_iter, E->inner_hist + j * 2 + 1))) return rc;
        return ao_new_pair_objective(E);
    }
    { ProfScope ps(E, "inner_w");
      if (ao_fused_enabled(E, admm_iter)) {
          if ((rc = ao_fused_subproblem(E, false, W, prox_w, (float)lam_w, admm_iter, E->inner_hist + j * 2 + 1, (int)(j & 1)))) return rc;
      } else {
          for (int r = 0; r < admm_iter; ++r) if ((rc = inner_rows(E, W, prox_w, (float)lam_w, r))) return rc;
          if ((rc = nmfx_inner_finish(E, (int)(E->mp / 64), admm_iter, E->inner_hist + j * 2 + 1))) return rc;
      } }
    // ---- objective of the new pair (utils.py:29), summed by the next pack / finish ----
    return ao_new_pair_objective(E);
}

static int aoadmm_kl_iteration(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h,
                               int admm_iter, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    const int64_t kk = (int64_t)E->kp * E->kp;
    const int* stop = &E->state->inner_stop;
    // ---- H sub-problem ----
    const bool fuse_g = nmfx_hphase_can_fuse_gram(E);
    if (!fuse_g && (rc = nmfx_launch_gram_tn(E, W, E->mp, E->G_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_hphase(E, W, fuse_g, E->S))) return rc;
    if ((rc = nmfx_launch_pack(E))) return rc;
    if ((rc = nmfx_launch_prepare(E, E->xf32 + (int64_t)E->kp * E->np, 1, j, min_iter, tol1, tol2, -1.0))) return rc;
    for (int r = 0; r < admm_iter; ++r) {
        if (r > 0) {
            if ((rc = nmfx_launch_hphase(E, W, false, E->S, stop))) return rc;
            if ((rc = nmfx_launch_pack(E, stop))) return rc;
        }
        if (ao_is_l1inf(prox_h)) rc = ao_l1inf_round(E, true, prox_h, lam_h, r);      // (r5: the reference runs these too, until its Cholesky fails)
        else rc = nmfx_inner_cols(E, E->Minv, E->auxH, 0, prox_h, (float)lam_h, r);
        if (rc) return rc;
        if ((rc = nmfx_launch_kl_vaux(E, W, E->auxH, stop))) return rc;
    }
    if ((rc = nmfx_inner_finish(E, (int)(E->np / 64), admm_iter, E->inner_hist + j * 2))) return rc;
    // ---- W sub-problem (transposed data) ----
    if ((rc = nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
    if ((rc = nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, -1.0))) return rc;
    for (int r = 0; r < admm_iter; ++r) {
        if ((rc = nmfx_launch_wphase(E, W, true, false, false, E->H, E->S, stop))) return rc;
        if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->wsplit, E->mp * E->kp, E->Asum))) return rc;
        if (ao_is_l1inf(prox_w)) rc = ao_l1inf_round(E, false, prox_w, lam_w, r);
        else rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 0, prox_w, (float)lam_w, r);
        if (rc) return rc;
        if ((rc = nmfx_launch_kl_vaux(E, E->auxW, E->H, stop))) return rc;
    }
    if ((rc = nmfx_inner_finish(E, (int)(E->mp / 64), admm_iter, E->inner_hist + j * 2 + 1))) return rc;
    return nmfx_launch_wphase(E, W, false, true, true);            // KL objective (utils.py:21-26)
}

// The KL-loss iteration on the split-bf16 kernels (r4; k padded to 64 or 128, one GPU; NMFX_KL_BF16=0 keeps the exact-f32 launches
// above).  Per round of a sub-problem: the right-hand side product with S in the place of V (the Euclidean product kernel on the
// tile-major S), the round kernel, the images of the aux matrix, and the m x n auxiliaries where the product W h_aux stands in the
// accumulators (xyt32_bf16_kernel<..., VAUX>).  S and dual_v live tile-major in the orientation of the sub-problem (rows n for H,
// rows m for W) and are transposed at the two switches of an outer iteration.  16384 x 8192, k = 128, ten rounds each: 26.0 -> see
// DESIGN 4b (the exact-f32 auxiliaries alone took 933 us per round).
static bool ao_kl_bf16(const nmfx_engine* E) {
    static const bool on = !(getenv("NMFX_KL_BF16") && atoi(getenv("NMFX_KL_BF16")) == 0);
    return on && ao_bf16(E);
}

// KL(V, W H) partials of the current pair from the images of W and H (split-bf16 product, one pass over the tile-major V)
static int ao_kl_objective(nmfx_engine* E) {
    int rc;
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if ((rc = nmfx_bf16_images_w(E, E->W[0], 0))) return rc;
    E->wimg_ok = true;
    if ((rc = nmfx_bf16_images_h(E, false))) return rc;
    return nmfx_bf16_kl_objective(E);
}

static int aoadmm_kl_iteration_bf16(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h,
                                    int admm_iter, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    const int64_t kk = (int64_t)E->kp * E->kp;
    const int* stop = &E->state->inner_stop;
    const int64_t nobj32 = E->obj_count;                               // partials of the KL objective pass that closed the iteration before (ao_kl_objective)
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if ((rc = nmfx_bf16_kl_state(E, false))) return rc;
    // ---- H sub-problem (state in the orientation of V^T) ----
    // r5: the first product of a sub-problem reads S where the other sub-problem's auxiliaries left it (transposing requests, NMFX_KL_GATHER=0:
    // a transposed copy as before); only dual_v changes orientation between the sub-problems
    static const bool gather = !(getenv("NMFX_KL_GATHER") && atoi(getenv("NMFX_KL_GATHER")) == 0);
    if ((rc = nmfx_bf16_kl_orient(E, 0, true, !gather))) return rc;
    if (!E->wimg_ok && (rc = nmfx_bf16_images_w(E, W, 0))) return rc;  // W^T images: Y of the products and of the auxiliaries (left by the objective pass)
    // r5: the auxiliaries of round r also form the product of round r + 1 (S stays in registers; NMFX_KL_FUSE=0: separate launches).  At k padded to
    // 64 the products' Gram by-product (W^T W, H H^T) is that of round 0's launch: the slabs stay where they are, the fixed factor does not change
    static const bool fuse = !(getenv("NMFX_KL_FUSE") && atoi(getenv("NMFX_KL_FUSE")) == 0);
    for (int r = 0; r < admm_iter; ++r) {
        if (r == 0 || !fuse) {
            E->xyt_flag2 = r > 0 ? stop : nullptr;
            rc = nmfx_bf16_kl_product(E, 0, 4, nullptr, r == 0 && E->kl_s_side != 0);   // B^T slabs = S^T W (FOUR terms: with three the KL objective history left its 5e-5 bar -- 1.2e-4 at k = 64, the objective near its optimum is 1e-5 of sum V)
            E->xyt_flag2 = nullptr;
            if (rc) return rc;
        }
        if (E->kp == 64) rc = nmfx_bf16_pack_t(E, E->G_part, nmfx_bf16_g_slabs(E), nobj32);
        else {
            if (r == 0 && (rc = nmfx_launch_gram_tn(E, W, E->mp, E->G_part, E->gsplit))) return rc;
            rc = nmfx_bf16_pack_t(E, E->G_part, E->gsplit, nobj32);
        }
        if (rc) return rc;
        if (r == 0 && (rc = nmfx_launch_prepare(E, E->xf32 + (int64_t)E->kp * E->np, 1, j, min_iter, tol1, tol2, -1.0))) return rc;
        if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 0, prox_h, (float)lam_h, r))) return rc;
        if ((rc = nmfx_bf16_images_h(E, true, E->auxH))) return rc;    // (h_aux)^T images: Z of the auxiliaries
        if (fuse) rc = nmfx_bf16_vaux_fused(E, 0, stop, E->nrm_part + (int64_t)(r & 1) * (E->np / 64) * 4, (int)(E->np / 64), r == admm_iter - 1);
        else rc = nmfx_bf16_vaux(E, 0, stop);
        if (rc) return rc;
    }
    if (admm_iter > 0) E->kl_s_side = 0;                               // (the last round's auxiliaries have stored S in this orientation)
    if ((rc = nmfx_inner_finish(E, (int)(E->np / 64), admm_iter, E->inner_hist + j * 2))) return rc;
    // ---- W sub-problem (transposed data: the orientation of V) ----
    if ((rc = nmfx_bf16_kl_orient(E, 1, true, !gather))) return rc;
    if ((rc = nmfx_bf16_images_h(E, false))) return rc;                // H images: Y
    if ((rc = nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
    if ((rc = nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, -1.0))) return rc;
    for (int r = 0; r < admm_iter; ++r) {
        if (r == 0 || !fuse) {
            E->xyt_flag2 = r > 0 ? stop : nullptr;
            rc = nmfx_bf16_kl_product(E, 1, 4, nullptr, r == 0 && E->kl_s_side != 1);   // A slabs = S H^T
            E->xyt_flag2 = nullptr;
            if (rc) return rc;
        }
        if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->bf_wsplit, E->mp * E->kp, E->Asum))) return rc;
        if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 0, prox_w, (float)lam_w, r))) return rc;
        if ((rc = nmfx_bf16_images_w(E, E->auxW, 0))) return rc;       // w_aux images: Z
        if (fuse) rc = nmfx_bf16_vaux_fused(E, 1, stop, E->nrm_part + (int64_t)(r & 1) * (E->mp / 64) * 4, (int)(E->mp / 64), r == admm_iter - 1);
        else rc = nmfx_bf16_vaux(E, 1, stop);
        if (rc) return rc;
    }
    if (admm_iter > 0) E->kl_s_side = 1;
    if ((rc = nmfx_inner_finish(E, (int)(E->mp / 64), admm_iter, E->inner_hist + j * 2 + 1))) return rc;
    E->himg_both = false;
    return ao_kl_objective(E);                                         // KL objective of the new pair (utils.py:21-26)
}

extern "C" int nmfx_aoadmm_run(nmfx_handle_t E, int distance, int prox_w, double lambda_w, int prox_h,
                               double lambda_h, int admm_iter, int64_t min_iter, double tol1, double tol2,
                               int64_t first, int64_t count) {
    if (!E) return NMFX_E_ARG;
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    if (distance != NMFX_EU && distance != NMFX_KL) { E->err = "Unknown loss function type."; return NMFX_E_ARG; }
    auto known = [](int p) { return p == NMFX_PROX_NN || p == NMFX_PROX_L1N || p == NMFX_PROX_L1INF || p == NMFX_PROX_L1INF_T; };
    if (!known(prox_w) || !known(prox_h)) { E->err = "Unknown prox_type."; return NMFX_E_ARG; }
    const bool any_l1inf = ao_is_l1inf(prox_w) || ao_is_l1inf(prox_h);
    if (any_l1inf && E->kp > 128) {
        E->err = "ao_admm with prox 'l1inf' / 'l1inf_transpose': at most 128 components in this build"; return NMFX_E_ARG; }
    if (first < 0 || count < 0 || admm_iter < 0) { E->err = "negative range"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    E->anls_a_ready = false; E->kl_h_iter = -2;
    int rc;
    if ((rc = nmfx_enter_family(E, 2))) return rc;
    if ((rc = nmfx_aoadmm_alloc(E))) return rc;
    if (distance == NMFX_KL && (rc = nmfx_kl_state_alloc(E))) return rc;
    if ((rc = nmfx_ensure_inner_capacity(E, first + count + 1))) return rc;
    if ((rc = nmfx_ensure_obj_capacity(E, first + count + 2))) return rc;
    E->wsel = 0;
    E->w_in_place = true;
    if (E->kp > 128)           // composed from the generic product kernel (kernels_generic.hip)
        return distance == NMFX_EU
            ? nmfx_generic_aoadmm_run(E, prox_w, lambda_w, prox_h, lambda_h, admm_iter, min_iter, tol1, tol2, first, count)
            : nmfx_generic_aoadmm_kl_run(E, prox_w, lambda_w, prox_h, lambda_h, admm_iter, min_iter, tol1, tol2, first, count);
    if (distance != NMFX_EU) { E->lazy_objective = false; E->himg_both = false; }
    // (KL loss with 'l1inf*': the exact-f32 launches -- the operator's rounds read the f32 right-hand sides those leave; the runs
    //  end after a few outer iterations anyway, in the reference's LinAlgError)
    const bool kl_bf = distance == NMFX_KL && ao_kl_bf16(E) && !any_l1inf;
    if (distance == NMFX_KL && any_l1inf) {
        if ((rc = nmfx_admm_state_alloc(E))) return rc;     // auxH / Asum of the "solve only" rounds
        if ((rc = ao_fused_alloc(E, admm_iter))) return rc; // bkX: X_prev
    }
    if (kl_bf && (first == 0 || E->obj_count <= 0)) E->wimg_ok = false;
    if (first == 0 && count > 0 && !(distance == NMFX_EU && ao_bf16(E))) {   // obj[0] of the initial factors (ao_admm.py:256)
        if (kl_bf) rc = ao_kl_objective(E);
        else rc = nmfx_launch_wphase(E, E->W[0], false, true, distance == NMFX_KL);
        if (rc) return rc;
    }
    for (int64_t j = first; j < first + count; ++j) {
        rc = distance == NMFX_EU
            ? aoadmm_eu_iteration(E, prox_w, lambda_w, prox_h, lambda_h, admm_iter, min_iter, tol1, tol2, j)
            : kl_bf ? aoadmm_kl_iteration_bf16(E, prox_w, lambda_w, prox_h, lambda_h, admm_iter, min_iter, tol1, tol2, j)
                    : aoadmm_kl_iteration(E, prox_w, lambda_w, prox_h, lambda_h, admm_iter, min_iter, tol1, tol2, j);
        if (rc) return rc;
    }
    return NMFX_OK;
}

// ---- row-sharded form (Euclidean loss) -----------------------------------------
static int ao_sharded_ready(nmfx_engine* E, int64_t j, bool any_k = false) {
    if (!E) return NMFX_E_ARG;
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    if (j < 0) { E->err = "negative iteration index"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    E->anls_a_ready = false; E->kl_h_iter = -2;
    E->wimg_ok = false; E->ao_images = false;          // (the phase entry points rebuild the W images themselves)
    int rc;
    // beyond 128 components (r4): the least-squares phases are composed from the generic kernels (kernels_generic.hip,
    // nmfx_generic_aoadmm_phase_*; r5: any number of components the handle takes); the fused W sub-problem stays at k <= 128
    if (!any_k && (rc = nmfx_small_k_only(E, "row-sharded AO-ADMM with the speculative (fused) W rounds"))) return rc;
    if ((rc = nmfx_enter_family(E, 2))) return rc;
    if ((rc = nmfx_aoadmm_alloc(E))) return rc;
    if ((rc = nmfx_ensure_inner_capacity(E, j + 2))) return rc;
    if ((rc = nmfx_ensure_obj_capacity(E, j + 3))) return rc;
    E->wsel = 0;
    E->w_in_place = true;
    return NMFX_OK;
}

extern "C" int nmfx_aoadmm_phase_h_products(nmfx_handle_t E, int64_t j) {
    int rc = ao_sharded_ready(E, j, true); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_aoadmm_phase(E, 0, 0, 0.0, 0, 0, 0.0, 0.0, j, 0);
    if (j == 0 && (rc = ao_new_pair_objective(E))) return rc;                      // obj[0] partials (ao_admm.py:256)
    return ao_h_products(E);
}

extern "C" int nmfx_aoadmm_phase_h_solve(nmfx_handle_t E, int prox_h, double lambda_h, int admm_iter,
                                         int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc = ao_sharded_ready(E, j, true); if (rc) return rc;
    if (prox_h != NMFX_PROX_NN && prox_h != NMFX_PROX_L1N) { E->err = "Unknown prox_type."; return NMFX_E_ARG; }
    if (E->kp > 128) return nmfx_generic_aoadmm_phase(E, 1, prox_h, lambda_h, admm_iter, min_iter, tol1, tol2, j, 0);
    return ao_h_solve(E, prox_h, lambda_h, admm_iter, min_iter, tol1, tol2, j);
}

extern "C" int nmfx_aoadmm_phase_w_products(nmfx_handle_t E, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc = ao_sharded_ready(E, j, true); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_aoadmm_phase(E, 2, 0, 0.0, 0, min_iter, tol1, tol2, j, 0);
    return ao_w_products(E, j, min_iter, tol1, tol2);
}

// One inner round of the W sub-problem on this rank's rows; leaves this rank's norm sums of
// the round in the f64 exchange buffer [1..4] for the caller to all-reduce.
extern "C" int nmfx_aoadmm_phase_w_round(nmfx_handle_t E, int prox_w, double lambda_w, int round) {
    int rc = ao_sharded_ready(E, 0, true); if (rc) return rc;
    if (prox_w != NMFX_PROX_NN && prox_w != NMFX_PROX_L1N) { E->err = "Unknown prox_type."; return NMFX_E_ARG; }
    if (round < 0) { E->err = "negative round"; return NMFX_E_ARG; }
    if (E->kp > 128) return nmfx_generic_aoadmm_phase(E, 3, prox_w, lambda_w, 0, 0, 0.0, 0.0, 0, round);
    if ((rc = inner_rows(E, E->W[0], prox_w, (float)lambda_w, round, round > 0 ? E->xf64 + 1 : nullptr))) return rc;
    hipLaunchKernelGGL(ao_norm_gather_kernel, dim3(1), dim3(256), 0, E->stream, E->state, E->nrm_part,
                       (int)(E->mp / 64), round, E->xf64 + 1);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// After the last round's all-reduce: inner-iteration bookkeeping and the objective partials of
// the new pair (summed into the exchange buffer by the next nmfx_aoadmm_phase_h_products or by
// nmfx_objective_partial).
extern "C" int nmfx_aoadmm_phase_w_close(nmfx_handle_t E, int admm_iter, int64_t j) {
    int rc = ao_sharded_ready(E, j, true); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_aoadmm_phase(E, 4, 0, 0.0, admm_iter, 0, 0.0, 0.0, j, 0);
    if ((rc = nmfx_inner_finish(E, (int)(E->mp / 64), admm_iter, E->inner_hist + j * 2 + 1, E->xf64 + 1))) return rc;
    return ao_new_pair_objective(E);
}

// ---- row-sharded form, KL loss (ao_admm.py:71-101, 277-283) ---------------------------------------------------------
// The KL inner loop keeps the V-sized products INSIDE the rounds (w.T @ (v_aux + dual_v) for the H sub-problem,
// (v_aux + dual_v) @ h.T for the W sub-problem), so the row-sharded form exchanges once per round:
//   H sub-problem, round r:  nmfx_aoadmm_kl_phase_h_products(j, r)  ->  all-reduce of [W^T S | W^T W] (round 0: and the objective)
//                            nmfx_aoadmm_kl_phase_h_round(..., r)      replicated solve / prox / dual step, rank-local v_aux step
//   then nmfx_aoadmm_kl_phase_h_close: inner-iteration bookkeeping, H H^T and its inverse for the W side (replicated)
//   W sub-problem, round r:  nmfx_aoadmm_kl_phase_w_round(..., r)   ->  all-reduce of the 4 norm sums (`terminate` looks at all of W)
//   then nmfx_aoadmm_kl_phase_w_close: bookkeeping + the KL objective partials of the new pair.
// After the inner stop has fired the remaining rounds are no-ops on every rank (the decision is taken from replicated /
// all-reduced numbers), their exchanges move stale buffers that nothing reads.
static int ao_kl_sharded_ready(nmfx_engine* E, int64_t j) {
    if (E && E->kp > 128) {                            // r5: beyond 128 components the phases are composed from the generic kernels (any k the handle takes)
        if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
        if (j < 0) { E->err = "negative iteration index"; return NMFX_E_ARG; }
        NMFX_HIP(hipSetDevice(E->device));
        E->anls_a_ready = false; E->kl_h_iter = -2;
        E->wimg_ok = false; E->ao_images = false;
        int rc;
        if ((rc = nmfx_enter_family(E, 2))) return rc;
        if ((rc = nmfx_aoadmm_alloc(E))) return rc;
        if ((rc = nmfx_kl_state_alloc(E))) return rc;
        if ((rc = nmfx_ensure_inner_capacity(E, j + 2))) return rc;
        if ((rc = nmfx_ensure_obj_capacity(E, j + 3))) return rc;
        E->wsel = 0; E->w_in_place = true;
        E->lazy_objective = false; E->himg_both = false;
        return NMFX_OK;
    }
    int rc = ao_sharded_ready(E, j); if (rc) return rc;
    if ((rc = nmfx_kl_state_alloc(E))) return rc;
    E->lazy_objective = false; E->himg_both = false;
    return NMFX_OK;
}
static bool kl_prox_ok(int p) { return p == NMFX_PROX_NN || p == NMFX_PROX_L1N; }

extern "C" int nmfx_aoadmm_kl_phase_h_products(nmfx_handle_t E, int64_t j, int round) {
    int rc = ao_kl_sharded_ready(E, j); if (rc) return rc;
    if (round < 0) { E->err = "negative round"; return NMFX_E_ARG; }
    if (E->kp > 128) return nmfx_generic_aoadmm_kl_phase(E, 0, 0, 0.0, 0, 0, 0.0, 0.0, j, round);
    float* W = E->W[0];
    if (round == 0) {
        if (j == 0 && (rc = nmfx_launch_wphase(E, W, false, true, true))) return rc;      // obj[0] partials (ao_admm.py:256)
        const bool fuse_g = nmfx_hphase_can_fuse_gram(E);
        if (!fuse_g && (rc = nmfx_launch_gram_tn(E, W, E->mp, E->G_part, E->gsplit))) return rc;
        if ((rc = nmfx_launch_hphase(E, W, fuse_g, E->S))) return rc;
        return nmfx_launch_pack(E);
    }
    const int* stop = &E->state->inner_stop;
    if ((rc = nmfx_launch_hphase(E, W, false, E->S, stop))) return rc;
    return nmfx_launch_pack(E, stop);
}

extern "C" int nmfx_aoadmm_kl_phase_h_round(nmfx_handle_t E, int prox_h, double lambda_h, int round, int64_t min_iter,
                                            double tol1, double tol2, int64_t j) {
    int rc = ao_kl_sharded_ready(E, j); if (rc) return rc;
    if (!kl_prox_ok(prox_h)) { E->err = "Unknown prox_type."; return NMFX_E_ARG; }
    if (round < 0) { E->err = "negative round"; return NMFX_E_ARG; }
    if (E->kp > 128) return nmfx_generic_aoadmm_kl_phase(E, 1, prox_h, lambda_h, 0, min_iter, tol1, tol2, j, round);
    if (round == 0 && (rc = nmfx_launch_prepare(E, E->xf32 + (int64_t)E->kp * E->np, 1, j, min_iter, tol1, tol2, -1.0))) return rc;
    if ((rc = nmfx_inner_cols(E, E->Minv, E->auxH, 0, prox_h, (float)lambda_h, round))) return rc;
    return nmfx_launch_kl_vaux(E, E->W[0], E->auxH, &E->state->inner_stop);
}

extern "C" int nmfx_aoadmm_kl_phase_h_close(nmfx_handle_t E, int admm_iter, int64_t min_iter, double tol1, double tol2,
                                            int64_t j) {
    int rc = ao_kl_sharded_ready(E, j); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_aoadmm_kl_phase(E, 2, 0, 0.0, admm_iter, min_iter, tol1, tol2, j, 0);
    const int64_t kk = (int64_t)E->kp * E->kp;
    if ((rc = nmfx_inner_finish(E, (int)(E->np / 64), admm_iter, E->inner_hist + j * 2))) return rc;
    if ((rc = nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
    return nmfx_launch_prepare(E, E->HHt, 0, j, min_iter, tol1, tol2, -1.0);
}

// One round of the W sub-problem on this rank's rows; leaves this rank's norm sums of the round in the f64 exchange
// buffer [1..4] for the caller to all-reduce (round r + 1 takes the stop decision of round r from the reduced sums).
extern "C" int nmfx_aoadmm_kl_phase_w_round(nmfx_handle_t E, int prox_w, double lambda_w, int round) {
    int rc = ao_kl_sharded_ready(E, 0); if (rc) return rc;
    if (!kl_prox_ok(prox_w)) { E->err = "Unknown prox_type."; return NMFX_E_ARG; }
    if (round < 0) { E->err = "negative round"; return NMFX_E_ARG; }
    if (E->kp > 128) return nmfx_generic_aoadmm_kl_phase(E, 3, prox_w, lambda_w, 0, 0, 0.0, 0.0, 0, round);
    float* W = E->W[0];
    const int* stop = &E->state->inner_stop;
    if ((rc = nmfx_launch_wphase(E, W, true, false, false, E->H, E->S, stop))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->wsplit, E->mp * E->kp, E->Asum))) return rc;
    if ((rc = nmfx_inner_rows(E, E->Asum, W, E->Minv, E->auxW, 0, prox_w, (float)lambda_w, round,
                              round > 0 ? E->xf64 + 1 : nullptr))) return rc;
    if ((rc = nmfx_launch_kl_vaux(E, E->auxW, E->H, stop))) return rc;
    hipLaunchKernelGGL(ao_norm_gather_kernel, dim3(1), dim3(256), 0, E->stream, E->state, E->nrm_part,
                       (int)(E->mp / 64), round, E->xf64 + 1);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

extern "C" int nmfx_aoadmm_kl_phase_w_close(nmfx_handle_t E, int admm_iter, int64_t j) {
    int rc = ao_kl_sharded_ready(E, j); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_aoadmm_kl_phase(E, 4, 0, 0.0, admm_iter, 0, 0.0, 0.0, j, 0);
    if ((rc = nmfx_inner_finish(E, (int)(E->mp / 64), admm_iter, E->inner_hist + j * 2 + 1, E->xf64 + 1))) return rc;
    return nmfx_launch_wphase(E, E->W[0], false, true, true);                             // KL objective partials (utils.py:21-26)
}

// Row-sharded W sub-problem with ONE exchange instead of one per round: all admm_iter rounds run speculatively on
// this rank's rows (ao_fused_rows_kernel) and the four norm sums of every round go to the f64 exchange buffer
// [8 + 4 round + component]; after the caller's all-reduce of that table nmfx_aoadmm_phase_w_repair finds the round at
// which the reference's `terminate` (ao_admm.py:33-43, norms over ALL rows of W) would have stopped and, if that is
// before the last one, reruns exactly that many rounds from the saved start -- what the single-GPU path does with
// its block partials.  admm_iter <= NMFX_MAX_FUSED_ROUNDS (64).
extern "C" int nmfx_aoadmm_phase_w_fused(nmfx_handle_t E, int prox_w, double lambda_w, int admm_iter) {
    if (!E) return NMFX_E_ARG;
    if (prox_w != NMFX_PROX_NN && prox_w != NMFX_PROX_L1N) { E->err = "Unknown prox_type."; return NMFX_E_ARG; }
    if (admm_iter < 1 || admm_iter > NMFX_MAX_FUSED_ROUNDS) { E->err = "phase_w_fused: admm_iter out of range"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if ((rc = nmfx_small_k_only(E, "row-sharded AO-ADMM with the speculative (fused) W rounds"))) return rc;      // (beyond that: nmfx_aoadmm_phase_w_round, one exchange per round)
    if ((rc = ao_fused_alloc(E, admm_iter))) return rc;
    ProfScope ps(E, "inner_w");
    if ((rc = fused_rows_any(E, E->W[0], prox_w, (float)lambda_w, admm_iter, 0, nullptr))) return rc;
    const int nblk = (int)(E->mp / ao_fused_rows_rb(E));
    hipLaunchKernelGGL(nrm_table_kernel, dim3(1), dim3(256), 0, E->stream, E->nrm_rounds, nblk, admm_iter, E->xf64 + 8,
                       &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

extern "C" int nmfx_aoadmm_phase_w_repair(nmfx_handle_t E, int prox_w, double lambda_w, int admm_iter, int64_t j) {
    int rc = ao_sharded_ready(E, j); if (rc) return rc;
    if (admm_iter < 1 || admm_iter > NMFX_MAX_FUSED_ROUNDS) { E->err = "phase_w_repair: admm_iter out of range"; return NMFX_E_ARG; }
    { ProfScope ps(E, "inner_w");
      if ((rc = fused_rows_any(E, E->W[0], prox_w, (float)lambda_w, admm_iter, 1, E->inner_hist + j * 2 + 1, E->xf64 + 8))) return rc; }
    return ao_new_pair_objective(E);
}

// f64 exchange buffer [0] = this rank's objective partial of the current pair.
extern "C" int nmfx_objective_partial(nmfx_handle_t E) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    if (E->kp > 128 && (E->family == 2 || E->family == 3)) return NMFX_OK;      // (the composed AO-ADMM / ADMM phases close every iteration with it: xf64[0] holds it)
    if (E->kp > 128 && E->family == 4) return nmfx_generic_anls_phase(E, 0, 0.0, 0, 0.0, 0.0, 0);
    return ao_final_objective(E);
}

// bookkeeping of the last iteration's objective (obj_part is already filled)
extern "C" int nmfx_aoadmm_finish(nmfx_handle_t E, int64_t min_iter, double tol1, double tol2, int64_t done) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if (E->kp <= 128 && (rc = ao_final_objective(E))) return rc;      // (k > 128: the run left the objective partial in place)
    return nmfx_finish_b(E, min_iter, tol1, tol2, done);
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_aoadmm() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(nrm_table_kernel)) == hipSuccess ? 0 : -1; }
