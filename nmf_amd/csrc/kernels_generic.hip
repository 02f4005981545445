// Any number of components: the MUR loops for k > 128 (padded to a multiple of 128).
//
//   reference: nmf/mur.py:20-49 (updates), :119-131 (loop), nmf/utils.py:18-33 (objective).  The reference takes any
//   `factors` (nmf/nmf.py:32-35, nmf/mur.py:52); the tuned kernels of this library keep the k x k Gram matrices and a k-wide
//   factor panel in LDS / registers, which ends at k = 128.  Beyond that the iteration is composed from ONE tiled exact-f32 MFMA
//   product kernel (v_mfma_f32_16x16x4_f32: f32 inputs, f32 accumulation -- the arithmetic of the exact-f32 mode) used for every
//   contraction, with the objective / the KL quotient fused into the epilogue of the W H product (the m x n matrix `wh` is
//   never stored), plus element-wise update kernels:
//
//     Euclidean:  obj = 1/2 ||V - W H||^2 (fused);  H H^T;  A = V H^T;  D = W (H H^T);  W <- W A / (D + lw W + 1e-9);
//                 G = W^T W;  B = W^T V;  E = G H;  H <- H B / (E + lh H + 1e-9)
//     KL:         Q = V / (W H + 1e-9) with the KL objective fused;  a = W (Q H^T);  b = 1 H^T;  W <- 2a / (b + sqrt(b^2 + 4 lw a));
//                 Q' = V / (W H + 1e-9);  a' = H (W^T Q');  d = W^T 1;  H <- 2a' / (d + sqrt(d^2 + 4 lh a'))
//
// Same phase protocol as the other MUR paths (phase A leaves [W^T V | W^T W | column sums] + the objective in the exchange
// buffers, phase B records the objective, evaluates the stop rule and updates H), so the row-sharded drivers work unchanged.
// Product kernel: block = 128 x 128 outputs (4 waves x 64 x 64), contraction in chunks of 16 through LDS ([k][row] planes,
// row stride 144 floats: bank = 16 q + x for the MFMA fragment reads, conflict free), next chunk prefetched into registers
// while the current one is multiplied; operands may be contiguous along the contraction or along the output index (16-byte
// global loads either way); optional split of the contraction over gridDim.z (partials summed by sum_partials).
#include "nmfx_internal.h"
#include "kernels_small.h"

#define GX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int GX_T = 128, GX_KC = 16, GX_LD = 144;

enum { GX_STORE = 0, GX_RESID = 1, GX_KLQ = 2 };

// one operand's chunk [GX_KC][128]: global -> registers (two float4 per thread), registers -> LDS plane [k][GX_LD].
// KCONTIG: element (row, t) at base[row * ld + t]; else element (t, col) at base[t * ld + col].
// (plain float4 values, no struct: hipcc kept a two-member staging struct in scratch)
template <bool KCONTIG>
__device__ __forceinline__ void gx_load(const float* __restrict__ base, int64_t ld, int64_t k0, int tid, float4& r0, float4& r1) {
    if constexpr (KCONTIG) {
        const float* p = base + (int64_t)(tid >> 1) * ld + k0 + 8 * (tid & 1);
        r0 = *reinterpret_cast<const float4*>(p);
        r1 = *reinterpret_cast<const float4*>(p + 4);
    } else {
        const float* p = base + (k0 + (tid >> 5)) * ld + 4 * (tid & 31);
        r0 = *reinterpret_cast<const float4*>(p);
        r1 = *reinterpret_cast<const float4*>(p + 8 * ld);
    }
}
template <bool KCONTIG>
__device__ __forceinline__ void gx_store(float* __restrict__ plane, int tid, const float4& r0, const float4& r1) {
    if constexpr (KCONTIG) {
        float* q = plane + (8 * (tid & 1)) * GX_LD + (tid >> 1);
        q[0] = r0.x; q[GX_LD] = r0.y; q[2 * GX_LD] = r0.z; q[3 * GX_LD] = r0.w;
        q[4 * GX_LD] = r1.x; q[5 * GX_LD] = r1.y; q[6 * GX_LD] = r1.z; q[7 * GX_LD] = r1.w;
    } else {
        float* q = plane + (tid >> 5) * GX_LD + 4 * (tid & 31);
        *reinterpret_cast<float4*>(q) = r0;
        *reinterpret_cast<float4*>(q + 8 * GX_LD) = r1;
    }
}

// C[z][i][j] = sum_{t in split z} A(i, t) B(t, j)     (i < 128 gridDim.y, j < 128 gridDim.x)
//   AK: A(i, t) = A[i * lda + t], else A[t * lda + i];   BK: B(t, j) = B[j * ldb + t], else B[t * ldb + j]
//   MODE GX_STORE: C stored.  GX_RESID: nothing stored, part[block] = 1/2 sum (X - C)^2 (utils.py:29).
//   GX_KLQ: Q = X / (C + 1e-9) stored (mur.py:25,41) and, with part != nullptr, part[block] = the KL objective of the tile
//   (utils.py:23-26: x log(x / c) with inf / nan -> 0, - x + c).
template <bool AK, bool BK, int MODE>
__global__ __launch_bounds__(256) void gx_gemm_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    int64_t cstride, int64_t K, const float* __restrict__ X, int64_t ldx, double* __restrict__ part, const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ __attribute__((aligned(16))) float lds[2][2][GX_KC * GX_LD];        // [buffer][A / B][k][row]
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * GX_T, j0 = (int64_t)blockIdx.x * GX_T;
    const int64_t kper = K / gridDim.z, kbeg = kper * blockIdx.z;
    const float* Ab = AK ? A + i0 * lda : A + i0;
    const float* Bb = BK ? B + j0 * ldb : B + j0;
    const int wr = 64 * (wave >> 1), wc = 64 * (wave & 1);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 a0, a1, b0, b1;
    gx_load<AK>(Ab, lda, kbeg, tid, a0, a1);
    gx_load<BK>(Bb, ldb, kbeg, tid, b0, b1);
    const int nch = (int)(kper / GX_KC);
    for (int ch = 0; ch < nch; ++ch) {
        float* pa = lds[ch & 1][0];
        float* pb = lds[ch & 1][1];
        gx_store<AK>(pa, tid, a0, a1);
        gx_store<BK>(pb, tid, b0, b1);
        __syncthreads();                               // (two buffers: the chunk multiplied below is not the one written next time)
        {   // branch-free prefetch (behind the last chunk it fetches that chunk again): a conditionally assigned prefetch array
            // is what hipcc parks in scratch
            const int64_t kn = kbeg + (int64_t)(ch + 1 < nch ? ch + 1 : ch) * GX_KC;
            gx_load<AK>(Ab, lda, kn, tid, a0, a1);
            gx_load<BK>(Bb, ldb, kn, tid, b0, b1);
        }
#pragma unroll
        for (int u = 0; u < GX_KC / 4; ++u) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = pa[(4 * u + q) * GX_LD + wr + 16 * a + x];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = pb[(4 * u + q) * GX_LD + wc + 16 * b + x];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = GX_MFMA(av[a], bv[b], acc[a][b]);
        }
    }
    // acc[a][b][r] = C(i0 + wr + 16 a + 4 q + r, j0 + wc + 16 b + x)
    if (MODE == GX_STORE) {
        float* Cz = C + (int64_t)blockIdx.z * cstride;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    Cz[(i0 + wr + 16 * a + 4 * q + r) * ldc + j0 + wc + 16 * b + x] = acc[a][b][r];
        return;
    }
    double tot = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int64_t idx = (i0 + wr + 16 * a + 4 * q + r) * ldx + j0 + wc + 16 * b + x;
                const float xv = X[idx], cv = acc[a][b][r];
                if (MODE == GX_RESID) {
                    const float d = xv - cv;
                    if (b & 1) s1 += d * d; else s0 += d * d;
                } else {
                    C[(i0 + wr + 16 * a + 4 * q + r) * ldc + j0 + wc + 16 * b + x] = xv / (cv + 1e-9f);
                    if (part) {
                        float t = xv * logf(xv / cv);
                        t = (t != t || t == __builtin_inff() || t == -__builtin_inff()) ? 0.f : t;
                        if (b & 1) s1 += (t - xv) + cv; else s0 += (t - xv) + cv;
                    }
                }
            }
        tot += (double)(s0 + s1);
    }
    if (!part) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (MODE == GX_RESID ? 0.5 : 1.0) * (((red[0] + red[1]) + red[2]) + red[3]);
}

// X_new = X * Num / (Den + lam X + 1e-9)   (nmf/mur.py:29 / :45), 4 elements per thread
__global__ __launch_bounds__(256) void gx_eu_update_kernel(const float* __restrict__ Xold, const float* __restrict__ Num,
                                                           const float* __restrict__ Den, float lam, float* __restrict__ Xnew,
                                                           int64_t count4, const int* __restrict__ flag)
{
    if (*flag) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count4) return;
    const float4 xo = reinterpret_cast<const float4*>(Xold)[i], nu = reinterpret_cast<const float4*>(Num)[i], de = reinterpret_cast<const float4*>(Den)[i];
    float4 o;
    o.x = xo.x * nu.x / (de.x + lam * xo.x + 1e-9f); o.y = xo.y * nu.y / (de.y + lam * xo.y + 1e-9f);
    o.z = xo.z * nu.z / (de.z + lam * xo.z + 1e-9f); o.w = xo.w * nu.w / (de.w + lam * xo.w + 1e-9f);
    reinterpret_cast<float4*>(Xnew)[i] = o;
}

// KL (nmf/mur.py:24-27, 40-43): a = X * Num, X_new = 2 a / (s + sqrt(s^2 + 4 lam a)); s = the factor's sum (b = 1 H^T for W:
// factor = column of X; d = W^T 1 for H: factor = row of X).  Factors >= k are forced to zero (their 0 / 0 would be NaN).
template <bool ROWFACTOR>
__global__ __launch_bounds__(256) void gx_kl_update_kernel(const float* __restrict__ Xold, const float* __restrict__ Num,
                                                           const float* __restrict__ sums, float lam, float* __restrict__ Xnew,
                                                           int64_t rows, int64_t cols, int k, const int* __restrict__ flag)
{
    if (*flag) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t f = ROWFACTOR ? i / cols : i % cols;
    float o = 0.f;
    if (f < k) {
        const float a = Xold[i] * Num[i], s = sums[f];
        o = 2.f * a / (s + sqrtf(s * s + 4.f * lam * a));
    }
    Xnew[i] = o;
}

// sums over the rows of a [rows][cols] matrix's ... ROWSUM: out[r] = sum_c X[r][c] (one block per row);
// else column sums of a row range: part[blockIdx.x][c] = sum over 64 rows (summed by sum_partials)
__global__ __launch_bounds__(256) void gx_rowsum_kernel(const float* __restrict__ X, int64_t cols, float* __restrict__ out,
                                                        const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ float sh[4];
    const float* row = X + (int64_t)blockIdx.x * cols;
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) s += row[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void gx_colsum_part_kernel(const float* __restrict__ X, int64_t cols, int rows_per, float* __restrict__ part,
                                                             const int* __restrict__ flag)
{
    if (*flag) return;
    const float* base = X + (int64_t)blockIdx.x * rows_per * cols;
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        float s = 0.f;
        for (int r = 0; r < rows_per; ++r) s += base[(int64_t)r * cols + c];
        part[(int64_t)blockIdx.x * cols + c] = s;
    }
}

// phase B, first launch: the objective of the pair that entered this iteration (all-reduced in sharded runs) -> history, stop rule
__global__ void gx_record_kernel(const double* __restrict__ xf64, long long j, long long min_iter, double tol1, double tol2,
                                 DevState* __restrict__ st, double* __restrict__ obj_hist)
{
    if (st->flag) return;
    nmfx_record_objective(st, obj_hist, xf64[0], j, min_iter, tol1, tol2, threadIdx.x == 0);
}

template <typename T>
int gx_alloc(nmfx_engine* E, T** p, int64_t count) {
    if (*p) return NMFX_OK;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T)));
    NMFX_HIP(hipMemsetAsync(*p, 0, (size_t)count * sizeof(T), E->stream));
    return NMFX_OK;
}

// a split of the contraction that fills the CUs about twice: S divides K / 16
int gx_split(const nmfx_engine* E, int64_t tiles, int64_t K, int cap) {
    int64_t want = std::max<int64_t>(1, (2 * (int64_t)E->ncu + tiles - 1) / tiles);
    want = std::min<int64_t>(want, cap);
    const int64_t ch = K / GX_KC;
    while (want > 1 && ch % want) --want;
    return (int)want;
}

template <bool AK, bool BK>
int gx_launch(nmfx_engine* E, int mode, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t cstride,
              int64_t M, int64_t N, int64_t K, int S, const float* X, int64_t ldx, double* part) {
    const dim3 grid((unsigned)(N / GX_T), (unsigned)(M / GX_T), (unsigned)S), block(256);
    const int* flag = &E->state->flag;
    if (mode == GX_STORE) hipLaunchKernelGGL((gx_gemm_kernel<AK, BK, GX_STORE>), grid, block, 0, E->stream, A, lda, B, ldb, C, ldc, cstride, K, X, ldx, part, flag);
    else if (mode == GX_RESID) hipLaunchKernelGGL((gx_gemm_kernel<AK, BK, GX_RESID>), grid, block, 0, E->stream, A, lda, B, ldb, C, ldc, cstride, K, X, ldx, part, flag);
    else hipLaunchKernelGGL((gx_gemm_kernel<AK, BK, GX_KLQ>), grid, block, 0, E->stream, A, lda, B, ldb, C, ldc, cstride, K, X, ldx, part, flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int gx_buffers(nmfx_engine* E, bool kl) {
    int rc;
    const int64_t nblk = (E->mp / GX_T) * (E->np / GX_T);
    if ((rc = gx_alloc(E, &E->gx_part, nblk + 64))) return rc;
    if ((rc = gx_alloc(E, &E->gx_d, std::max(E->mp, E->np) * (int64_t)E->kp))) return rc;      // D = W HHt / E = G H / the update's numerator
    // slabs of the split products: B = W^T V (<= 8 x kp x np), the Gram matrices (<= 64 x kp x kp), the column sums of W
    if ((rc = gx_alloc(E, &E->gx_s, std::max<int64_t>(std::max<int64_t>(8 * (int64_t)E->kp * E->np, 64 * (int64_t)E->kp * E->kp), (E->mp / 64) * (int64_t)E->kp)))) return rc;
    if (kl && (rc = gx_alloc(E, &E->S, E->mp * E->np))) return rc;                              // the quotient Q (m x n), KL only
    return nmfx_need_v(E);
}

// split-K product into the slab buffer gx_s, summed into `out`
template <bool AK, bool BK>
int gx_split_product(nmfx_engine* E, const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t M, int64_t N, int64_t K,
                     int cap) {
    int rc;
    const int S = gx_split(E, (M / GX_T) * (N / GX_T), K, cap);
    if (S == 1) return gx_launch<AK, BK>(E, GX_STORE, A, lda, B, ldb, out, N, 0, M, N, K, 1, nullptr, 0, nullptr);
    if ((rc = gx_launch<AK, BK>(E, GX_STORE, A, lda, B, ldb, E->gx_s, N, M * N, M, N, K, S, nullptr, 0, nullptr))) return rc;
    return nmfx_launch_sum_partials(E, E->gx_s, S, M * N, out);
}

}  // namespace

// ---- MUR, Euclidean ----------------------------------------------------------------------------------------------------------
int nmfx_generic_mur_phase_a(nmfx_engine* E, int distance, double lambda, int64_t j) {
    int rc;
    const bool kl = distance == NMFX_KL;
    if ((rc = gx_buffers(E, kl))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    const float* W = E->W[j & 1];
    float* Wn = E->W[(j + 1) & 1];
    const int64_t nblk = (mp / GX_T) * (np / GX_T);
    float* xB = E->xf32;                               // [kp][np]
    float* xG = E->xf32 + kp * np;                     // [kp][kp]
    float* xS = xG + kp * kp;                          // [kp] column sums of W (KL)
    if (!kl) {
        { ProfScope ps(E, "objective");                // 1/2 ||V - W H||^2 of the pair entering the iteration
          if ((rc = gx_launch<true, false>(E, GX_RESID, W, kp, E->H, np, nullptr, 0, 0, mp, np, kp, 1, E->V, np, E->gx_part))) return rc; }
        if ((rc = nmfx_launch_obj_reduce(E, nblk, E->gx_part))) return rc;
        { ProfScope ps(E, "gram_nt");
          if ((rc = gx_split_product<true, true>(E, E->H, np, E->H, np, E->HHt, kp, kp, np, 64))) return rc; }
        { ProfScope ps(E, "wphase");                   // A = V H^T
          if ((rc = gx_split_product<true, true>(E, E->V, np, E->H, np, E->A_part, mp, kp, np, 1))) return rc; }
        { ProfScope ps(E, "w_update");
          if ((rc = gx_launch<true, false>(E, GX_STORE, W, kp, E->HHt, kp, E->gx_d, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr))) return rc;
          const int64_t c4 = mp * kp / 4;
          hipLaunchKernelGGL(gx_eu_update_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, E->stream, W, (const float*)E->A_part,
                             (const float*)E->gx_d, (float)lambda, Wn, c4, (const int*)&E->state->flag);
          NMFX_HIP(hipGetLastError()); }
        { ProfScope ps(E, "gram_tn");
          if ((rc = gx_split_product<false, false>(E, Wn, kp, Wn, kp, xG, kp, kp, mp, 64))) return rc; }
        { ProfScope ps(E, "hphase");                   // B = W^T V
          if ((rc = gx_split_product<false, false>(E, Wn, kp, E->V, np, xB, kp, np, mp, 8))) return rc; }
        return NMFX_OK;
    }
    { ProfScope ps(E, "objective");                    // Q = V / (W H + 1e-9) and the KL objective of the pair entering the iteration
      if ((rc = gx_launch<true, false>(E, GX_KLQ, W, kp, E->H, np, E->S, np, 0, mp, np, kp, 1, E->V, np, E->gx_part))) return rc; }
    if ((rc = nmfx_launch_obj_reduce(E, nblk, E->gx_part))) return rc;
    { ProfScope ps(E, "wphase");                       // Q H^T
      if ((rc = gx_split_product<true, true>(E, E->S, np, E->H, np, E->A_part, mp, kp, np, 1))) return rc; }
    { ProfScope ps(E, "w_update");
      hipLaunchKernelGGL(gx_rowsum_kernel, dim3((unsigned)kp), dim3(256), 0, E->stream, (const float*)E->H, np, E->HHt, (const int*)&E->state->flag);
      const int64_t cnt = mp * kp;
      hipLaunchKernelGGL((gx_kl_update_kernel<false>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, E->stream, W, (const float*)E->A_part,
                         (const float*)E->HHt, (float)lambda, Wn, mp, kp, E->k, (const int*)&E->state->flag);
      NMFX_HIP(hipGetLastError()); }
    { ProfScope ps(E, "hphase");                       // Q' = V / (W_new H + 1e-9), B = W_new^T Q', d = W_new^T 1
      if ((rc = gx_launch<true, false>(E, GX_KLQ, Wn, kp, E->H, np, E->S, np, 0, mp, np, kp, 1, E->V, np, nullptr))) return rc;
      if ((rc = gx_split_product<false, false>(E, Wn, kp, E->S, np, xB, kp, np, mp, 8))) return rc;
      const int rb = (int)(mp / 64);
      hipLaunchKernelGGL(gx_colsum_part_kernel, dim3((unsigned)rb), dim3(256), 0, E->stream, (const float*)Wn, kp, 64, E->gx_s, (const int*)&E->state->flag);
      NMFX_HIP(hipGetLastError());
      if ((rc = nmfx_launch_sum_partials(E, E->gx_s, rb, kp, xS))) return rc; }
    return NMFX_OK;
}

int nmfx_generic_mur_phase_b(nmfx_engine* E, int distance, double lambda, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    const bool kl = distance == NMFX_KL;
    if ((rc = gx_buffers(E, kl))) return rc;
    const int64_t np = E->np, kp = E->kp;
    float* xB = E->xf32;
    float* xG = E->xf32 + kp * np;
    float* xS = xG + kp * kp;
    ProfScope ps(E, "h_update");
    hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                       E->state, E->obj_hist);
    NMFX_HIP(hipGetLastError());
    if (!kl) {
        if ((rc = gx_launch<true, false>(E, GX_STORE, xG, kp, E->H, np, E->gx_d, np, 0, kp, np, kp, 1, nullptr, 0, nullptr))) return rc;
        const int64_t c4 = kp * np / 4;
        hipLaunchKernelGGL(gx_eu_update_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, E->stream, (const float*)E->H, (const float*)xB,
                           (const float*)E->gx_d, (float)lambda, E->H, c4, (const int*)&E->state->flag);
    } else {
        const int64_t cnt = kp * np;
        hipLaunchKernelGGL((gx_kl_update_kernel<true>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, E->stream, (const float*)E->H, (const float*)xB,
                           (const float*)xS, (float)lambda, E->H, kp, np, E->k, (const int*)&E->state->flag);
    }
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// objective partial of the current pair -> xf64[0] (the closing step: nmfx_mur_finish_a)
int nmfx_generic_mur_finish_a(nmfx_engine* E, int distance, int64_t j) {
    int rc;
    const bool kl = distance == NMFX_KL;
    if ((rc = gx_buffers(E, kl))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    const int64_t nblk = (mp / GX_T) * (np / GX_T);
    { ProfScope ps(E, "objective");
      if (!kl) rc = gx_launch<true, false>(E, GX_RESID, E->W[j & 1], kp, E->H, np, nullptr, 0, 0, mp, np, kp, 1, E->V, np, E->gx_part);
      else rc = gx_launch<true, false>(E, GX_KLQ, E->W[j & 1], kp, E->H, np, E->S, np, 0, mp, np, kp, 1, E->V, np, E->gx_part);
      if (rc) return rc; }
    return nmfx_launch_obj_reduce(E, nblk, E->gx_part);
}

int nmfx_preload_generic() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(gx_record_kernel)) == hipSuccess ? 0 : -1; }
