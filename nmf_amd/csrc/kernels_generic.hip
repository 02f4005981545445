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

enum { GX_STORE = 0, GX_RESID = 1, GX_KLQ = 2, GX_VAUX = 3 };

// r4: the multiplicative update of the Euclidean MUR loop inside the image kernel of the new factor (gxt_split_kernel): the element
// the kernel splits is  X * (sum of the numerator's slabs) / (Den + lam X + 1e-9)  (nmf/mur.py:29 / :45), written back as the new
// factor on the way -- one launch where the composed path had three (slab sum, element-wise update, image kernel)
struct GxUpd {
    const float* xold = nullptr;                                          // nullptr: plain images of M
    const float* num = nullptr; int nslab = 0; int64_t nstride = 0;      // numerator = sum of nslab slabs
    const float* den = nullptr;
    float lam = 0.f;
    float* xnew = nullptr;
};

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
// An operand that is contiguous along the contraction keeps that layout in LDS: [row][GX_LDK] with a row stride of 20 floats -- the
// two float4 go in as they are, and the MFMA fragment read of lane (x, q) -- element (row x, k = 4 u + q) -- hits bank
// (20 x + 4 u + q) mod 64: sixteen distinct multiples of four plus q, conflict free.  (A first form transposed such operands into
// the [k][row] plane with eight scalar stores per thread and chunk: V H^T ran at 82 TFLOP/s against 109 for the W H product.)
constexpr int GX_LDK = 20;
template <bool KCONTIG>
__device__ __forceinline__ void gx_store(float* __restrict__ plane, int tid, const float4& r0, const float4& r1) {
    if constexpr (KCONTIG) {
        float* q = plane + (tid >> 1) * GX_LDK + 8 * (tid & 1);
        *reinterpret_cast<float4*>(q) = r0;
        *reinterpret_cast<float4*>(q + 4) = r1;
    } else {
        float* q = plane + (tid >> 5) * GX_LD + 4 * (tid & 31);
        *reinterpret_cast<float4*>(q) = r0;
        *reinterpret_cast<float4*>(q + 8 * GX_LD) = r1;
    }
}

// C[z][i][j] = sum_{t in split z} A(i, t) B(t, j)     (i < 128 gridDim.y, j < 128 gridDim.x)
//   AK: A(i, t) = A[i * lda + t], else A[t * lda + i];   BK: B(t, j) = B[j * ldb + t], else B[t * ldb + j]
//   MODE GX_STORE: C stored.  GX_RESID: nothing stored, part[block] = 1/2 sum (X - C)^2 (utils.py:29).
//   GX_KLQ: Q = X / (C + 1e-9) stored (mur.py:25,41; not with C == nullptr) and, with part != nullptr, part[block] = the KL objective of the tile
//   (utils.py:23-26: x log(x / c) with inf / nan -> 0, - x + c).
//   GX_VAUX (KL-loss ADMM, ao_admm.py:90-95 / admm.py:312-315): with P = the product, X = V, C = dual_v, C2 = S:
//   v_aux = ((P - dual_v - 1) + sqrt((P - dual_v - 1)^2 + 4 V)) / 2;  dual_v += v_aux - P;  S = v_aux + dual_v  (P is not stored).
template <bool AK, bool BK, int MODE>
__global__ __launch_bounds__(256) void gx_gemm_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    int64_t cstride, int64_t K, const float* __restrict__ X, int64_t ldx, double* __restrict__ part, const int* __restrict__ flag,
    const int* __restrict__ flag2, float* __restrict__ C2 = nullptr)
{
    if (*flag || (flag2 && *flag2)) return;
    __shared__ __attribute__((aligned(16))) float lds[2][2][128 * GX_LDK];         // [buffer][A / B]: [k][GX_LD] or [row][GX_LDK] (>= 16 x 144 floats)
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
    // Two chunks in flight: a chunk is requested two iterations (~4000 cycles of MFMA work) before it is stored to LDS -- with one
    // (~2000 cycles) the loads of a loaded HBM system had not landed when the store needed them.  Branch-free prefetch (behind the
    // last chunk it fetches that chunk again): a conditionally assigned prefetch register is what hipcc parks in scratch.
    const int nch = (int)(kper / GX_KC);
    float4 a0, a1, b0, b1, a2, a3, b2, b3;
    gx_load<AK>(Ab, lda, kbeg, tid, a0, a1);
    gx_load<BK>(Bb, ldb, kbeg, tid, b0, b1);
    { const int64_t k1 = kbeg + (int64_t)(nch > 1 ? 1 : 0) * GX_KC;
      gx_load<AK>(Ab, lda, k1, tid, a2, a3);
      gx_load<BK>(Bb, ldb, k1, tid, b2, b3); }
    auto multiply = [&](const float* pa, const float* pb) {
#pragma unroll
        for (int u = 0; u < GX_KC / 4; ++u) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = AK ? pa[(wr + 16 * a + x) * GX_LDK + 4 * u + q] : pa[(4 * u + q) * GX_LD + wr + 16 * a + x];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = BK ? pb[(wc + 16 * b + x) * GX_LDK + 4 * u + q] : pb[(4 * u + q) * GX_LD + wc + 16 * b + x];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = GX_MFMA(av[a], bv[b], acc[a][b]);
        }
    };
    for (int ch = 0; ch < nch; ch += 2) {
        gx_store<AK>(lds[0][0], tid, a0, a1);
        gx_store<BK>(lds[0][1], tid, b0, b1);
        __syncthreads();                               // (two buffers: the chunk multiplied below is not the one written next)
        { const int64_t kn = kbeg + (int64_t)(ch + 2 < nch ? ch + 2 : nch - 1) * GX_KC;
          gx_load<AK>(Ab, lda, kn, tid, a0, a1);
          gx_load<BK>(Bb, ldb, kn, tid, b0, b1); }
        multiply(lds[0][0], lds[0][1]);
        if (ch + 1 < nch) {                            // (block-uniform)
            gx_store<AK>(lds[1][0], tid, a2, a3);
            gx_store<BK>(lds[1][1], tid, b2, b3);
            __syncthreads();
            { const int64_t kn = kbeg + (int64_t)(ch + 3 < nch ? ch + 3 : nch - 1) * GX_KC;
              gx_load<AK>(Ab, lda, kn, tid, a2, a3);
              gx_load<BK>(Bb, ldb, kn, tid, b2, b3); }
            multiply(lds[1][0], lds[1][1]);
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
    if (MODE == GX_VAUX) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int64_t row = i0 + wr + 16 * a + 4 * q + r, col = j0 + wc + 16 * b + x;
                    const float pv = acc[a][b][r], dv = C[row * ldc + col], t = (pv - dv) - 1.f;
                    const float va = 0.5f * (t + sqrtf(t * t + 4.f * X[row * ldx + col]));
                    const float dn = dv + va - pv;
                    C[row * ldc + col] = dn;
                    C2[row * ldc + col] = va + dn;
                }
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
                    if (C) C[(i0 + wr + 16 * a + 4 * q + r) * ldc + j0 + wc + 16 * b + x] = xv / (cv + 1e-9f);
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

// 1/2 sum (X - W H)^2 with the product AND the sum in float64 (v_mfma_f64_16x16x4_f64; W, H, X are float32 values, exact in
// f64): the objective of the device's iterate as the reference's float64 arithmetic would evaluate it.  The f32-evaluated
// objective of every iteration jitters by ~3e-9 relative (each element of W H carries its own f32 rounding, and they change with
// the iterate), which decides the stop rule once tol2 is below ~1e-6 of the objective; this kernel is the referee near the stop
// (nmfx_objective_f64).  Same tiling as gx_gemm_kernel<true, false, .>; C/D layout of the f64 MFMA: column = lane & 15,
// row = (lane >> 4) + 4 reg.
typedef double gx_f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gx_resid64_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, int64_t K, const float* __restrict__ X, int64_t ldx,
    double* __restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float lds[2][2][128 * GX_LDK];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * GX_T, j0 = (int64_t)blockIdx.x * GX_T;
    const float* Ab = A + i0 * lda;
    const float* Bb = B + j0;
    const int wr = 64 * (wave >> 1), wc = 64 * (wave & 1);
    gx_f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (gx_f64x4){0.0, 0.0, 0.0, 0.0};
    const int nch = (int)(K / GX_KC);
    float4 a0, a1, b0, b1;
    gx_load<true>(Ab, lda, 0, tid, a0, a1);
    gx_load<false>(Bb, ldb, 0, tid, b0, b1);
    for (int ch = 0; ch < nch; ++ch) {
        float* pa = lds[ch & 1][0];
        float* pb = lds[ch & 1][1];
        gx_store<true>(pa, tid, a0, a1);
        gx_store<false>(pb, tid, b0, b1);
        __syncthreads();
        { const int64_t kn = (int64_t)(ch + 1 < nch ? ch + 1 : ch) * GX_KC;
          gx_load<true>(Ab, lda, kn, tid, a0, a1);
          gx_load<false>(Bb, ldb, kn, tid, b0, b1); }
#pragma unroll
        for (int u = 0; u < GX_KC / 4; ++u) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = (double)pa[(wr + 16 * a + x) * GX_LDK + 4 * u + q];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = (double)pb[(4 * u + q) * GX_LD + wc + 16 * b + x];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    double tot = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double d = (double)X[(i0 + wr + 16 * a + q + 4 * r) * ldx + j0 + wc + 16 * b + x] - acc[a][b][r];
                tot += d * d;
            }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = 0.5 * (((red[0] + red[1]) + red[2]) + red[3]);
}

// sum of `n` doubles in a fixed order -> out[0]
__global__ __launch_bounds__(256) void gx_sum64_kernel(const double* __restrict__ part, int64_t n, double* __restrict__ out)
{
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += part[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
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
              int64_t M, int64_t N, int64_t K, int S, const float* X, int64_t ldx, double* part, const int* flag2 = nullptr, float* C2 = nullptr) {
    const dim3 grid((unsigned)(N / GX_T), (unsigned)(M / GX_T), (unsigned)S), block(256);
    const int* flag = &E->state->flag;
    if (mode == GX_STORE) hipLaunchKernelGGL((gx_gemm_kernel<AK, BK, GX_STORE>), grid, block, 0, E->stream, A, lda, B, ldb, C, ldc, cstride, K, X, ldx, part, flag, flag2);
    else if (mode == GX_RESID) hipLaunchKernelGGL((gx_gemm_kernel<AK, BK, GX_RESID>), grid, block, 0, E->stream, A, lda, B, ldb, C, ldc, cstride, K, X, ldx, part, flag, flag2);
    else if (mode == GX_VAUX) hipLaunchKernelGGL((gx_gemm_kernel<AK, BK, GX_VAUX>), grid, block, 0, E->stream, A, lda, B, ldb, C, ldc, cstride, K, X, ldx, part, flag, flag2, C2);
    else hipLaunchKernelGGL((gx_gemm_kernel<AK, BK, GX_KLQ>), grid, block, 0, E->stream, A, lda, B, ldb, C, ldc, cstride, K, X, ldx, part, flag, flag2);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

bool gx_fuse_update() { static const bool on = !(getenv("NMFX_GX_FUSE_UPDATE") && atoi(getenv("NMFX_GX_FUSE_UPDATE")) == 0); return on; }

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
// (flag2: a second skip flag for the product launches; the slab sum behind them is idempotent and runs regardless)
template <bool AK, bool BK>
int gx_split_product(nmfx_engine* E, const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t M, int64_t N, int64_t K,
                     int cap, const int* flag2 = nullptr) {
    int rc;
    const int S = gx_split(E, (M / GX_T) * (N / GX_T), K, cap);
    if (S == 1) return gx_launch<AK, BK>(E, GX_STORE, A, lda, B, ldb, out, N, 0, M, N, K, 1, nullptr, 0, nullptr, flag2);
    if ((rc = gx_launch<AK, BK>(E, GX_STORE, A, lda, B, ldb, E->gx_s, N, M * N, M, N, K, S, nullptr, 0, nullptr, flag2))) return rc;
    return nmfx_launch_sum_partials(E, E->gx_s, S, M * N, out);
}

// ---- split-bf16 form of the big products (r3) ------------------------------------------------------------------------------------
// The exact-f32 product kernel above runs at 0.7-0.8 of the f32 MFMA peak and that is what a k > 128 iteration costs (three V-sized
// products: 1.8 of 1.94 ms at 16384 x 8192, k = 256).  The same three products in the arithmetic of the tuned k <= 128 kernels --
// every operand x = hi + lo in bf16, a product = hi hi + lo hi + hi lo with f32 accumulation (kernels_bf16.hip, top: the three-term
// form MUR runs on) -- on v_mfma_f32_32x32x16_bf16.  Every operand is kept as PLANES that are contiguous along the contraction, so
// ONE kernel form serves all of them:
//     C[M][N] = sum_t A[i][t] B[j][t]          A = (Ahi, Alo) [M][K],  B = (Bhi, Blo) [N][K]
//   objective   1/2 ||V - W H||^2 :  A = W images [mp][kp],   B = H^T images [np][kp],  residual against the f32 V in the epilogue
//   V H^T                         :  A = V planes [mp][np],   B = H images [kp][np]
//   W^T V   ([kp][np])            :  A = W^T images [kp][mp], B = V^T planes [np][mp]
//   H H^T, W^T W                  :  A = B = H images / W^T images
// (V planes: built once per upload; factor images: rebuilt by gxt_split_kernel after each update, both orientations in one launch.)
// Since the later part of r3 this kernel carries the SHORT contractions only (objective, KL quotient: contraction over the factor index,
// row-major images of W and H^T); every product with a long contraction runs on gxt_gemm_kernel further down, on tiled planes.
// Block = 128 x 128 outputs, 4 waves x (2 x 2 tiles of 32 x 32), contraction in chunks of 64: four planes [128 rows][64 bf16] in
// LDS with a row stride of 144 bytes -- an odd multiple of 16, so the 16 rows of a ds_read_b128 lane group (MI355X_MICROARCH.md,
// LDS table) fall on 16 distinct bank quadruples, and the 8 lanes of a ds_write_b128 group write one contiguous row --, the
// next chunk prefetched into registers (16 x 16 bytes per thread) while the current one is multiplied; 72 KiB of LDS: two
// blocks per CU, so one block's barriers and load latencies are covered by the other's MFMAs.
typedef __bf16 gxb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gxb_f32x16 __attribute__((ext_vector_type(16)));
union GxbFrag { uint4 u; gxb_bf16x8 v; };
#define GXB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).v, (b).v, (c), 0, 0, 0)
constexpr int GXB_KC = 64, GXB_LDB = 144, GXB_PLANE = 128 * GXB_LDB;      // bytes
constexpr int GXB_SHM = 4 * GXB_PLANE + 64;

template <int MODE, int TERMS = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gxb_gemm_kernel(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, int64_t lda,
    const unsigned short* __restrict__ Bhi, const unsigned short* __restrict__ Blo, int64_t ldb,
    float* __restrict__ C, int64_t ldc, int64_t cstride, int64_t K, const float* __restrict__ X, int64_t ldx,
    double* __restrict__ part, const int* __restrict__ flag, const int* __restrict__ flag2,
    unsigned short* __restrict__ Qhi = nullptr, unsigned short* __restrict__ Qlo = nullptr)
{
    // MODE GX_KLQ (MUR with the KL divergence, mur.py:25,41): the quotient Q = X / (C + 1e-9) leaves as bf16 hi / lo PLANES
    // [M][N] (row stride ldc) -- the operand layout of the product that consumes it -- and, with part != nullptr, part[block] =
    // the KL objective of the tile (utils.py:23-26), both exactly as gx_gemm_kernel<.., GX_KLQ> forms them
    if (*flag || (flag2 && *flag2)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char gxb_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n31 = lane & 31, b = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.y * GX_T, j0 = (int64_t)blockIdx.x * GX_T;
    const int64_t kper = K / gridDim.z, kbeg = kper * blockIdx.z;
    const int wr = 64 * (wave >> 1), wc = 64 * (wave & 1);
    // staging: piece p = tid + 256 i (i < 4) of a plane = 16 bytes: row p >> 3, bytes 16 (p & 7) of the row's 128-byte chunk
    const int prow = tid >> 3, pcol = tid & 7;
    const unsigned short* sAh = Ahi + (i0 + prow) * lda + kbeg + 8 * pcol;
    const unsigned short* sAl = Alo + (i0 + prow) * lda + kbeg + 8 * pcol;
    const unsigned short* sBh = Bhi + (j0 + prow) * ldb + kbeg + 8 * pcol;
    const unsigned short* sBl = Blo + (j0 + prow) * ldb + kbeg + 8 * pcol;
    unsigned char* dst = gxb_smem + prow * GXB_LDB + 16 * pcol;
    // (named registers, no arrays behind a lambda: hipcc parks such a staging array in scratch)
    uint4 ah0, ah1, ah2, ah3, al0, al1, al2, al3, bh0, bh1, bh2, bh3, bl0, bl1, bl2, bl3;
#define GXB_LD4(p, ld_, koff, r0_, r1_, r2_, r3_) do { \
        r0_ = *reinterpret_cast<const uint4*>((p) + (koff)); r1_ = *reinterpret_cast<const uint4*>((p) + 32 * (ld_) + (koff)); \
        r2_ = *reinterpret_cast<const uint4*>((p) + 64 * (ld_) + (koff)); r3_ = *reinterpret_cast<const uint4*>((p) + 96 * (ld_) + (koff)); } while (0)
#define GXB_FETCH(koff) do { GXB_LD4(sAh, lda, koff, ah0, ah1, ah2, ah3); GXB_LD4(sAl, lda, koff, al0, al1, al2, al3); \
        GXB_LD4(sBh, ldb, koff, bh0, bh1, bh2, bh3); GXB_LD4(sBl, ldb, koff, bl0, bl1, bl2, bl3); } while (0)
#define GXB_ST4(pl, r0_, r1_, r2_, r3_) do { \
        *reinterpret_cast<uint4*>(dst + (pl) * GXB_PLANE) = r0_; *reinterpret_cast<uint4*>(dst + (pl) * GXB_PLANE + 32 * GXB_LDB) = r1_; \
        *reinterpret_cast<uint4*>(dst + (pl) * GXB_PLANE + 64 * GXB_LDB) = r2_; *reinterpret_cast<uint4*>(dst + (pl) * GXB_PLANE + 96 * GXB_LDB) = r3_; } while (0)
    gxb_f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
    const unsigned char* fa = gxb_smem + (wr + n31) * GXB_LDB + 16 * b;                    // + 32 rows per tile, + 32 bytes per k-step, + GXB_PLANE: lo
    const unsigned char* fb = gxb_smem + 2 * GXB_PLANE + (wc + n31) * GXB_LDB + 16 * b;
    const int nch = (int)(kper / GXB_KC);
    auto multiply = [&]() {
#pragma unroll
        for (int ks = 0; ks < GXB_KC / 16; ++ks) {
            GxbFrag ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t].u = *reinterpret_cast<const uint4*>(fa + t * 32 * GXB_LDB + 32 * ks);
                al[t].u = *reinterpret_cast<const uint4*>(fa + t * 32 * GXB_LDB + 32 * ks + GXB_PLANE);
                bh[t].u = *reinterpret_cast<const uint4*>(fb + t * 32 * GXB_LDB + 32 * ks);
                bl[t].u = *reinterpret_cast<const uint4*>(fb + t * 32 * GXB_LDB + 32 * ks + GXB_PLANE);
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    acc[ti][tj] = GXB_MFMA(ah[ti], bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(al[ti], bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(ah[ti], bl[tj], acc[ti][tj]);
                    if (TERMS >= 4) acc[ti][tj] = GXB_MFMA(al[ti], bl[tj], acc[ti][tj]);      // (ADMM: its systems carry the caller's fixed rho, kernels_bf16.hip top)
                }
        }
    };
    GXB_FETCH(0);
    for (int ch = 0; ch + 1 < nch; ++ch) {
        __syncthreads();                               // the previous chunk has been multiplied
        GXB_ST4(0, ah0, ah1, ah2, ah3); GXB_ST4(1, al0, al1, al2, al3); GXB_ST4(2, bh0, bh1, bh2, bh3); GXB_ST4(3, bl0, bl1, bl2, bl3);
        __syncthreads();
        { const int64_t koff = (int64_t)(ch + 1) * GXB_KC; GXB_FETCH(koff); }      // the next chunk, while this one is multiplied
        multiply();
    }
    __syncthreads();                                   // the last chunk: nothing left to prefetch but the epilogue's operand
    GXB_ST4(0, ah0, ah1, ah2, ah3); GXB_ST4(1, al0, al1, al2, al3); GXB_ST4(2, bh0, bh1, bh2, bh3); GXB_ST4(3, bl0, bl1, bl2, bl3);
    __syncthreads();
    // acc[ti][tj][r] = C(i0 + wr + 32 ti + (r & 3) + 8 (r >> 2) + 4 b, j0 + wc + 32 tj + n31)
    float xv[MODE != GX_STORE ? 64 : 1];
    if constexpr (MODE != GX_STORE) {                  // the epilogue's X tile is requested here and lands under the last chunk's MFMAs
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xv[(2 * ti + tj) * 16 + r] = X[(i0 + wr + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * b) * ldx + j0 + wc + 32 * tj + n31];
    }
    multiply();
#undef GXB_LD4
#undef GXB_FETCH
#undef GXB_ST4
    if (MODE == GX_STORE) {
        float* Cz = C + (int64_t)blockIdx.z * cstride;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    Cz[(i0 + wr + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * b) * ldc + j0 + wc + 32 * tj + n31] = acc[ti][tj][r];
        return;
    }
    double tot = 0.0;                                  // GX_RESID: 1/2 sum (X - C)^2 of the tile (utils.py:29)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x1 = xv[MODE != GX_STORE ? (2 * ti + tj) * 16 + r : 0], cv = acc[ti][tj][r];
                float d2;
                if (MODE == GX_RESID) { const float d = x1 - cv; d2 = d * d; }
                else {
                    // (v_rcp_f32 / v_log_f32 as in the tuned MUR-KL kernels, kernels_bf16.hip: the inf / nan cases of utils.py:24 come
                    //  out the same and are zeroed the same way)
                    const float qv = x1 * __builtin_amdgcn_rcpf(cv + 1e-9f);
                    unsigned hi, lo;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(qv), "v"(0.f));
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(qv - __uint_as_float(hi << 16)), "v"(0.f));
                    // two adjacent lanes hold adjacent columns: the even lane stores the pair of the hi plane, the odd lane the pair
                    // of the lo plane -- one 4-byte store per lane instead of two 2-byte stores
                    const unsigned other = (unsigned)__shfl_xor((int)((n31 & 1) ? hi : lo), 1, 64);
                    // (the planes in the TILED image of gxt_gemm_kernel, contraction along the columns: row I, columns J, J + 1)
                    const int64_t I = i0 + wr + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * b, J = j0 + wc + 32 * tj + (n31 & ~1);
                    const int rr = (int)(I & 127);
                    const int64_t at = ((I >> 7) * (ldc / 32) + (J >> 5)) * 4096 + rr * 32 + 8 * ((int)((J & 31) >> 3) ^ ((rr >> 2) & 3)) + (J & 7);
                    if (n31 & 1) *reinterpret_cast<unsigned*>(Qlo + at) = (other & 0xffffu) | (lo << 16);
                    else *reinterpret_cast<unsigned*>(Qhi + at) = (hi & 0xffffu) | (other << 16);
                    d2 = 0.f;
                    if (part) {
                        float t = x1 * (__builtin_amdgcn_logf(x1 * __builtin_amdgcn_rcpf(cv)) * 0.69314718055994531f);
                        t = (t != t || t == __builtin_inff() || t == -__builtin_inff()) ? 0.f : t;
                        d2 = (t - x1) + cv;
                    }
                }
                if ((r & 3) == 0) s0 += d2; else if ((r & 3) == 1) s1 += d2; else if ((r & 3) == 2) s2 += d2; else s3 += d2;
            }
            tot += (double)((s0 + s1) + (s2 + s3));
        }
    if (MODE == GX_KLQ && !part) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    __syncthreads();
    double* red = reinterpret_cast<double*>(gxb_smem + 4 * GXB_PLANE);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (MODE == GX_RESID ? 0.5 : 1.0) * (((red[0] + red[1]) + red[2]) + red[3]);
}

bool gxr_on();
int gxr_launch(nmfx_engine* E, int mode, const unsigned short* Ahi, const unsigned short* Alo, const unsigned short* Bhi, const unsigned short* Blo,
               int64_t ldq, int64_t M, int64_t N, int64_t K, const float* X, int64_t ldx, double* part, const int* flag2,
               unsigned short* Qhi, unsigned short* Qlo);
int gxb_launch(nmfx_engine* E, int mode, const unsigned short* Ahi, const unsigned short* Alo, int64_t lda, const unsigned short* Bhi,
               const unsigned short* Blo, int64_t ldb, float* C, int64_t ldc, int64_t cstride, int64_t M, int64_t N, int64_t K, int S,
               const float* X, int64_t ldx, double* part, const int* flag2 = nullptr, unsigned short* Qhi = nullptr, unsigned short* Qlo = nullptr) {
    // (GX_RESID / GX_KLQ only: the long contractions -- every GX_STORE product -- run on gxt_gemm_kernel below)
    if (gxr_on() && mode != GX_STORE && S == 1) return gxr_launch(E, mode, Ahi, Alo, Bhi, Blo, ldc, M, N, K, X, ldx, part, flag2, Qhi, Qlo);
    if (mode == GX_STORE || M % GX_T || N % GX_T || K % ((int64_t)S * GXB_KC)) { E->err = "gxb_launch: shape / mode"; return NMFX_E_ARG; }
    const dim3 grid((unsigned)(N / GX_T), (unsigned)(M / GX_T), (unsigned)S), block(256);
    const int* flag = &E->state->flag;
    int rc;
    if (mode == GX_KLQ) {                              // (ldc = the contraction length of the product that consumes the Q planes)
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gxb_gemm_kernel<GX_KLQ>), GXB_SHM))) return rc;
        hipLaunchKernelGGL((gxb_gemm_kernel<GX_KLQ>), grid, block, GXB_SHM, E->stream, Ahi, Alo, lda, Bhi, Blo, ldb, C, ldc, cstride, K, X, ldx, part, flag, flag2, Qhi, Qlo);
    } else {
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gxb_gemm_kernel<GX_RESID>), GXB_SHM))) return rc;
        hipLaunchKernelGGL((gxb_gemm_kernel<GX_RESID>), grid, block, GXB_SHM, E->stream, Ahi, Alo, lda, Bhi, Blo, ldb, C, ldc, cstride, K, X, ldx, part, flag, flag2);
    }
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- the products with a long contraction (V H^T, W^T V, Q H^T, W^T Q', the Gram matrices) on TILED planes filled by LDS-DMA (r3) ------
// gxb_gemm_kernel stages its operands through registers: per CU and chunk the ds_write_b128 path (~79 B/clk), the texture path and
// the matrix pipe carry loads within 30 % of each other (matrix pipe 0.47 busy, V H^T 217 us at 16384 x 8192, k = 256).  Here the planes
// are stored as TILES in the image the LDS wants --
//     tile (row tile rt of 128, chunk c of 32) = 8 KiB contiguous at ((rt (K / 32) + c) 4096) elements:
//     element (row rr, k) at rr 32 + 8 ((k / 8) ^ ((rr >> 2) & 3)) + k % 8
// (the 16 rows of a ds_read_b128 lane group then fall on 16 distinct bank quadruples: 4 (rr & 3) + (chunk ^ (rr >> 2 & 3))) -- so a
// stage is filled by plain linear copies global -> LDS (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPRs, no
// ds_write), three stages of a 256 x 128 block tile (48 KiB each: two chunks in flight), eight waves x (2 x 2 tiles of 32 x 32), one
// barrier per chunk, counted vmcnt; 98 VGPRs.  C[M][N] = sum_t A[i][t] B[j][t], stored (V H^T 175-181 us on the same boxes = 1.2 PFLOP/s
// of executed bf16 MFMA work = 0.48 of the dense peak).  The short contractions stay on gxb_gemm_kernel: built on this kernel too
// (X tile requested in front of the first chunk, one 256 x 128 tile per block) they were no faster -- eight chunks per block do not
// amortise the head of the DMA ring with one block per CU, and register loads of the next tile's X cannot ride in the same
// in-order vmcnt queue as the DMA stream without serialising with it.
__device__ __forceinline__ unsigned gxt_lds_off(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void gxt_dma(unsigned long long base, unsigned dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(dst), "v"(voff) : "memory");
}
__device__ __forceinline__ void gxt_dma_nt(unsigned long long base, unsigned dst, unsigned voff) {      // the once-read operand: non-temporal
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(dst), "v"(voff) : "memory");
}
constexpr int GXT_STAGE = 6 * 8192, GXT_SHM = 3 * GXT_STAGE;

int gx_stagger() { static const int v = getenv("NMFX_GX_STAGGER") ? atoi(getenv("NMFX_GX_STAGGER")) : 1; return v; }

template <int TERMS>
__global__ __launch_bounds__(512) void gxt_gemm_kernel(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, const unsigned short* __restrict__ Bhi,
    const unsigned short* __restrict__ Blo, float* __restrict__ C, int64_t ldc, int64_t cstride, int64_t M, int64_t K,
    const int* __restrict__ flag, const int* __restrict__ flag2, int stagger)
{
    if (*flag || (flag2 && *flag2)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char gxt_smem[];
    const int tid = threadIdx.x, lane = tid & 63, n31 = lane & 31, b = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t kch = K / 32, nch = kch / gridDim.z, c0 = nch * blockIdx.z;
    const int64_t mt = M / 128;
    // r4: the column tiles of one row block on ONE XCD, next to each other in time.  Blocks are dealt to the 8 XCDs round-robin in
    // linear order (x fastest), so with (x, y) = blockIdx the nx column tiles of a row block -- which stream the SAME A planes (the
    // V planes of V H^T: 512 MiB) -- sat on nx different XCDs and each fetched them from HBM: 923 MB per launch against 562 MB of
    // operands at k = 256 (PMC, profiles/r04_pmc_summary.json), four times V at k = 512.  Within groups of 8 row blocks the linear
    // index is re-read as (column tile, row block) = (L' / 8, L' % 8): L' and L' + 8 -- the same XCD, dispatched in the same round --
    // are neighbouring column tiles of one row block, and the second reader finds the planes in that XCD's L2.
    int bxi = blockIdx.x, byi = blockIdx.y;
    if (gridDim.x > 1 && gridDim.y >= 8) {
        const int nx = gridDim.x, L = byi * nx + bxi, grp = L / (8 * nx), Lp = L % (8 * nx);
        if (8 * (grp + 1) <= (int)gridDim.y) { byi = 8 * grp + (Lp & 7); bxi = Lp >> 3; }      // (a last partial group keeps its order)
    }
    const int64_t rt0 = 2 * (int64_t)byi, rt1 = (rt0 + 1 < mt) ? rt0 + 1 : rt0;          // (an odd number of row tiles: the last block loads its tile twice)
    const bool second = rt0 + 1 < mt;
    // DMA: per chunk six units of 8 KiB (A hi t0, A hi t1, A lo t0, A lo t1, B hi, B lo); wave w copies piece w (1 KiB) of each
    unsigned long long src[6];
    src[0] = (unsigned long long)(Ahi + (rt0 * kch + c0) * 4096) + wave * 1024ull;
    src[1] = (unsigned long long)(Ahi + (rt1 * kch + c0) * 4096) + wave * 1024ull;
    src[2] = (unsigned long long)(Alo + (rt0 * kch + c0) * 4096) + wave * 1024ull;
    src[3] = (unsigned long long)(Alo + (rt1 * kch + c0) * 4096) + wave * 1024ull;
    src[4] = (unsigned long long)(Bhi + ((int64_t)bxi * kch + c0) * 4096) + wave * 1024ull;
    src[5] = (unsigned long long)(Blo + ((int64_t)bxi * kch + c0) * 4096) + wave * 1024ull;
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(gxt_lds_off(gxt_smem));
    const unsigned voff = (unsigned)lane * 16u;
    auto issue = [&](int64_t c, int stage) {
        const unsigned dst = smem0 + stage * GXT_STAGE + wave * 1024;
        const unsigned long long adv = (unsigned long long)c * 8192ull;
#pragma unroll
        for (int u = 0; u < 6; ++u) gxt_dma(src[u] + adv, dst + u * 8192, voff);
    };
    // fragment offsets inside a stage: wave = (row group of 64 of the 256, column group of 64 of the 128)
    const int wr = 64 * (wave >> 1), wc = 64 * (wave & 1);
    int aoff[2][2], boff[2][2];                       // [tile][k-step]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = (wr & 127) + 32 * t + n31, rb = wc + 32 * t + n31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            aoff[t][ks] = (wr >> 7) * 8192 + ra * 64 + 16 * ((2 * ks + b) ^ ((ra >> 2) & 3));        // + 16384: lo
            boff[t][ks] = 4 * 8192 + rb * 64 + 16 * ((2 * ks + b) ^ ((rb >> 2) & 3));                // + 8192: lo
        }
    }
    gxb_f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
    issue(0, 0);
    if (nch > 1) issue(1, 1);
    int stage = 0;
    // il: this wave's six DMA pieces of chunk c + 2 go out BETWEEN the MFMA groups of chunk c (one piece behind each of the first six
    // groups of three MFMAs, pinned by scheduling barriers), so that their issue (60-180 cycles each) runs in the shadow of the
    // wave's own matrix work instead of in front of it
    auto multiply = [&](bool il, int64_t cn, int nstage) {
        const unsigned char* st = gxt_smem + stage * GXT_STAGE;
        const unsigned dst = smem0 + nstage * GXT_STAGE + wave * 1024;
        const unsigned long long adv = (unsigned long long)cn * 8192ull;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            GxbFrag ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t].u = *reinterpret_cast<const uint4*>(st + aoff[t][ks]);
                al[t].u = *reinterpret_cast<const uint4*>(st + aoff[t][ks] + 16384);
                bh[t].u = *reinterpret_cast<const uint4*>(st + boff[t][ks]);
                bl[t].u = *reinterpret_cast<const uint4*>(st + boff[t][ks] + 8192);
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    acc[ti][tj] = GXB_MFMA(ah[ti], bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(al[ti], bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(ah[ti], bl[tj], acc[ti][tj]);
                    if (TERMS >= 4) acc[ti][tj] = GXB_MFMA(al[ti], bl[tj], acc[ti][tj]);
                    const int g = 4 * ks + 2 * ti + tj;
                    if (g < 6) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (il) gxt_dma(src[g] + adv, dst + g * 8192, voff);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
    };
    // r4: the two waves of a SIMD (w and w + 4) run the same program between the same barriers and would issue their six DMA
    // pieces at the same time, the SIMD's matrix pipe idle meanwhile, then queue for it together.  stagger 1: waves 0-3 request chunk
    // c + 2 BEFORE their MFMAs of chunk c, waves 4-7 BEHIND them (the target stage held chunk c - 1, which every wave has left at this
    // iteration's barrier; at the next wait chunk c + 2 is the only younger request either way).  stagger 16: interleaved (above).
    const bool inl = stagger & 16;
    const bool early = (stagger & 1) ? wave < 4 : true;
    for (int64_t c = 0; c < nch; ++c) {
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // chunk c has landed everywhere; everybody is done with the stage chunk c + 2 goes to
        const int nstage = stage >= 1 ? stage - 1 : 2; // (c + 2) % 3
        const bool more = c + 2 < nch;
        if (!inl && early && more) issue(c + 2, nstage);
        multiply(inl && more, c + 2, nstage);          // (ONE instance between the two conditional requests: the accumulators stay in place)
        if (!inl && !early && more) issue(c + 2, nstage);
        stage = (stage == 2) ? 0 : stage + 1;
    }
    if (wr >= 128 && !second) return;                  // (the duplicated tile of an odd last block)
    float* Cz = C + (int64_t)blockIdx.z * cstride;
    const int64_t i0 = (int64_t)byi * 256, j0 = (int64_t)bxi * 128;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
                Cz[(i0 + wr + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * b) * ldc + j0 + wc + 32 * tj + n31] = acc[ti][tj][r];
}

// ---- r4: 256 x 256 block tiles for the long contractions -------------------------------------------------------------------------
// Measured on gxr_kernel: two more lines per lane and tile through the vector-memory path (an L2 prefetch of the next X tile) cost as
// much as the 64 X loads themselves -- the CU's texture path, at ~36 bytes per clock for the 48 KiB of operand planes of every
// 256 x 128 x 32 chunk, sets the pace of these kernels, not the matrix pipe (48 KiB / 36 = 1365 of the 1536 cycles a SIMD's MFMAs take).
// A 256 x 256 tile needs 64 KiB for twice the MFMA work: 0.67 of the bytes.  Two stages of 64 KiB (A: hi / lo of two row tiles, B:
// hi / lo of two column tiles), the request of chunk c + 1 behind the barrier of chunk c; eight waves x (4 x 2 tiles of 32 x 32),
// 164 VGPRs.  Same box, 16384 x 8192: V H^T k = 256 165.8 -> 154.8 us, k = 512 309.6 -> 286.6; W^T V 170.3 -> 164.0, 315.2 -> 298.0
// (the numerator's four slabs instead of two cost the update launch 8 us of that at k = 256).  A five-slot ring of half stages (one
// and a half chunks in flight, all 160 KiB) was no faster than the two whole stages.  NMFX_GXT2=0: the 256 x 128 kernel everywhere.
constexpr int GXT2_STAGE = 8 * 8192, GXT2_SHM = 2 * GXT2_STAGE;
template <int TERMS>
__global__ __launch_bounds__(512) void gxt2_gemm_kernel(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, const unsigned short* __restrict__ Bhi,
    const unsigned short* __restrict__ Blo, float* __restrict__ C, int64_t ldc, int64_t cstride, int64_t K,
    const int* __restrict__ flag, const int* __restrict__ flag2, int nt)
{
    // nt: bit 0 / 1 = the A / B planes are read once per launch (the V planes of V H^T, the V^T planes of W^T V): non-temporal
    // requests, so that they do not push the other operand -- the factor images every block re-reads -- out of L2
    if (*flag || (flag2 && *flag2)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char gxt_smem[];
    const int tid = threadIdx.x, lane = tid & 63, n31 = lane & 31, b = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t kch = K / 32, nch = kch / gridDim.z, c0 = nch * blockIdx.z;
    const int64_t at0 = 2 * (int64_t)blockIdx.y, bt0 = 2 * (int64_t)blockIdx.x;
    unsigned long long src[8];
    src[0] = (unsigned long long)(Ahi + (at0 * kch + c0) * 4096) + wave * 1024ull;
    src[1] = (unsigned long long)(Ahi + ((at0 + 1) * kch + c0) * 4096) + wave * 1024ull;
    src[2] = (unsigned long long)(Alo + (at0 * kch + c0) * 4096) + wave * 1024ull;
    src[3] = (unsigned long long)(Alo + ((at0 + 1) * kch + c0) * 4096) + wave * 1024ull;
    src[4] = (unsigned long long)(Bhi + (bt0 * kch + c0) * 4096) + wave * 1024ull;
    src[5] = (unsigned long long)(Bhi + ((bt0 + 1) * kch + c0) * 4096) + wave * 1024ull;
    src[6] = (unsigned long long)(Blo + (bt0 * kch + c0) * 4096) + wave * 1024ull;
    src[7] = (unsigned long long)(Blo + ((bt0 + 1) * kch + c0) * 4096) + wave * 1024ull;
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(gxt_lds_off(gxt_smem));
    const unsigned voff = (unsigned)lane * 16u;
    auto issue = [&](int64_t c, int stage) {
        const unsigned dst = smem0 + stage * GXT2_STAGE + wave * 1024;
        const unsigned long long adv = (unsigned long long)c * 8192ull;
        if (nt & 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) gxt_dma_nt(src[u] + adv, dst + u * 8192, voff);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) gxt_dma(src[u] + adv, dst + u * 8192, voff);
        }
        if (nt & 2) {
#pragma unroll
            for (int u = 4; u < 8; ++u) gxt_dma_nt(src[u] + adv, dst + u * 8192, voff);
        } else {
#pragma unroll
            for (int u = 4; u < 8; ++u) gxt_dma(src[u] + adv, dst + u * 8192, voff);
        }
    };
    const int wr = 128 * (wave >> 2), wc = 64 * (wave & 3);
    int aoff[4][2], boff[2][2];                       // [tile][k-step]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ra = 32 * t + n31;
            aoff[t][ks] = (wr >> 7) * 8192 + ra * 64 + 16 * ((2 * ks + b) ^ ((ra >> 2) & 3));            // + 16384: lo
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int rb = wc + 32 * t + n31, rr = rb & 127;
            boff[t][ks] = 4 * 8192 + (rb >> 7) * 8192 + rr * 64 + 16 * ((2 * ks + b) ^ ((rr >> 2) & 3));    // + 16384: lo
        }
    }
    gxb_f32x16 acc[4][2];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
    issue(0, 0);
    for (int64_t c = 0; c < nch; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // chunk c has landed everywhere; everybody has left the other stage
        if (c + 1 < nch) issue(c + 1, (int)((c + 1) & 1));
        const unsigned char* sta = gxt_smem + (c & 1) * GXT2_STAGE;
        const unsigned char* stb = sta;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            GxbFrag bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t].u = *reinterpret_cast<const uint4*>(stb + boff[t][ks]);
                bl[t].u = *reinterpret_cast<const uint4*>(stb + boff[t][ks] + 16384);
            }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                GxbFrag ah, al;
                ah.u = *reinterpret_cast<const uint4*>(sta + aoff[ti][ks]);
                al.u = *reinterpret_cast<const uint4*>(sta + aoff[ti][ks] + 16384);
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    acc[ti][tj] = GXB_MFMA(ah, bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(al, bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(ah, bl[tj], acc[ti][tj]);
                    if (TERMS >= 4) acc[ti][tj] = GXB_MFMA(al, bl[tj], acc[ti][tj]);
                }
            }
        }
    }
    float* Cz = C + (int64_t)blockIdx.z * cstride;
    const int64_t i0 = (int64_t)blockIdx.y * 256 + wr, j0 = (int64_t)blockIdx.x * 256 + wc;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
                Cz[(i0 + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * b) * ldc + j0 + 32 * tj + n31] = acc[ti][tj][r];
}

// ---- the SHORT contractions (objective, KL quotient: contraction over the factor index, V-sized output that never leaves the chip as
// f32) as a PERSISTENT form of the kernel above (r4) ---------------------------------------------------------------------------------
// gxb_gemm_kernel spent a block per 128 x 128 output tile: with eight 32-deep chunks per tile (k = 256) the head of the operand
// pipeline, the X tile's round trip and the epilogue were paid once per tile and two blocks per CU had to cover each other's.  Here
// one 512-thread block per CU walks a run of 256 x 128 output tiles (row-block major, the runs of one XCD next to each other, so
// that its L2 holds the A planes of its 8 row blocks) and the three-stage LDS-DMA ring of gxt_gemm_kernel runs THROUGH the tile
// boundaries: while a tile's epilogue executes, the first two chunks of the next tile are already in flight.  Both factors are read
// as tiled images (W: rows m; H^T: rows n; contraction kp).  The epilogue's X tile (V in f32, row-major) arrives in the register
// image of the accumulators, requested in four quarters behind the tile's first four chunks.  Those loads are inline asm and waited
// for by the loop's own counted vmcnt (hipcc does not count the LDS-DMA requests; loads return in order; the quotient's stores are
// never counted, which only makes a wait stricter).
//   GX_RESID: part[block] = 1/2 sum (X - C)^2 over the block's tiles (utils.py:29)
//   GX_KLQ:   Q = X / (C + 1e-9) as bf16 hi / lo planes in the TILED image of the product that consumes them (contraction along
//             the columns), and with want_obj the KL objective (utils.py:23-26) into part[block]
template <int MODE, bool OBJ>
__global__ __launch_bounds__(512) void gxr_kernel(
    const unsigned short* __restrict__ Ahi, const unsigned short* __restrict__ Alo, const unsigned short* __restrict__ Bhi,
    const unsigned short* __restrict__ Blo, int64_t M, int64_t N, int64_t K, const float* __restrict__ X, int64_t ldx,
    double* __restrict__ part, int64_t npart, const int* __restrict__ flag, const int* __restrict__ flag2,
    unsigned short* __restrict__ Qhi, unsigned short* __restrict__ Qlo, int64_t ldq, int stagger)
{
    if (*flag || (flag2 && *flag2)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char gxt_smem[];
    const int tid = threadIdx.x, lane = tid & 63, n31 = lane & 31, b = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = (int)(K / 32);
    const int64_t mt = M / 128, nrb = (mt + 1) / 2, nct = N / 128, T = nrb * nct;
    const int nb = (int)gridDim.x;
    int vb = (int)blockIdx.x;
    if ((nb & 7) == 0) vb = (vb & 7) * (nb >> 3) + (vb >> 3);          // (blocks b, b + 8, .. share an XCD: give them neighbouring runs)
    const int64_t t0 = T * vb / nb, t1 = T * (vb + 1) / nb;
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(gxt_lds_off(gxt_smem));
    const unsigned voff = (unsigned)lane * 16u;
    // the issue pointer: (tile, chunk) of the next request, two chunks ahead of the one being multiplied
    int64_t q_rb = t0 / nct, q_ct = t0 % nct;
    int q_c = 0, q_stage = 0;
    int64_t q_left = (t1 - t0) * nch;
    unsigned long long q_a0, q_a1, q_b;                // byte offsets of the next chunk in the A planes (two row tiles) and the B planes
    auto q_tile = [&]() {
        const int64_t rt0 = 2 * q_rb, rt1 = (rt0 + 1 < mt) ? rt0 + 1 : rt0;
        q_a0 = (unsigned long long)(rt0 * nch * 8192 + wave * 1024); q_a1 = (unsigned long long)(rt1 * nch * 8192 + wave * 1024);
        q_b = (unsigned long long)(q_ct * nch * 8192 + wave * 1024);
    };
    q_tile();
    auto issue = [&]() {
        const unsigned dst = smem0 + q_stage * GXT_STAGE + wave * 1024;
        gxt_dma((unsigned long long)Ahi + q_a0, dst, voff);
        gxt_dma((unsigned long long)Ahi + q_a1, dst + 8192, voff);
        gxt_dma((unsigned long long)Alo + q_a0, dst + 2 * 8192, voff);
        gxt_dma((unsigned long long)Alo + q_a1, dst + 3 * 8192, voff);
        gxt_dma((unsigned long long)Bhi + q_b, dst + 4 * 8192, voff);
        gxt_dma((unsigned long long)Blo + q_b, dst + 5 * 8192, voff);
        q_stage = (q_stage == 2) ? 0 : q_stage + 1;
        --q_left;
        q_a0 += 8192; q_a1 += 8192; q_b += 8192;
        if (++q_c == nch) { q_c = 0; if (++q_ct == nct) { q_ct = 0; ++q_rb; } q_tile(); }
    };
    const int wr = 64 * (wave >> 1), wc = 64 * (wave & 1);
    int aoff[2][2], boff[2][2];                       // [tile][k-step]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = (wr & 127) + 32 * t + n31, rb = wc + 32 * t + n31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            aoff[t][ks] = (wr >> 7) * 8192 + ra * 64 + 16 * ((2 * ks + b) ^ ((ra >> 2) & 3));        // + 16384: lo
            boff[t][ks] = 4 * 8192 + rb * 64 + 16 * ((2 * ks + b) ^ ((rb >> 2) & 3));                // + 8192: lo
        }
    }
    gxb_f32x16 acc[2][2];
    float xv[64];                                      // [(2 ti + tj) 16 + r], the register image of acc
    int stage = 0;
    auto multiply = [&]() {
        const unsigned char* st = gxt_smem + stage * GXT_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            GxbFrag ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t].u = *reinterpret_cast<const uint4*>(st + aoff[t][ks]);
                al[t].u = *reinterpret_cast<const uint4*>(st + aoff[t][ks] + 16384);
                bh[t].u = *reinterpret_cast<const uint4*>(st + boff[t][ks]);
                bl[t].u = *reinterpret_cast<const uint4*>(st + boff[t][ks] + 8192);
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    acc[ti][tj] = GXB_MFMA(ah[ti], bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(al[ti], bh[tj], acc[ti][tj]);
                    acc[ti][tj] = GXB_MFMA(ah[ti], bl[tj], acc[ti][tj]);
                }
        }
        stage = (stage == 2) ? 0 : stage + 1;
    };
    double tot = 0.0;
    gxb_f32x16 pacc[2][2];                             // GX_KLQ: the accumulators of the tile before, until its quotient has left
    bool pending = false, p_live = false, act_prev = false;
    unsigned short *p_q0 = nullptr, *p_q1 = nullptr;
    // quarter h of the quotient epilogue: ti = h >> 1, rows r = 8 (h & 1) .. + 7, both column tiles (v_rcp_f32 / v_log_f32 as in the
    // tuned MUR-KL kernels, kernels_bf16.hip: the inf / nan cases of utils.py:24 come out the same and are zeroed the same way)
    auto klq_piece = [&](int pc) {                     // pc = 2 h + half: rows r = 8 (h & 1) + 4 half .. + 3 of X quarter h
        if (!p_live) return;
        const int h = pc >> 1, ti = h >> 1;
        const bool odd = n31 & 1;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 8 * (h & 1) + 4 * (pc & 1) + e;
                const float x1 = xv[(2 * ti + tj) * 16 + r], cv = pacc[ti][tj][r];
                const float qv = x1 * __builtin_amdgcn_rcpf(cv + 1e-9f);
                unsigned hi, lo;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(qv), "v"(0.f));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(qv - __uint_as_float(hi << 16)), "v"(0.f));
                // two adjacent lanes hold adjacent columns: one 4-byte store per lane -- the even lane the pair of the hi plane, the odd lane
                // the pair of the lo plane; the neighbour's half comes by DPP (quad_perm 1,0,3,2), the word by one byte permute, no branches
                // (__shfl_xor compiles to ds_bpermute_b32 + a full lgkmcnt wait per element: that was most of the old epilogue's time)
                const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? hi : lo), 0xB1, 0xf, 0xf, false);
                const unsigned word = __builtin_amdgcn_perm(odd ? lo : other, odd ? other : hi, 0x05040100u);
                unsigned short* q = ((r >> 2) & 1) ? p_q1 : p_q0;
                // (plain stores: non-temporal ones made this launch 8 % slower and the product that reads the planes 4 % faster)
                *reinterpret_cast<unsigned*>(q + tj * 4096 + ti * 1024 + 32 * ((r & 3) + 8 * (r >> 2))) = word;
                if (OBJ) {
                    float t = x1 * (__builtin_amdgcn_logf(x1 * __builtin_amdgcn_rcpf(cv)) * 0.69314718055994531f);
                    t = (t != t || t == __builtin_inff() || t == -__builtin_inff()) ? 0.f : t;
                    const float d2 = (t - x1) + cv;
                    if (r & 1) s1 += d2; else s0 += d2;
                }
            }
        if (OBJ) tot += (double)(s0 + s1);
    };
    issue();
    issue();
    const bool early = (stagger & 1) ? wave < 4 : true;      // (see gxt_gemm_kernel: the two waves of a SIMD request at different times)
    const unsigned xlane = (unsigned)(((int64_t)(4 * b) * ldx + n31) * 4);
    const unsigned long long ldx4 = (unsigned long long)ldx * 4ull;
    int64_t rbk = t0 / nct, ctk = t0 % nct;
    for (int64_t tile = t0; tile < t1; ++tile) {
        const bool second = 2 * rbk + 1 < mt;
        const bool live = second || wr < 128;          // (the duplicated tile of an odd last row block: its waves read the first tile's X and drop it)
        const int64_t i0 = rbk * 256 + ((live ? wr : wr - 128)), j0 = ctk * 128 + wc;
        const unsigned long long xb = (unsigned long long)(X + i0 * ldx + j0);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
        // X quarter h = (ti = h >> 1, rows r = 8 (h & 1) .. + 7, both column tiles): 16 dword loads, each a full 128-byte line per
        // half-wave (a lane's 16 entries of a 32 x 32 accumulator tile are one COLUMN: 16-byte loads along the rows touched 64 lines
        // per instruction and the texture path, not the matrix pipe, set the pace: 240 us per launch against 170 without X);
        // non-temporal, so that the stream of V does not push the factor images out of the XCD's L2 (PMC: 849 MB fetched per
        // launch against 562 MB of V and images; objective 206.7 -> 202 us, KL quotient 336 -> 310)
#define GXR_XQ(h) do { \
            const unsigned long long r0_ = xb + (unsigned long long)(32 * ((h) >> 1) + 16 * ((h) & 1)) * ldx4; \
            const int o_ = 32 * ((h) >> 1) + 8 * ((h) & 1); \
            asm volatile("global_load_dword %0, %16, %17 nt\n\tglobal_load_dword %1, %16, %17 offset:128 nt\n\t" \
                         "global_load_dword %2, %16, %18 nt\n\tglobal_load_dword %3, %16, %18 offset:128 nt\n\t" \
                         "global_load_dword %4, %16, %19 nt\n\tglobal_load_dword %5, %16, %19 offset:128 nt\n\t" \
                         "global_load_dword %6, %16, %20 nt\n\tglobal_load_dword %7, %16, %20 offset:128 nt\n\t" \
                         "global_load_dword %8, %16, %21 nt\n\tglobal_load_dword %9, %16, %21 offset:128 nt\n\t" \
                         "global_load_dword %10, %16, %22 nt\n\tglobal_load_dword %11, %16, %22 offset:128 nt\n\t" \
                         "global_load_dword %12, %16, %23 nt\n\tglobal_load_dword %13, %16, %23 offset:128 nt\n\t" \
                         "global_load_dword %14, %16, %24 nt\n\tglobal_load_dword %15, %16, %24 offset:128 nt" \
                         : "=&v"(xv[o_ + 0]), "=&v"(xv[o_ + 16]), "=&v"(xv[o_ + 1]), "=&v"(xv[o_ + 17]), "=&v"(xv[o_ + 2]), "=&v"(xv[o_ + 18]), \
                           "=&v"(xv[o_ + 3]), "=&v"(xv[o_ + 19]), "=&v"(xv[o_ + 4]), "=&v"(xv[o_ + 20]), "=&v"(xv[o_ + 5]), "=&v"(xv[o_ + 21]), \
                           "=&v"(xv[o_ + 6]), "=&v"(xv[o_ + 22]), "=&v"(xv[o_ + 7]), "=&v"(xv[o_ + 23]) \
                         : "v"(xlane), "s"(r0_), "s"(r0_ + ldx4), "s"(r0_ + 2 * ldx4), "s"(r0_ + 3 * ldx4), "s"(r0_ + 8 * ldx4), "s"(r0_ + 9 * ldx4), \
                           "s"(r0_ + 10 * ldx4), "s"(r0_ + 11 * ldx4) : "memory"); } while (0)
        // The first eight iterations are written out: each requests chunk g + 2 (waves 0-3 before their MFMAs, waves 4-7 behind them)
        // and, last, possibly a quarter of X (16 loads) -- GX_RESID: iterations 0-3 (its epilogue closes the tile); GX_KLQ: iterations
        // 1, 3, 5, 7.  GX_KLQ: an eighth of the quotient epilogue of the tile BEFORE (8 elements per lane: half of the X quarter that
        // the request at the end of the next odd iteration then overwrites, 8 stores) rides in each of them -- in front of the MFMAs
        // for waves 0-3, behind them for waves 4-7, so that one wave's VALU work runs beside its SIMD partner's MFMAs instead of both
        // queueing for each pipe in turn.  Vector-memory operations retire in issue order, loads and stores alike
        // (MI355X_MICROARCH.md, waitcnt), so the wait of iteration c leaves exactly those in flight that were issued behind chunk g:
        //     waves 0-3:  S(c-2) + X(c-2) + 6 + S(c-1) + X(c-1)        waves 4-7 (stores in front of the request):  X(c-2) + S(c-1) + 6 + X(c-1)
        // with S(j) = 8 where a piece ran in iteration j, X(j) = 16 where X was requested, j < 0 = the tail of the tile before
        // (counter limit 63: rounded down to the next 6 + 8 i, which only makes a wait stricter).
        const bool act = MODE == GX_KLQ && pending && p_live;          // pieces run in this tile's first eight iterations
        auto n_s = [&](int j) { return j < 0 ? ((j + nch < 8 && act_prev) ? 8 : 0) : ((j < 8 && act) ? 8 : 0); };
        auto n_x = [&](int j) {
            if (j < 0) { if (tile == t0) return 0; j += nch; }
            return MODE == GX_KLQ ? ((j < 8 && (j & 1)) ? 16 : 0) : (j < 4 ? 16 : 0);
        };
        // (Measured and dropped: touching the next tile's X half a tile ahead -- two dword loads per lane, one per 128-byte line of the
        //  wave's 64 x 64 floats, values dropped -- so that the requests proper find their lines in L2: objective 203 -> 235 us, KL
        //  quotient 330 -> 415.  Those 128 extra lines per wave and tile cost as much as the 64 X loads themselves: the launch is
        //  paced by the lines the CU's vector-memory path takes in, ~36 bytes per clock, not by the latency of X.)
        auto wait_chunk = [&](int c, bool last) {
            if (last) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
            const int n = 6 + n_x(c - 2) + n_s(c - 1) + n_x(c - 1) + (early ? n_s(c - 2) : 0);
            if (n >= 62) asm volatile("s_waitcnt vmcnt(62)" ::: "memory");
            else if (n >= 54) asm volatile("s_waitcnt vmcnt(54)" ::: "memory");
            else if (n >= 46) asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
            else if (n >= 38) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
            else if (n >= 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
            else if (n >= 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        };
#define GXR_ITER(c_) do { \
            wait_chunk(c_, false); \
            __syncthreads(); \
            if (early && q_left > 0) issue(); \
            if (MODE == GX_KLQ && early && pending) klq_piece(c_); \
            multiply();                                /* (ONE instance of the MFMA body per iteration: the accumulators stay in place) */ \
            if (MODE == GX_KLQ && !early && pending) klq_piece(c_); \
            if (!early && q_left > 0) issue(); } while (0)
        if (MODE == GX_KLQ) {
            GXR_ITER(0); GXR_ITER(1); GXR_XQ(0);
            GXR_ITER(2); GXR_ITER(3); GXR_XQ(1);
            GXR_ITER(4); GXR_ITER(5); GXR_XQ(2);
            GXR_ITER(6);
            wait_chunk(7, tile + 1 == t1 && nch == 8);
            __syncthreads();
            if (early && q_left > 0) issue();
            if (early && pending) klq_piece(7);
            multiply();
            if (!early && pending) klq_piece(7);
            if (!early && q_left > 0) issue();
            GXR_XQ(3);
        } else {
            GXR_ITER(0); GXR_XQ(0);
            GXR_ITER(1); GXR_XQ(1);
            GXR_ITER(2); GXR_XQ(2);
            GXR_ITER(3); GXR_XQ(3);
            GXR_ITER(4); GXR_ITER(5); GXR_ITER(6);
            wait_chunk(7, tile + 1 == t1 && nch == 8);
            __syncthreads();
            if (early && q_left > 0) issue();
            multiply();
            if (!early && q_left > 0) issue();
        }
#undef GXR_ITER
#undef GXR_XQ
        for (int c = 8; c < nch; ++c) {
            wait_chunk(c, tile + 1 == t1 && c + 1 == nch);
            __syncthreads();
            if (early && q_left > 0) issue();
            multiply();
            if (!early && q_left > 0) issue();
        }
        act_prev = act;
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(xv[i]));       // (the uses stay below the loop's counted waits)
        // acc[ti][tj][r] = C(i0 + 32 ti + (r & 3) + 8 (r >> 2) + 4 b, j0 + 32 tj + n31)
        if (MODE == GX_KLQ) {                          // handed to the next tile's first four iterations (or to the flush below)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) pacc[ti][tj] = acc[ti][tj];
            pending = true;
            p_live = live;
            // this lane's 4-byte slots in the tiled image (contraction along the columns; row I = i0 + 32 ti + (r & 3) + 8 (r >> 2) + 4 b,
            // columns J, J + 1 with J = j0 + 32 tj + (n31 & ~1)): tile ((I >> 7) (ldq / 32) + (J >> 5)) 4096 + (I & 127) 32 +
            // 8 (((J & 31) >> 3) ^ ((I >> 2) & 3)) + (J & 7), where (I >> 2) & 3 = (2 (r >> 2) + b) & 3 = b for even r >> 2, b ^ 2 for odd
            const int64_t tb = ((2 * rbk + (wr >> 7)) * (ldq / 32) + 4 * ctk + (wc >> 5)) * 4096 + 32 * ((wr & 127) + 4 * b) + (n31 & 6);
            unsigned short* const pl = (n31 & 1) ? Qlo : Qhi;      // the even lane stores the pair of the hi plane, the odd lane that of the lo plane
            p_q0 = pl + tb + 8 * ((n31 >> 3) ^ b);
            p_q1 = pl + tb + 8 * ((n31 >> 3) ^ (b ^ 2));
        } else if (live) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = xv[(2 * ti + tj) * 16 + r] - acc[ti][tj][r], d2 = d * d;
                        if ((r & 3) == 0) s0 += d2; else if ((r & 3) == 1) s1 += d2; else if ((r & 3) == 2) s2 += d2; else s3 += d2;
                    }
                    tot += (double)((s0 + s1) + (s2 + s3));
                }
        }
        if (++ctk == nct) { ctk = 0; ++rbk; }
    }
    if (MODE == GX_KLQ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (nch = 8: the last X quarter was requested behind the last wait)
    if (MODE == GX_KLQ && pending) { klq_piece(0); klq_piece(1); klq_piece(2); klq_piece(3); klq_piece(4); klq_piece(5); klq_piece(6); klq_piece(7); }
    if (!part) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    __syncthreads();                                   // (every wave has read its last stage)
    double* red = reinterpret_cast<double*>(gxt_smem);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < 8; ++w) s += red[w];
        part[blockIdx.x] = (MODE == GX_RESID ? 0.5 : 1.0) * s;
    }
    if (blockIdx.x == 0)                               // (the callers sum one entry per 128 x 128 tile: the unused ones are zero)
        for (int64_t i = nb + tid; i < npart; i += 512) part[i] = 0.0;
}

// the switch between the two forms of the short contractions: the factor images are written in the format the chosen kernel reads
bool gxr_on() { static const bool on = !(getenv("NMFX_GXR") && atoi(getenv("NMFX_GXR")) == 0); return on; }

int gxr_launch(nmfx_engine* E, int mode, const unsigned short* Ahi, const unsigned short* Alo, const unsigned short* Bhi, const unsigned short* Blo,
               int64_t ldq, int64_t M, int64_t N, int64_t K, const float* X, int64_t ldx, double* part, const int* flag2,
               unsigned short* Qhi, unsigned short* Qlo) {
    if (M % 128 || N % 128 || K % 128 || K < 256) { E->err = "gxr_launch: shape"; return NMFX_E_ARG; }
    const int64_t T = ((M / 128 + 1) / 2) * (N / 128), npart = (M / GX_T) * (N / GX_T);
    const int nb = (int)std::min<int64_t>(E->ncu, T);
    const int* flag = &E->state->flag;
    int rc;
    const dim3 grid((unsigned)nb), block(512);
#define GXR_GO(MODE_, OBJ_) do { \
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gxr_kernel<MODE_, OBJ_>), GXT_SHM))) return rc; \
        hipLaunchKernelGGL((gxr_kernel<MODE_, OBJ_>), grid, block, GXT_SHM, E->stream, Ahi, Alo, Bhi, Blo, M, N, K, X, ldx, part, npart, flag, flag2, \
                           Qhi, Qlo, ldq, gx_stagger()); } while (0)
    if (mode == GX_KLQ) { if (part) GXR_GO(GX_KLQ, true); else GXR_GO(GX_KLQ, false); }
    else GXR_GO(GX_RESID, true);
#undef GXR_GO
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// bf16 hi / lo images of M [rows][cols] (f32, row-major) in the formats the product kernels read.  Natural orientation (operand rows =
// rows of M, contraction along the columns) and transposed orientation (operand rows = columns of M, contraction along the rows),
// each 0 = not wanted, 1 = row-major planes (gxb_gemm_kernel: the short contractions), 2 = the tiled image above (gxt_gemm_kernel).
// One 64 x 64 tile per block through LDS; every global store is a 16-byte chunk of eight contraction indices.
__global__ __launch_bounds__(256) void gxt_split_kernel(const float* __restrict__ M, int64_t rows, int64_t cols, int fmt_n,
                                                        unsigned short* __restrict__ nhi, unsigned short* __restrict__ nlo, int fmt_t,
                                                        unsigned short* __restrict__ thi, unsigned short* __restrict__ tlo, const int* __restrict__ flag,
                                                        GxUpd up)
{
    if (flag && *flag) return;                         // (a stopped run keeps the images of the iterate it stopped at)
    __shared__ __attribute__((aligned(16))) unsigned short sh[64][72], sl[64][72];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    for (int r = ty; r < 64; r += 4) {
        const int64_t idx = (r0 + r) * cols + c0 + tx;
        float v;
        if (up.xold) {                                 // (block-uniform)
            const float xo = up.xold[idx];
            float nu = up.num[idx];
            for (int sl = 1; sl < up.nslab; ++sl) nu += up.num[sl * up.nstride + idx];
            v = xo * nu / (up.den[idx] + up.lam * xo + 1e-9f);
            up.xnew[idx] = v;
        } else v = M[idx];
        unsigned hi, lo;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v), "v"(0.f));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v - __uint_as_float(hi << 16)), "v"(0.f));
        sh[r][tx] = (unsigned short)hi; sl[r][tx] = (unsigned short)lo;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i, a = q >> 3, e8 = q & 7;
        if (fmt_n) {                                   // row r0 + a, contraction indices c0 + 8 e8 ..
            const uint4 h = *reinterpret_cast<const uint4*>(&sh[a][8 * e8]), l = *reinterpret_cast<const uint4*>(&sl[a][8 * e8]);
            const int64_t R = r0 + a, Kc = c0 + 8 * e8;
            int64_t at;
            if (fmt_n == 1) at = R * cols + Kc;
            else { const int rr = (int)(R & 127); at = ((R >> 7) * (cols / 32) + (Kc >> 5)) * 4096 + rr * 32 + 8 * ((int)((Kc & 31) >> 3) ^ ((rr >> 2) & 3)); }
            *reinterpret_cast<uint4*>(nhi + at) = h; *reinterpret_cast<uint4*>(nlo + at) = l;
        }
        if (fmt_t) {                                   // operand row c0 + a (a column of M), contraction indices r0 + 8 e8 ..
            union { unsigned short s[8]; uint4 u; } h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) { h.s[e] = sh[8 * e8 + e][a]; l.s[e] = sl[8 * e8 + e][a]; }
            const int64_t R = c0 + a, Kc = r0 + 8 * e8;
            int64_t at;
            if (fmt_t == 1) at = R * rows + Kc;
            else { const int rr = (int)(R & 127); at = ((R >> 7) * (rows / 32) + (Kc >> 5)) * 4096 + rr * 32 + 8 * ((int)((Kc & 31) >> 3) ^ ((rr >> 2) & 3)); }
            *reinterpret_cast<uint4*>(thi + at) = h.u; *reinterpret_cast<uint4*>(tlo + at) = l.u;
        }
    }
}

int gxt_split(nmfx_engine* E, const float* M, int64_t rows, int64_t cols, int fmt_n, unsigned short* nhi, unsigned short* nlo, int fmt_t,
              unsigned short* thi, unsigned short* tlo, bool check_flag, const GxUpd& up = GxUpd()) {
    hipLaunchKernelGGL(gxt_split_kernel, dim3((unsigned)(cols / 64), (unsigned)(rows / 64)), dim3(256), 0, E->stream, M, rows, cols, fmt_n, nhi, nlo,
                       fmt_t, thi, tlo, check_flag ? (const int*)&E->state->flag : (const int*)nullptr, up);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
// the factor images of the composed path: W -> Whi / Wlo [mp][kp] row-major (short contractions: objective, quotient) and W^T tiled
// (W^T V, W^T W); H -> tiled (V H^T, H H^T) and H^T [np][kp] row-major
int gxb_images_w(nmfx_engine* E, const float* W, const GxUpd& up = GxUpd()) {
    return gxt_split(E, W, E->mp, E->kp, gxr_on() ? 2 : 1, E->Whi[0], E->Wlo[0], 2, E->WThi, E->WTlo, true, up);
}
int gxb_images_h(nmfx_engine* E, const float* H, const GxUpd& up = GxUpd()) {
    return gxt_split(E, H, E->kp, E->np, 2, E->Hhi, E->Hlo, gxr_on() ? 2 : 1, E->HThi, E->HTlo, true, up);
}

// split-K product on the tiled planes into the slab buffer gx_s, summed into `out` (M: rows of A, N: rows of B, K: contraction)
int gxt_split_product(nmfx_engine* E, const unsigned short* Ahi, const unsigned short* Alo, const unsigned short* Bhi, const unsigned short* Blo,
                      float* out, int64_t M, int64_t N, int64_t K, int cap, int terms = 3, const float** slabs = nullptr, int* nslab = nullptr,
                      bool stream_hint = true) {
    // (slabs != nullptr: the caller sums the slabs itself -- *slabs / *nslab say where they are -- and `out` is only written when there is one)
    int rc;
    static const int big = getenv("NMFX_GXT2") ? atoi(getenv("NMFX_GXT2")) : 1;
    if (big && M % 256 == 0 && N % 256 == 0 && M * N >= 2 * 256 * 256) {      // 256 x 256 tiles (not the Gram matrices: one tile)
        const int64_t blocks = (M / 256) * (N / 256), ch = K / 32;
        int64_t S = std::max<int64_t>(1, std::min<int64_t>(2 * cap, ((int64_t)E->ncu + blocks - 1) / blocks));
        const int64_t slab_cap = std::max<int64_t>(std::max<int64_t>(8 * (int64_t)E->kp * E->np, 64 * (int64_t)E->kp * E->kp), (E->mp / 64) * (int64_t)E->kp);
        while (S > 1 && (S * M * N > slab_cap || ch % S)) --S;
        const dim3 grid((unsigned)(N / 256), (unsigned)(M / 256), (unsigned)S), block(512);
        static const int nt_on = getenv("NMFX_GXT_NT") ? atoi(getenv("NMFX_GXT_NT")) : 1;
        // (an operand is read once where the other dimension is a single tile: with two column tiles -- k = 512 -- the second one finds
        //  the V planes in L2 and non-temporal requests cost 4 %; k = 256: V H^T 174.6 -> 169.4 us, W^T V 180.3 -> 176.5)
        const int nt = (!nt_on || !stream_hint) ? 0 : ((N == 256 && M > 256 ? 1 : 0) | (M == 256 && N > 256 ? 2 : 0));
        float* C = S == 1 ? out : E->gx_s;
        const int* flag = &E->state->flag;
        if (terms == 4) {
            if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gxt2_gemm_kernel<4>), GXT2_SHM))) return rc;
            hipLaunchKernelGGL((gxt2_gemm_kernel<4>), grid, block, GXT2_SHM, E->stream, Ahi, Alo, Bhi, Blo, C, N, M * N, K, flag, (const int*)nullptr, nt);
        } else {
            if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gxt2_gemm_kernel<3>), GXT2_SHM))) return rc;
            hipLaunchKernelGGL((gxt2_gemm_kernel<3>), grid, block, GXT2_SHM, E->stream, Ahi, Alo, Bhi, Blo, C, N, M * N, K, flag, (const int*)nullptr, nt);
        }
        NMFX_HIP(hipGetLastError());
        if (slabs) { *slabs = C; *nslab = (int)S; return NMFX_OK; }
        if (S == 1) return NMFX_OK;
        return nmfx_launch_sum_partials(E, E->gx_s, (int)S, M * N, out);
    }
    const int64_t blocks = ((M / 128 + 1) / 2) * (N / 128), ch = K / 32;
    int64_t S = std::max<int64_t>(1, std::min<int64_t>(cap, ((int64_t)E->ncu + blocks - 1) / blocks));
    const int64_t slab_cap = std::max<int64_t>(std::max<int64_t>(8 * (int64_t)E->kp * E->np, 64 * (int64_t)E->kp * E->kp), (E->mp / 64) * (int64_t)E->kp);
    while (S > 1 && (S * M * N > slab_cap || ch % S)) --S;
    const dim3 grid((unsigned)(N / 128), (unsigned)((M / 128 + 1) / 2), (unsigned)S), block(512);
    float* C = S == 1 ? out : E->gx_s;
    const int* flag = &E->state->flag;
    if (terms == 4) {
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gxt_gemm_kernel<4>), GXT_SHM))) return rc;
        hipLaunchKernelGGL((gxt_gemm_kernel<4>), grid, block, GXT_SHM, E->stream, Ahi, Alo, Bhi, Blo, C, N, M * N, M, K, flag, (const int*)nullptr, gx_stagger());
    } else {
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gxt_gemm_kernel<3>), GXT_SHM))) return rc;
        hipLaunchKernelGGL((gxt_gemm_kernel<3>), grid, block, GXT_SHM, E->stream, Ahi, Alo, Bhi, Blo, C, N, M * N, M, K, flag, (const int*)nullptr, gx_stagger());
    }
    NMFX_HIP(hipGetLastError());
    if (slabs) { *slabs = C; *nslab = (int)S; return NMFX_OK; }
    if (S == 1) return NMFX_OK;
    return nmfx_launch_sum_partials(E, E->gx_s, (int)S, M * N, out);
}

// the split-bf16 products are the default for the Euclidean MUR loop beyond k = 128 (NMFX_PRECISION=f32 keeps the exact-f32 kernel)
bool gxb_on(const nmfx_engine* E) { return E->precision == 1 && !E->gxb_disabled && E->mp % 128 == 0 && E->np % 128 == 0 && E->kp % 128 == 0; }
constexpr int GXB_NOFIT = 1;       // gxb_prepare: not an error -- the planes do not fit, the handle falls back to the exact-f32 product kernel

// out [cols][rows] = in [rows][cols]^T (f32, 64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void gx_transpose_kernel(const float* __restrict__ in, int64_t rows, int64_t cols, float* __restrict__ out)
{
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = in[(r0 + r) * cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4) out[(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

// V planes (once per upload) and the factor images of (W, H) (whenever they are not the ones the loop left); kl: also V^T in f32
// and the two quotient planes
int gxb_prepare(nmfx_engine* E, const float* W, bool kl = false) {
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    {   // what is still to be allocated must fit beside what the handle holds (like nmfx_create's fall back for k <= 128: visible in nmfx_get_note)
        size_t need = 0, free_b = 0, total_b = 0;
        for (int i = 0; i < 4; ++i) if (!E->gxb_v[i]) need += (size_t)mp * np * 2;
        if (!E->WThi) need += (size_t)4 * mp * kp * 2;
        if (!E->HThi) need += (size_t)4 * kp * np * 2;
        if (kl && !E->gxb_vt) need += (size_t)mp * np * 8;
        static const bool pretend = getenv("NMFX_GXB_NOFIT") != nullptr;          // (tests: take the fall back without filling the device)
        if (need && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (pretend || free_b < need + ((size_t)256 << 20))) {
            E->gxb_disabled = true;
            E->gxb_img_ready = false;
            nmfx_comm_invalidate(E);               // (a captured graph of a sharded run would replay the split-bf16 launches)
            char buf[256];
            snprintf(buf, sizeof buf, "%sk > 128: %.1f GiB free, the bf16 operand planes need %.1f GiB more: exact-f32 product kernel on this handle",
                     E->note.empty() ? "" : "; ", (double)free_b / (1 << 30), (double)need / (1 << 30));
            E->note += buf;
            return GXB_NOFIT;
        }
    }
    if (kl) {
        if ((rc = gx_alloc(E, &E->gxb_vt, mp * np)) || (rc = gx_alloc(E, &E->gxb_q[0], mp * np)) || (rc = gx_alloc(E, &E->gxb_q[1], mp * np))) return rc;
        if (!E->gxb_vt_ready) {
            ProfScope ps(E, "images");
            hipLaunchKernelGGL(gx_transpose_kernel, dim3((unsigned)(np / 64), (unsigned)(mp / 64)), dim3(256), 0, E->stream, (const float*)E->V, mp, np, E->gxb_vt);
            NMFX_HIP(hipGetLastError());
            E->gxb_vt_ready = true;
        }
    }
    for (int i = 0; i < 4; ++i) if ((rc = gx_alloc(E, &E->gxb_v[i], mp * np))) return rc;
    if ((rc = gx_alloc(E, &E->Whi[0], mp * kp)) || (rc = gx_alloc(E, &E->Wlo[0], mp * kp)) || (rc = gx_alloc(E, &E->WThi, mp * kp)) ||
        (rc = gx_alloc(E, &E->WTlo, mp * kp)) || (rc = gx_alloc(E, &E->Hhi, kp * np)) || (rc = gx_alloc(E, &E->Hlo, kp * np)) ||
        (rc = gx_alloc(E, &E->HThi, kp * np)) || (rc = gx_alloc(E, &E->HTlo, kp * np))) return rc;
    if (!E->gxb_v_ready) {
        ProfScope ps(E, "images");
        if ((rc = gxt_split(E, E->V, mp, np, 2, E->gxb_v[0], E->gxb_v[1], 2, E->gxb_v[2], E->gxb_v[3], false))) return rc;      // both TILED: V (contraction n), V^T (contraction m)
        E->gxb_v_ready = true;
    }
    if (!E->gxb_img_ready) {
        ProfScope ps(E, "images");
        if ((rc = gxb_images_w(E, W)) || (rc = gxb_images_h(E, E->H))) return rc;
        E->gxb_img_ready = true;
    }
    return NMFX_OK;
}

}  // namespace

static bool gx_anls_bf16() { static const bool on = !(getenv("NMFX_GX_ANLS_BF16") && atoi(getenv("NMFX_GX_ANLS_BF16")) == 0); return on; }

// r4: the denominator products of the Euclidean updates, D = W (H H^T) and E = (W^T W) H, from the factor images and the images of the
// k x k Gram matrix (three terms, like the numerators they are divided into) instead of an exact-f32 product: 2 m k^2 flop that took
// 28 us at k = 256 and 110 us at k = 512 on the f32 matrix cores.  Measured (16384 x 8192): W side k = 512 130 -> 102 us per update,
// k = 256 no gain (the image launch and a 64-block grid eat it): used for the W side from kp = 384 on; H side never (see phase B).  left = true: out [rows][kp] = F G (F images: rows x kp);
// false: out [kp][cols] = G F^T^T, i.e. G times the factor whose TRANSPOSED images are given (cols x kp)
static bool gx_bf16_den() { static const bool on = !(getenv("NMFX_GX_DEN_BF16") && atoi(getenv("NMFX_GX_DEN_BF16")) == 0); return on; }
static int gx_den_product(nmfx_engine* E, bool left, const float* G, const unsigned short* Fhi, const unsigned short* Flo, int64_t ents, float* out) {
    int rc;
    const int64_t kp = E->kp;
    if ((rc = gx_alloc(E, &E->gx_gimg, 2 * kp * kp))) return rc;
    if ((rc = gxt_split(E, G, kp, kp, 2, E->gx_gimg, E->gx_gimg + kp * kp, 0, nullptr, nullptr, true))) return rc;      // (G symmetric: rows = either index)
    if (left) return gxt_split_product(E, Fhi, Flo, E->gx_gimg, E->gx_gimg + kp * kp, out, ents, kp, kp, 0, 3, nullptr, nullptr, false);
    return gxt_split_product(E, E->gx_gimg, E->gx_gimg + kp * kp, Fhi, Flo, out, kp, ents, kp, 0, 3, nullptr, nullptr, false);
}

// ---- MUR, Euclidean ----------------------------------------------------------------------------------------------------------
int nmfx_generic_mur_phase_a(nmfx_engine* E, int distance, double lambda, int64_t j) {
    int rc;
    const bool kl = distance == NMFX_KL;
    if ((rc = gx_buffers(E, kl && !gxb_on(E)))) return rc;
    if (!gxb_on(E)) E->gxb_img_ready = false;          // (exact-f32 iterations rewrite W and H without their images)
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    const float* W = E->W[j & 1];
    float* Wn = E->W[(j + 1) & 1];
    const int64_t nblk = (mp / GX_T) * (np / GX_T);
    float* xB = E->xf32;                               // [kp][np]
    float* xG = E->xf32 + kp * np;                     // [kp][kp]
    float* xS = xG + kp * kp;                          // [kp] column sums of W (KL)
    if (gxb_on(E)) {                                   // (operands of the split-bf16 products; if they do not fit: exact f32 from here on, and its m x n quotient buffer for KL)
        rc = gxb_prepare(E, W, kl);
        if (rc == GXB_NOFIT) { if ((rc = gx_buffers(E, kl))) return rc; }
        else if (rc) return rc;
    }
    if (!kl && gxb_on(E)) {                            // the same steps with the V-sized products and the Gram matrices in split bf16
        { ProfScope ps(E, "objective");
          if ((rc = gxb_launch(E, GX_RESID, E->Whi[0], E->Wlo[0], kp, E->HThi, E->HTlo, kp, nullptr, 0, 0, mp, np, kp, 1, E->V, np, E->gx_part))) return rc; }
        if ((rc = nmfx_launch_obj_reduce(E, nblk, E->gx_part))) return rc;
        { ProfScope ps(E, "gram_nt");
          if ((rc = gxt_split_product(E, E->Hhi, E->Hlo, E->Hhi, E->Hlo, E->HHt, kp, kp, np, 64))) return rc; }
        if (gx_fuse_update()) {                        // r4: slab sum, update and the images of W_new in one launch behind D = W (H H^T)
            const float* slabs = nullptr; int nslab = 0;
            { ProfScope ps(E, "wphase");               // A = V H^T
              if ((rc = gxt_split_product(E, E->gxb_v[0], E->gxb_v[1], E->Hhi, E->Hlo, E->A_part, mp, kp, np, 4, 3, &slabs, &nslab))) return rc; }
            ProfScope ps(E, "w_update");
            if (gxr_on() && gx_bf16_den() && kp >= 384) rc = gx_den_product(E, true, E->HHt, E->Whi[0], E->Wlo[0], mp, E->gx_d);      // (tiled images of W: the persistent kernels' format)
            else rc = gx_launch<true, false>(E, GX_STORE, W, kp, E->HHt, kp, E->gx_d, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr);
            if (rc) return rc;
            GxUpd up;
            up.xold = W; up.num = slabs; up.nslab = nslab; up.nstride = mp * kp; up.den = E->gx_d; up.lam = (float)lambda; up.xnew = Wn;
            if ((rc = gxb_images_w(E, Wn, up))) return rc;
        } else {
        { ProfScope ps(E, "wphase");                   // A = V H^T
          if ((rc = gxt_split_product(E, E->gxb_v[0], E->gxb_v[1], E->Hhi, E->Hlo, E->A_part, mp, kp, np, 4))) return rc; }
        { ProfScope ps(E, "w_update");
          if ((rc = gx_launch<true, false>(E, GX_STORE, W, kp, E->HHt, kp, E->gx_d, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr))) return rc;
          const int64_t c4 = mp * kp / 4;
          hipLaunchKernelGGL(gx_eu_update_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, E->stream, W, (const float*)E->A_part,
                             (const float*)E->gx_d, (float)lambda, Wn, c4, (const int*)&E->state->flag);
          NMFX_HIP(hipGetLastError()); }
        { ProfScope ps(E, "images");
          if ((rc = gxb_images_w(E, Wn))) return rc; }
        }
        { ProfScope ps(E, "gram_tn");
          if ((rc = gxt_split_product(E, E->WThi, E->WTlo, E->WThi, E->WTlo, xG, kp, kp, mp, 64))) return rc; }
        { ProfScope ps(E, "hphase");                   // B = W^T V
          if ((rc = gxt_split_product(E, E->WThi, E->WTlo, E->gxb_v[2], E->gxb_v[3], xB, kp, np, mp, 8))) return rc; }
        return NMFX_OK;
    }
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
    if (gxb_on(E)) {
        // split bf16: the quotient leaves its product as bf16 planes in the layout the next product contracts over -- Q [mp][np] from
        // (W images) x (H^T images) for A = Q H^T, and the H side's Q'^T [np][mp] from (H^T images) x (W_new images) against V^T for
        // B = W_new^T Q' -- so all four V-sized products are the one NT kernel (three terms, like the tuned MUR-KL kernels)
        { ProfScope ps(E, "objective");
          if ((rc = gxb_launch(E, GX_KLQ, E->Whi[0], E->Wlo[0], kp, E->HThi, E->HTlo, kp, nullptr, np, 0, mp, np, kp, 1, E->V, np, E->gx_part, nullptr,
                               E->gxb_q[0], E->gxb_q[1]))) return rc; }
        if ((rc = nmfx_launch_obj_reduce(E, nblk, E->gx_part))) return rc;
        { ProfScope ps(E, "wphase");                   // Q H^T
          if ((rc = gxt_split_product(E, E->gxb_q[0], E->gxb_q[1], E->Hhi, E->Hlo, E->A_part, mp, kp, np, 4))) return rc; }
        { ProfScope ps(E, "w_update");
          hipLaunchKernelGGL(gx_rowsum_kernel, dim3((unsigned)kp), dim3(256), 0, E->stream, (const float*)E->H, np, E->HHt, (const int*)&E->state->flag);
          const int64_t cnt = mp * kp;
          hipLaunchKernelGGL((gx_kl_update_kernel<false>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, E->stream, W, (const float*)E->A_part,
                             (const float*)E->HHt, (float)lambda, Wn, mp, kp, E->k, (const int*)&E->state->flag);
          NMFX_HIP(hipGetLastError()); }
        { ProfScope ps(E, "images");
          if ((rc = gxb_images_w(E, Wn))) return rc; }
        { ProfScope ps(E, "objective");                // Q'^T = V^T / (H^T W_new^T + 1e-9)  (the second quotient product of the iteration, no objective)
          if ((rc = gxb_launch(E, GX_KLQ, E->HThi, E->HTlo, kp, E->Whi[0], E->Wlo[0], kp, nullptr, mp, 0, np, mp, kp, 1, E->gxb_vt, mp, nullptr, nullptr,
                               E->gxb_q[0], E->gxb_q[1]))) return rc; }
        { ProfScope ps(E, "hphase");                   // B = W_new^T Q', d = W_new^T 1
          if ((rc = gxt_split_product(E, E->WThi, E->WTlo, E->gxb_q[0], E->gxb_q[1], xB, kp, np, mp, 8))) return rc;
          const int rb = (int)(mp / 64);
          hipLaunchKernelGGL(gx_colsum_part_kernel, dim3((unsigned)rb), dim3(256), 0, E->stream, (const float*)Wn, kp, 64, E->gx_s, (const int*)&E->state->flag);
          NMFX_HIP(hipGetLastError());
          if ((rc = nmfx_launch_sum_partials(E, E->gx_s, rb, kp, xS))) return rc; }
        return NMFX_OK;
    }
    E->gxb_img_ready = false;                          // (the exact-f32 KL loop rewrites W and H without their images)
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
    if ((rc = gx_buffers(E, kl && !gxb_on(E)))) return rc;
    const int64_t np = E->np, kp = E->kp;
    float* xB = E->xf32;
    float* xG = E->xf32 + kp * np;
    float* xS = xG + kp * kp;
    ProfScope ps(E, "h_update");
    hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                       E->state, E->obj_hist);
    NMFX_HIP(hipGetLastError());
    if (!kl && gxb_on(E) && E->gxb_img_ready && gx_fuse_update()) {
        // r4: the update and the images of the new H in one launch behind E = G H (element-wise, so H is updated in place)
        if (gxr_on() && gx_bf16_den() && false) rc = gx_den_product(E, false, xG, E->HThi, E->HTlo, np, E->gx_d);      // (measured: 80 against 77 us at k = 512, 51 against 47 at k = 256 -- 32 blocks of 256 x 256 for a kp x np output)
        else rc = gx_launch<true, false>(E, GX_STORE, xG, kp, E->H, np, E->gx_d, np, 0, kp, np, kp, 1, nullptr, 0, nullptr);
        if (rc) return rc;
        GxUpd up;
        up.xold = E->H; up.num = xB; up.nslab = 1; up.den = E->gx_d; up.lam = (float)lambda; up.xnew = E->H;
        if ((rc = gxb_images_h(E, E->H, up))) return rc;
    } else if (!kl) {
        if ((rc = gx_launch<true, false>(E, GX_STORE, xG, kp, E->H, np, E->gx_d, np, 0, kp, np, kp, 1, nullptr, 0, nullptr))) return rc;
        const int64_t c4 = kp * np / 4;
        hipLaunchKernelGGL(gx_eu_update_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, E->stream, (const float*)E->H, (const float*)xB,
                           (const float*)E->gx_d, (float)lambda, E->H, c4, (const int*)&E->state->flag);
        NMFX_HIP(hipGetLastError());
        if (gxb_on(E) && E->gxb_img_ready &&           // (images of the new H for the next iteration's split-bf16 products)
            (rc = gxb_images_h(E, E->H))) return rc;
    } else {
        const int64_t cnt = kp * np;
        hipLaunchKernelGGL((gx_kl_update_kernel<true>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, E->stream, (const float*)E->H, (const float*)xB,
                           (const float*)xS, (float)lambda, E->H, kp, np, E->k, (const int*)&E->state->flag);
        NMFX_HIP(hipGetLastError());
        if (gxb_on(E) && E->gxb_img_ready &&
            (rc = gxb_images_h(E, E->H))) return rc;
    }
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// objective partial of the current pair -> xf64[0] (the closing step: nmfx_mur_finish_a)
int nmfx_generic_mur_finish_a(nmfx_engine* E, int distance, int64_t j) {
    int rc;
    const bool kl = distance == NMFX_KL;
    if ((rc = gx_buffers(E, kl && !gxb_on(E)))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    const int64_t nblk = (mp / GX_T) * (np / GX_T);
    if (gxb_on(E)) {
        rc = gxb_prepare(E, E->W[j & 1], kl);
        if (rc == GXB_NOFIT) { if ((rc = gx_buffers(E, kl))) return rc; }
        else if (rc) return rc;
    }
    { ProfScope ps(E, "objective");
      if (!kl && gxb_on(E)) {
          rc = gxb_launch(E, GX_RESID, E->Whi[0], E->Wlo[0], kp, E->HThi, E->HTlo, kp, nullptr, 0, 0, mp, np, kp, 1, E->V, np, E->gx_part);
      } else if (!kl) rc = gx_launch<true, false>(E, GX_RESID, E->W[j & 1], kp, E->H, np, nullptr, 0, 0, mp, np, kp, 1, E->V, np, E->gx_part);
      else if (gxb_on(E)) {
          rc = gxb_launch(E, GX_KLQ, E->Whi[0], E->Wlo[0], kp, E->HThi, E->HTlo, kp, nullptr, np, 0, mp, np, kp, 1, E->V, np, E->gx_part, nullptr,
                          E->gxb_q[0], E->gxb_q[1]);
      } else rc = gx_launch<true, false>(E, GX_KLQ, E->W[j & 1], kp, E->H, np, E->S, np, 0, mp, np, kp, 1, E->V, np, E->gx_part);
      if (rc) return rc; }
    return nmfx_launch_obj_reduce(E, nblk, E->gx_part);
}

// ---- AO-ADMM, least-squares loss, prox nn / l1n (nmf/ao_admm.py:46-68, 113-124, 33-43, 259-292) for k > 128 -------------------------
namespace {

// rho = trace(G) / k, M^-1 = (G + rho I)^-1 (ao_admm.py:53-55, 59: cholesky + cho_solve) by an in-place Gauss-Jordan inversion in
// f64 without pivoting (the matrix is positive definite: a pivot <= 0 is the reference's LinAlgError -> st->notpd), one workgroup,
// the matrix in a global f64 work area (L2-resident), pivot row and column through LDS, two barriers per pivot.  fixed_rho >= 0
// replaces trace / k.  Also opens the sub-problem: inner_stop = inner_count = 0.
__global__ __launch_bounds__(1024) void gx_prepare_kernel(const float* __restrict__ G, int kp, int k, double* __restrict__ work,
                                                          float* __restrict__ Minv, DevState* __restrict__ st, double fixed_rho)
{
    if (st->flag) return;
    extern __shared__ double gsh[];                    // row [kp] | col [kp] | 16 partials
    double* prow = gsh;
    double* pcol = gsh + kp;
    double* part = pcol + kp;
    const int tid = threadIdx.x, nt = blockDim.x;
    double tr = 0.0;
    for (int i = tid; i < k; i += nt) tr += (double)G[(int64_t)i * kp + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tr += __shfl_down(tr, off, 64);
    if ((tid & 63) == 0) part[tid >> 6] = tr;
    __syncthreads();
    double rho = 0.0;
    for (int w = 0; w < nt / 64; ++w) rho += part[w];
    rho /= (double)k;
    if (fixed_rho >= 0.0) rho = fixed_rho;
    const int64_t kk = (int64_t)kp * kp;
    for (int64_t e = tid; e < kk; e += nt) work[e] = (double)G[e] + ((e / kp) == (e % kp) ? rho : 0.0);
    __syncthreads();
    int bad = 0;
    for (int p = 0; p < kp; ++p) {
        for (int i = tid; i < kp; i += nt) { prow[i] = work[(int64_t)p * kp + i]; pcol[i] = work[(int64_t)i * kp + p]; }
        __syncthreads();
        const double d = prow[p];
        if (!(d > 0.0)) bad = 1;
        const double inv = 1.0 / d;
        for (int64_t e = tid; e < kk; e += nt) {
            const int i = (int)(e / kp), j = (int)(e % kp);
            double v;
            if (i == p) v = (j == p) ? inv : prow[j] * inv;
            else v = (j == p) ? -pcol[i] * inv : work[e] - pcol[i] * (prow[j] * inv);
            work[e] = v;
        }
        __syncthreads();
    }
    for (int64_t e = tid; e < kk; e += nt) Minv[e] = (float)work[e];
    if (tid == 0) { st->rho = rho; st->inner_stop = 0; st->inner_count = 0; if (bad) st->notpd = 1; }
}

// rhs = B + rho (X + U)   (ao_admm.py:59, the argument of cho_solve)
__global__ __launch_bounds__(256) void gx_rhs_kernel(const float* __restrict__ B, const float* __restrict__ X, const float* __restrict__ U,
                                                     float* __restrict__ rhs, int64_t count4, const DevState* __restrict__ st)
{
    if (st->flag || st->inner_stop) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count4) return;
    const float rho = (float)st->rho;
    const float4 b = reinterpret_cast<const float4*>(B)[i], x = reinterpret_cast<const float4*>(X)[i], u = reinterpret_cast<const float4*>(U)[i];
    reinterpret_cast<float4*>(rhs)[i] = make_float4(b.x + rho * (x.x + u.x), b.y + rho * (x.y + u.y), b.z + rho * (x.z + u.z), b.w + rho * (x.w + u.w));
}

// X = prox(aux, U); U += X - aux; the four sums of squares `terminate` needs (ao_admm.py:60-62, 33-43): part[block][4].
// With Bm != nullptr it also leaves the NEXT round's right-hand side rhs = Bm + rho (X + U) -- the expression of gx_rhs_kernel on
// the values it has just stored -- so only the first round of a sub-problem launches gx_rhs_kernel (r3: one launch and two array
// reads less per round).
__global__ __launch_bounds__(256) void gx_prox_kernel(const float* __restrict__ aux, float* __restrict__ X, float* __restrict__ U, int prox, float lam,
                                                      int64_t count4, double* __restrict__ part, const DevState* __restrict__ st,
                                                      const float* __restrict__ Bm = nullptr, float* __restrict__ rhs = nullptr)
{
    if (st->flag || st->inner_stop) return;
    __shared__ double sh[4][4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float shift = (prox == NMFX_PROX_L1N) ? (float)((double)lam / st->rho) : 0.f;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
    if (i < count4) {
        const float4 a4 = reinterpret_cast<const float4*>(aux)[i], x4 = reinterpret_cast<const float4*>(X)[i], u4 = reinterpret_cast<const float4*>(U)[i];
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w};
        float xn[4], un[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = (av[e] - uv[e]) - shift;
            xn[e] = (d < 0.f) ? 0.f : d;
            un[e] = uv[e] + xn[e] - av[e];
            const float d0 = xn[e] - av[e], d2 = xn[e] - xv[e];
            n0 += d0 * d0; n1 += xn[e] * xn[e]; n2 += d2 * d2; n3 += un[e] * un[e];
        }
        reinterpret_cast<float4*>(X)[i] = make_float4(xn[0], xn[1], xn[2], xn[3]);
        reinterpret_cast<float4*>(U)[i] = make_float4(un[0], un[1], un[2], un[3]);
        if (Bm) {
            const float rho = (float)st->rho;
            const float4 b = reinterpret_cast<const float4*>(Bm)[i];
            reinterpret_cast<float4*>(rhs)[i] = make_float4(b.x + rho * (xn[0] + un[0]), b.y + rho * (xn[1] + un[1]), b.z + rho * (xn[2] + un[2]),
                                                            b.w + rho * (xn[3] + un[3]));
        }
    }
    double v[4] = {(double)n0, (double)n1, (double)n2, (double)n3};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][c] = v[c];
    }
    __syncthreads();
    if (threadIdx.x < 4) part[(int64_t)blockIdx.x * 4 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// terminate (ao_admm.py:33-43) on the sums of this round: ||X - aux|| / ||X|| < 1e-2 and ||X - X_prev|| / ||U|| < 1e-2, as
// a < 1e-4 b on the sums of squares (a zero denominator gives inf / nan in the reference: both compare False, as here)
__global__ __launch_bounds__(256) void gx_decide_kernel(const double* __restrict__ part, int nblk, DevState* __restrict__ st)
{
    if (st->flag || st->inner_stop) return;
    __shared__ double sh[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = tid; b < nblk; b += 256)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += part[(int64_t)b * 4 + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
        if (lane == 0) sh[wave][c] = v[c];
    }
    __syncthreads();
    if (tid == 0) {
        double n[4];
        for (int c = 0; c < 4; ++c) n[c] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
        st->inner_count += 1;
        if (n[0] < 1e-4 * n[1] && n[2] < 1e-4 * n[3]) st->inner_stop = 1;
    }
}

// end of a sub-problem: rounds run | fired << 16 (the slot nmfx_get_inner_counts reads)
__global__ void gx_close_kernel(DevState* __restrict__ st, int32_t* __restrict__ slot)
{
    if (st->flag) return;
    *slot = st->inner_count | (st->inner_stop << 16);
    st->inner_stop = 0;
}

// M^-1 = (G + rho I)^-1 -> E->Minv, st->rho (fixed_rho < 0: trace(G) / k), the inner-round state reset
// ---- the same inverse by BLOCKS of 128 (r3) --------------------------------------------------------------------------------------
// The one-workgroup kernel above walks the whole kp x kp f64 matrix in global memory once per pivot: 6.0 ms at kp = 256, twice per
// outer iteration -- 12 of the 13.8 ms of an AO-ADMM iteration at 16384 x 8192.  Block Gauss-Jordan over 128-wide blocks instead:
// per block step p the diagonal block T[p][p] is inverted by the k = 128 solvers' blocked f64-MFMA kernel (kernels_aoadmm.hip,
// ao_prepare_mfma_kernel: one workgroup, ~40 us; its pivots are those of the unblocked elimination, so "not positive definite"
// fires on the same condition), and the rest of the step is f64 matrix products spread over the chip (v_mfma_f64_16x16x4_f64, one
// 16 x 16 tile per wave, operands straight from the L2-resident matrices):
//     C = -T[:, p] D^-1;     T'[p][p] = D^-1,  T'[p][J] = D^-1 T[p][J],  T'[I][p] = C[I],  T'[I][J] = T[I][J] + C[I] T[p][J]
// from the old matrix into the other of two buffers.  3 launches per block step, kp / 128 steps.
__device__ __forceinline__ gx_f64x4 gx_tile64(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                                              gx_f64x4 acc, int c, int q)
{   // acc[r] (row q + 4 r, column c) += sum_t A[row][t] B[t][column], t < 128
#pragma unroll 8
    for (int s4 = 0; s4 < 32; ++s4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(int64_t)c * lda + 4 * s4 + q], B[(int64_t)(4 * s4 + q) * ldb + c], acc, 0, 0, 0);
    return acc;
}

// one block per row of T = G + rho I; every block takes the trace itself (k values), block 0 publishes rho and opens the sub-problem
// (r4: as ONE block of 1024 threads, with a 64-bit division per element, this copy took 25 us of the 134 us chain at kp = 256)
__global__ __launch_bounds__(256) void gx_bgj_init_kernel(const float* __restrict__ G, int kp, int k, double* __restrict__ T,
                                                          DevState* __restrict__ st, double fixed_rho)
{
    if (st->flag) return;
    __shared__ double part[4];
    const int tid = threadIdx.x, row = blockIdx.x;
    double tr = 0.0;
    for (int i = tid; i < k; i += 256) tr += (double)G[(int64_t)i * kp + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tr += __shfl_down(tr, off, 64);
    if ((tid & 63) == 0) part[tid >> 6] = tr;
    __syncthreads();
    double rho = (((part[0] + part[1]) + part[2]) + part[3]) / (double)k;
    if (fixed_rho >= 0.0) rho = fixed_rho;
    for (int c = tid; c < kp; c += 256) T[(int64_t)row * kp + c] = (double)G[(int64_t)row * kp + c] + (c == row ? rho : 0.0);
    if (row == 0 && tid == 0) { st->rho = rho; st->inner_stop = 0; st->inner_count = 0; }
}

// C[kp][128] = -T[:, block p] D^-1
__global__ __launch_bounds__(256) void gx_bgj_col_kernel(const double* __restrict__ T, int kp, int p, const double* __restrict__ Dinv,
                                                         double* __restrict__ C, const DevState* __restrict__ st)
{
    if (st->flag) return;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), it = tile >> 3, jt = tile & 7;
    gx_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = gx_tile64(T + (int64_t)16 * it * kp + 128 * p, kp, Dinv + 16 * jt, 128, acc, c, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(int64_t)(16 * it + q + 4 * r) * 128 + 16 * jt + c] = -acc[r];
}

__global__ __launch_bounds__(256) void gx_bgj_update_kernel(const double* __restrict__ T, double* __restrict__ Tn, int kp, int p,
                                                            const double* __restrict__ Dinv, const double* __restrict__ C,
                                                            const DevState* __restrict__ st)
{
    if (st->flag) return;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int nt16 = kp / 16;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), it = tile / nt16, jt = tile % nt16;
    const int I = it >> 3, J = jt >> 3;
    gx_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    if (I == p && J == p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Dinv[(16 * (it - 8 * p) + q + 4 * r) * 128 + 16 * (jt - 8 * p) + c];
    } else if (J == p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = C[(int64_t)(16 * it + q + 4 * r) * 128 + 16 * (jt - 8 * p) + c];
    } else if (I == p) {
        acc = gx_tile64(Dinv + 16 * (it - 8 * p) * 128, 128, T + (int64_t)128 * p * kp + 16 * jt, kp, acc, c, q);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = T[(int64_t)(16 * it + q + 4 * r) * kp + 16 * jt + c];
        acc = gx_tile64(C + (int64_t)16 * it * 128, 128, T + (int64_t)128 * p * kp + 16 * jt, kp, acc, c, q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Tn[(int64_t)(16 * it + q + 4 * r) * kp + 16 * jt + c] = acc[r];
}

__global__ __launch_bounds__(256) void gx_bgj_finish_kernel(const double* __restrict__ T, int64_t kk, float* __restrict__ Minv,
                                                            const DevState* __restrict__ st)
{
    if (st->flag) return;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < kk) Minv[e] = (float)T[e];
}

// work area gx_w64: two kp x kp matrices, D^-1 [128][128], C [kp][128]
int64_t gx_w64_count(int64_t kp) { return 2 * kp * kp + 128 * 128 + kp * 128; }

int gx_prepare(nmfx_engine* E, const float* G, double fixed_rho) {
    ProfScope ps(E, "prepare");
    static const bool scalar = getenv("NMFX_PREPARE_SCALAR") != nullptr;      // the one-workgroup kernel (A/B runs)
    if (!scalar) {
        int rc;
        const int kp = (int)E->kp, nb = kp / 128;
        const int64_t kk = (int64_t)kp * kp;
        double* Tb[2] = {E->gx_w64, E->gx_w64 + kk};
        double* Dinv = E->gx_w64 + 2 * kk;
        double* C = Dinv + 128 * 128;
        hipLaunchKernelGGL(gx_bgj_init_kernel, dim3((unsigned)kp), dim3(256), 0, E->stream, G, kp, E->k, Tb[0], E->state, fixed_rho);
        NMFX_HIP(hipGetLastError());
        for (int p = 0; p < nb; ++p) {
            const double* T = Tb[p & 1];
            if ((rc = nmfx_launch_inverse64_block(E, T + (int64_t)128 * p * kp + 128 * p, kp, Dinv))) return rc;
            hipLaunchKernelGGL(gx_bgj_col_kernel, dim3((unsigned)(kp / 16 * 8 / 4)), dim3(256), 0, E->stream, T, kp, p, (const double*)Dinv, C,
                               (const DevState*)E->state);
            hipLaunchKernelGGL(gx_bgj_update_kernel, dim3((unsigned)((kp / 16) * (kp / 16) / 4)), dim3(256), 0, E->stream, T, Tb[(p + 1) & 1], kp, p,
                               (const double*)Dinv, (const double*)C, (const DevState*)E->state);
            NMFX_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(gx_bgj_finish_kernel, dim3((unsigned)((kk + 255) / 256)), dim3(256), 0, E->stream, (const double*)Tb[nb & 1], kk, E->Minv,
                           (const DevState*)E->state);
        NMFX_HIP(hipGetLastError());
        return NMFX_OK;
    }
    const size_t shm = (size_t)(2 * E->kp + 16) * sizeof(double);
    int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gx_prepare_kernel), (int)shm); if (rc) return rc;
    hipLaunchKernelGGL(gx_prepare_kernel, dim3(1), dim3(1024), shm, E->stream, G, (int)E->kp, E->k, E->gx_w64, E->Minv, E->state, fixed_rho);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// one sub-problem (ao_admm.py:46-68) on the factor X ([rows][cols] with the factor index along `cols` for W: rows = m,
// cols = kp -- or along `rows` for H: rows = kp, cols = n), its dual U, the Gram matrix G and the cross product B
int gx_ao_subproblem(nmfx_engine* E, bool hside, const float* G, const float* B, float* X, float* U, int prox, float lam, int admm_iter,
                     int32_t* slot) {
    int rc;
    const int64_t kp = E->kp, rows = hside ? kp : E->mp, cols = hside ? E->np : kp, cnt4 = rows * cols / 4;
    if ((rc = gx_prepare(E, G, -1.0))) return rc;
    ProfScope ps(E, hside ? "inner_h" : "inner_w");
    static const bool one_launch = !(getenv("NMFX_GX_ROUNDS") && atoi(getenv("NMFX_GX_ROUNDS")) == 4);
    if (one_launch && kp <= 512) {                     // r4: one launch per round (kernels_aoadmm.hip, ao_round_*_any_kernel)
        for (int r = 0; r < admm_iter; ++r)
            if ((rc = nmfx_round_any(E, hside, B, X, U, prox, lam, r))) return rc;
        return nmfx_inner_finish(E, (int)((hside ? E->np : E->mp) / 64), admm_iter, slot);
    }
    const int nblk = (int)((cnt4 + 255) / 256);
    const int* stop = &E->state->inner_stop;
    for (int r = 0; r < admm_iter; ++r) {
        if (r == 0) {                                  // (later rounds: the prox launch of the round before has left the right-hand side)
            hipLaunchKernelGGL(gx_rhs_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, B, (const float*)X, (const float*)U, E->gx_r, cnt4,
                               (const DevState*)E->state);
            NMFX_HIP(hipGetLastError());
        }
        if (hside) rc = gx_launch<true, false>(E, GX_STORE, E->Minv, kp, E->gx_r, cols, E->gx_d, cols, 0, kp, cols, kp, 1, nullptr, 0, nullptr, stop);
        else rc = gx_launch<true, false>(E, GX_STORE, E->gx_r, kp, E->Minv, kp, E->gx_d, kp, 0, rows, kp, kp, 1, nullptr, 0, nullptr, stop);
        if (rc) return rc;
        hipLaunchKernelGGL(gx_prox_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, (const float*)E->gx_d, X, U, prox, lam, cnt4, E->gx_nrm,
                           (const DevState*)E->state, r + 1 < admm_iter ? B : (const float*)nullptr, E->gx_r);
        hipLaunchKernelGGL(gx_decide_kernel, dim3(1), dim3(256), 0, E->stream, (const double*)E->gx_nrm, nblk, E->state);
        NMFX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gx_close_kernel, dim3(1), dim3(1), 0, E->stream, E->state, slot);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int gx_objective_partial(nmfx_engine* E, bool bf = false) {      // 1/2 ||V - W H||^2 of the current pair -> xf64[0]  (bf: from the images of W[0] and H^T)
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    { ProfScope ps(E, "objective");
      if (bf) rc = gxb_launch(E, GX_RESID, E->Whi[0], E->Wlo[0], kp, E->HThi, E->HTlo, kp, nullptr, 0, 0, mp, np, kp, 1, E->V, np, E->gx_part);
      else rc = gx_launch<true, false>(E, GX_RESID, E->W[0], kp, E->H, np, nullptr, 0, 0, mp, np, kp, 1, E->V, np, E->gx_part);
      if (rc) return rc; }
    return nmfx_launch_obj_reduce(E, (mp / GX_T) * (np / GX_T), E->gx_part);
}

}  // namespace

int nmfx_generic_aoadmm_run(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h, int admm_iter, int64_t min_iter,
                            double tol1, double tol2, int64_t first, int64_t count) {
    E->gxb_img_ready = false;                          // (this solver rewrites W and H without the images of the k > 128 MUR loop)
    int rc;
    if ((rc = gx_buffers(E, false))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    if ((rc = gx_alloc(E, &E->gx_r, std::max(mp, np) * kp))) return rc;
    if ((rc = gx_alloc(E, &E->gx_w64, gx_w64_count(kp)))) return rc;
    if ((rc = gx_alloc(E, &E->gx_nrm, std::max(mp, np) * kp / 1024 * 4 + 64))) return rc;
    float* W = E->W[0];
    float* xB = E->xf32;
    float* xG = E->xf32 + kp * np;
    // split-bf16 runs: the three V-sized products (W^T V, V H^T, the objective's W H) from the V planes and the factor images, three
    // terms as in the tuned AO-ADMM (kernels_bf16.hip, top); the Gram matrices, whose shifted inverse the rounds apply, stay exact f32
    bool bf = gxb_on(E);
    if (bf) { rc = gxb_prepare(E, W); if (rc == GXB_NOFIT) bf = false; else if (rc) return rc; }
    E->gxb_img_ready = false;                          // (valid inside this call only: the MUR loop keeps the images of W[j & 1])
    if (first == 0 && count > 0 && (rc = gx_objective_partial(E, bf))) return rc;      // obj[0] (ao_admm.py:256)
    for (int64_t j = first; j < first + count; ++j) {
        hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                           E->state, E->obj_hist);
        NMFX_HIP(hipGetLastError());
        // H sub-problem: G = W^T W, B = W^T V
        { ProfScope ps(E, "gram_tn");
          if ((rc = gx_split_product<false, false>(E, W, kp, W, kp, xG, kp, kp, mp, 64))) return rc; }
        { ProfScope ps(E, "hphase");
          if (bf) rc = gxt_split_product(E, E->WThi, E->WTlo, E->gxb_v[2], E->gxb_v[3], xB, kp, np, mp, 8);
          else rc = gx_split_product<false, false>(E, W, kp, E->V, np, xB, kp, np, mp, 8);
          if (rc) return rc; }
        if ((rc = gx_ao_subproblem(E, true, xG, xB, E->H, E->dualH, prox_h, (float)lam_h, admm_iter, E->inner_hist + j * 2))) return rc;
        if (bf) { ProfScope ps(E, "images");
                  if ((rc = gxb_images_h(E, E->H))) return rc; }
        // W sub-problem on the transposed data: G = H H^T, B^T = V H^T
        { ProfScope ps(E, "gram_nt");
          if ((rc = gx_split_product<true, true>(E, E->H, np, E->H, np, E->HHt, kp, kp, np, 64))) return rc; }
        { ProfScope ps(E, "wphase");
          if (bf) rc = gxt_split_product(E, E->gxb_v[0], E->gxb_v[1], E->Hhi, E->Hlo, E->A_part, mp, kp, np, 4);
          else rc = gx_split_product<true, true>(E, E->V, np, E->H, np, E->A_part, mp, kp, np, 1);
          if (rc) return rc; }
        if ((rc = gx_ao_subproblem(E, false, E->HHt, E->A_part, W, E->dualW, prox_w, (float)lam_w, admm_iter, E->inner_hist + j * 2 + 1))) return rc;
        if (bf) { ProfScope ps(E, "images");
                  if ((rc = gxb_images_w(E, W))) return rc; }
        if ((rc = gx_objective_partial(E, bf))) return rc;
    }
    return NMFX_OK;
}

// (r5) the W side of a row-sharded sub-problem: a round's norm sums of the rank's rows for the caller's all-reduce, and the stop decision
// of the round before from the all-reduced sums
namespace {
__global__ __launch_bounds__(256) void gx_gather_norms_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out, const DevState* __restrict__ st)
{
    __shared__ double sh[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (!(st->flag || st->inner_stop)) {               // (behind the inner stop nothing is read from the sums: zeros keep the exchange finite)
        for (int b = tid; b < nblk; b += 256)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += part[(int64_t)b * 4 + c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
        if (lane == 0) sh[wave][c] = v[c];
    }
    __syncthreads();
    if (tid < 4) out[tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
}

// terminate (ao_admm.py:33-43) of the round before from the sums over ALL ranks' rows
__global__ void gx_decide_global_kernel(const double* __restrict__ n, DevState* __restrict__ st)
{
    if (st->flag || st->inner_stop) return;
    st->inner_count += 1;
    if (n[0] < 1e-4 * n[1] && n[2] < 1e-4 * n[3]) st->inner_stop = 1;
}
}  // namespace

// The same iteration in the pieces a ROW-SHARDED run needs (r4; nmfx_aoadmm_phase_* dispatch here beyond 128 components, k <= 512 with one launch per W round,
// beyond that the rhs / solve / prox launches of gx_ao_subproblem):
//   0 h_products: [W^T V | W^T W] of this rank's rows into the f32 exchange buffer (xf64[0] holds the objective partial the
//                 closing phase of the iteration before -- or, at j = 0, this phase -- left)      -> all-reduce f32 + f64[:8]
//   1 h_solve:    obj[j] and the stop rule, then the H sub-problem (replicated work on the all-reduced sums)
//   2 w_products: H H^T (replicated), V H^T of the rank's rows, (H H^T + rho I)^-1
//   3 w_round:    one round of the W sub-problem on the rank's rows (decision of the round before from the ALL-REDUCED norm sums
//                 xf64[1..4]), this rank's norm sums of the round -> xf64[1..4]                   -> all-reduce f64[1:5]
//   4 w_close:    inner-round bookkeeping, the objective partial of the new pair -> xf64[0]
int nmfx_generic_aoadmm_phase(nmfx_engine* E, int phase, int prox, double lam, int admm_iter, int64_t min_iter, double tol1, double tol2,
                              int64_t j, int round) {
    int rc;
    E->gxb_img_ready = false;
    if ((rc = gx_buffers(E, false))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    if ((rc = gx_alloc(E, &E->gx_r, std::max(mp, np) * kp))) return rc;
    if ((rc = gx_alloc(E, &E->gx_w64, gx_w64_count(kp)))) return rc;
    if ((rc = gx_alloc(E, &E->gx_nrm, std::max(mp, np) * kp / 1024 * 4 + 64))) return rc;
    float* W = E->W[0];
    float* xB = E->xf32;
    float* xG = E->xf32 + kp * np;
    bool bf = gxb_on(E);
    // The bf16 images of W and H are READ by the products of phases 0 and 2 only; phases 1 and 4 rebuild the images of the factor they
    // have updated.  So they are (re)built from the factors at the head of every outer iteration (phase 0: whatever ran on this handle
    // before, the images are those of the current pair) and left alone in the other phases -- r4 rebuilt both factors' images in EVERY
    // phase call, the admm_iter W rounds of phase 3 included.  The planes of V and the buffers are made on first use.
    if (phase != 0) E->gxb_img_ready = true;
    if (bf) { rc = gxb_prepare(E, W); if (rc == GXB_NOFIT) bf = false; else if (rc) return rc; }
    E->gxb_img_ready = false;
    switch (phase) {
    case 0:
        if (j == 0 && (rc = gx_objective_partial(E, bf))) return rc;              // obj[0] (ao_admm.py:256)
        { ProfScope ps(E, "gram_tn");
          if ((rc = gx_split_product<false, false>(E, W, kp, W, kp, xG, kp, kp, mp, 64))) return rc; }
        { ProfScope ps(E, "hphase");
          if (bf) rc = gxt_split_product(E, E->WThi, E->WTlo, E->gxb_v[2], E->gxb_v[3], xB, kp, np, mp, 8);
          else rc = gx_split_product<false, false>(E, W, kp, E->V, np, xB, kp, np, mp, 8);
          return rc; }
    case 1:
        hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                           E->state, E->obj_hist);
        NMFX_HIP(hipGetLastError());
        if ((rc = gx_ao_subproblem(E, true, xG, xB, E->H, E->dualH, prox, (float)lam, admm_iter, E->inner_hist + j * 2))) return rc;
        if (bf) { ProfScope ps(E, "images");
                  if ((rc = gxb_images_h(E, E->H))) return rc; }
        return NMFX_OK;
    case 2:
        { ProfScope ps(E, "gram_nt");
          if ((rc = gx_split_product<true, true>(E, E->H, np, E->H, np, E->HHt, kp, kp, np, 64))) return rc; }
        { ProfScope ps(E, "wphase");
          if (bf) rc = gxt_split_product(E, E->gxb_v[0], E->gxb_v[1], E->Hhi, E->Hlo, E->A_part, mp, kp, np, 4);
          else rc = gx_split_product<true, true>(E, E->V, np, E->H, np, E->A_part, mp, kp, np, 1);
          if (rc) return rc; }
        return gx_prepare(E, E->HHt, -1.0);
    case 3:
        if (kp > 512) {                                // (r5) the launches of gx_ao_subproblem's long form, with the decision taken from the all-reduced sums
            const int64_t cnt4 = mp * kp / 4;
            const int nblk = (int)((cnt4 + 255) / 256);
            const int* stop = &E->state->inner_stop;
            if (round > 0) hipLaunchKernelGGL(gx_decide_global_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)(E->xf64 + 1), E->state);
            else hipLaunchKernelGGL(gx_rhs_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, (const float*)E->A_part, (const float*)W, (const float*)E->dualW,
                                    E->gx_r, cnt4, (const DevState*)E->state);
            NMFX_HIP(hipGetLastError());
            if ((rc = gx_launch<true, false>(E, GX_STORE, E->gx_r, kp, E->Minv, kp, E->gx_d, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr, stop))) return rc;
            hipLaunchKernelGGL(gx_prox_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, (const float*)E->gx_d, W, E->dualW, prox, (float)lam, cnt4, E->gx_nrm,
                               (const DevState*)E->state, (const float*)E->A_part, E->gx_r);      // (leaves the next round's right-hand side)
            hipLaunchKernelGGL(gx_gather_norms_kernel, dim3(1), dim3(256), 0, E->stream, (const double*)E->gx_nrm, nblk, E->xf64 + 1, (const DevState*)E->state);
            NMFX_HIP(hipGetLastError());
            return NMFX_OK;
        }
        if ((rc = nmfx_round_any(E, false, E->A_part, W, E->dualW, prox, (float)lam, round, round > 0 ? E->xf64 + 1 : nullptr))) return rc;
        return nmfx_gather_round_norms(E, (int)(mp / 64), round);
    case 4:
        if (kp > 512) {
            if (admm_iter > 0) hipLaunchKernelGGL(gx_decide_global_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)(E->xf64 + 1), E->state);
            hipLaunchKernelGGL(gx_close_kernel, dim3(1), dim3(1), 0, E->stream, E->state, E->inner_hist + j * 2 + 1);
            NMFX_HIP(hipGetLastError());
        } else
        if ((rc = nmfx_inner_finish(E, (int)(mp / 64), admm_iter, E->inner_hist + j * 2 + 1, E->xf64 + 1))) return rc;
        if (bf) { ProfScope ps(E, "images");
                  if ((rc = gxb_images_w(E, W))) return rc; }
        return gx_objective_partial(E, bf);
    default:
        E->err = "generic AO-ADMM phase: 0 .. 4"; return NMFX_E_ARG;
    }
}

// ---- AO-ADMM with the KL loss (nmf/ao_admm.py:71-101, 274-289) and ADMM (nmf/admm.py:216-230, 292-334) for k > 128 --------------------
namespace {

// out = a - b
__global__ __launch_bounds__(256) void gx_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t count4,
                                                      const int* __restrict__ flag)
{
    if (*flag) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count4) return;
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
}

// prox 'l2n' behind its operator product (admm.py:141-156): X = max(P (aux - U), 0) = max(d, 0);  U += X - aux (admm.py:321-322)
__global__ __launch_bounds__(256) void gx_l2n_finish_kernel(const float* __restrict__ d, const float* __restrict__ aux, float* __restrict__ X,
                                                            float* __restrict__ U, int64_t count4, const int* __restrict__ flag)
{
    if (*flag) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count4) return;
    const float4 d4 = reinterpret_cast<const float4*>(d)[i], a4 = reinterpret_cast<const float4*>(aux)[i];
    float4 u4 = reinterpret_cast<float4*>(U)[i];
    const float4 x4 = make_float4(d4.x < 0.f ? 0.f : d4.x, d4.y < 0.f ? 0.f : d4.y, d4.z < 0.f ? 0.f : d4.z, d4.w < 0.f ? 0.f : d4.w);
    u4.x += x4.x - a4.x; u4.y += x4.y - a4.y; u4.z += x4.z - a4.z; u4.w += x4.w - a4.w;
    reinterpret_cast<float4*>(X)[i] = x4;
    reinterpret_cast<float4*>(U)[i] = u4;
}

int gx_kl_objective_partial(nmfx_engine* E) {        // KL(V, W H) (utils.py:21-26) of the current pair -> xf64[0]; the quotient is not stored
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    { ProfScope ps(E, "objective");
      if ((rc = gx_launch<true, false>(E, GX_KLQ, E->W[0], kp, E->H, np, nullptr, np, 0, mp, np, kp, 1, E->V, np, E->gx_part))) return rc; }
    return nmfx_launch_obj_reduce(E, (mp / GX_T) * (np / GX_T), E->gx_part);
}

int gx_admm_buffers(nmfx_engine* E) {
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    if ((rc = gx_buffers(E, false))) return rc;
    if ((rc = gx_alloc(E, &E->gx_r, std::max(mp, np) * kp))) return rc;
    if ((rc = gx_alloc(E, &E->gx_w64, gx_w64_count(kp)))) return rc;
    return gx_alloc(E, &E->gx_nrm, std::max(mp, np) * kp / 1024 * 4 + 64);
}

// one KL sub-problem (ao_admm.py:71-101): hside: X = H ([kp][np]), other factor W, data S = v_aux + dual_v; else X = W on the
// transposed data.  G = the other factor's Gram matrix.
int gx_ao_kl_subproblem(nmfx_engine* E, bool hside, const float* G, int prox, float lam, int admm_iter, int32_t* slot) {
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp, rows = hside ? kp : mp, cols = hside ? np : kp, cnt4 = rows * cols / 4;
    float* W = E->W[0];
    float* X = hside ? E->H : W;
    float* U = hside ? E->dualH : E->dualW;
    if ((rc = gx_prepare(E, G, -1.0))) return rc;
    ProfScope ps(E, hside ? "inner_h" : "inner_w");
    const int nblk = (int)((cnt4 + 255) / 256);
    const int* stop = &E->state->inner_stop;
    for (int r = 0; r < admm_iter; ++r) {
        // right-hand side product with the CURRENT S (ao_admm.py:85)
        if (hside) rc = gx_split_product<false, false>(E, W, kp, E->S, np, E->xf32, kp, np, mp, 8, stop);
        else rc = gx_split_product<true, true>(E, E->S, np, E->H, np, E->A_part, mp, kp, np, 1, stop);
        if (rc) return rc;
        hipLaunchKernelGGL(gx_rhs_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, hside ? (const float*)E->xf32 : (const float*)E->A_part,
                           (const float*)X, (const float*)U, E->gx_r, cnt4, (const DevState*)E->state);
        NMFX_HIP(hipGetLastError());
        if (hside) rc = gx_launch<true, false>(E, GX_STORE, E->Minv, kp, E->gx_r, np, E->gx_d, np, 0, kp, np, kp, 1, nullptr, 0, nullptr, stop);
        else rc = gx_launch<true, false>(E, GX_STORE, E->gx_r, kp, E->Minv, kp, E->gx_d, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr, stop);
        if (rc) return rc;
        hipLaunchKernelGGL(gx_prox_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, (const float*)E->gx_d, X, U, prox, lam, cnt4, E->gx_nrm,
                           (const DevState*)E->state);
        NMFX_HIP(hipGetLastError());
        // v_aux, dual_v from P = (other factor) x (this round's aux) (ao_admm.py:90-95): the aux matrix is gx_d
        if (hside) rc = gx_launch<true, false>(E, GX_VAUX, W, kp, E->gx_d, np, E->DV, np, 0, mp, np, kp, 1, E->V, np, nullptr, stop, E->S);
        else rc = gx_launch<true, false>(E, GX_VAUX, E->gx_d, kp, E->H, np, E->DV, np, 0, mp, np, kp, 1, E->V, np, nullptr, stop, E->S);
        if (rc) return rc;
        hipLaunchKernelGGL(gx_decide_kernel, dim3(1), dim3(256), 0, E->stream, (const double*)E->gx_nrm, nblk, E->state);
        NMFX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gx_close_kernel, dim3(1), dim3(1), 0, E->stream, E->state, slot);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// X = prox(aux, U), U += X - aux for one side of the ADMM state (admm.py:117-156, 321-322)
int gx_admm_prox(nmfx_engine* E, bool hside, int prox, double rho, double lam) {
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp, cnt4 = (hside ? kp * np : mp * kp) / 4;
    const int nblk = (int)((cnt4 + 255) / 256);
    const float* aux = hside ? E->auxH : E->auxW;
    float* X = hside ? E->H : E->W[0];
    float* U = hside ? E->dualH : E->dualW;
    const int* flag = &E->state->flag;
    ProfScope ps(E, hside ? "inner_h" : "inner_w");
    if (prox == NMFX_PROX_NN || prox == NMFX_PROX_L1N) {
        hipLaunchKernelGGL(gx_prox_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, aux, X, U, prox, (float)lam, cnt4, E->gx_nrm, (const DevState*)E->state);
        NMFX_HIP(hipGetLastError());
        return NMFX_OK;
    }
    if (prox == NMFX_PROX_L2N) {                       // P is symmetric ((lambda T^T T + rho I) / rho)^-1: W-like operands take it from the right
        hipLaunchKernelGGL(gx_diff_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, aux, (const float*)U, E->gx_r, cnt4, flag);
        NMFX_HIP(hipGetLastError());
        if (hside) rc = gx_launch<true, false>(E, GX_STORE, E->Ph, kp, E->gx_r, np, E->gx_d, np, 0, kp, np, kp, 1, nullptr, 0, nullptr);
        else rc = gx_launch<true, false>(E, GX_STORE, E->gx_r, kp, E->Pw, kp, E->gx_d, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL(gx_l2n_finish_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, (const float*)E->gx_d, aux, X, U, cnt4, flag);
        NMFX_HIP(hipGetLastError());
        return NMFX_OK;
    }
    if (prox == NMFX_PROX_L1INF || prox == NMFX_PROX_L1INF_T)      // (r4: 'l1inf_transpose' takes one workgroup per column beyond 128 entries)
        return nmfx_launch_prox_l1inf(E, hside, prox == NMFX_PROX_L1INF_T, rho, lam, 1.0, true);
    E->err = "Unknown prox_type.";
    return NMFX_E_ARG;
}

}  // namespace

int nmfx_generic_aoadmm_kl_run(nmfx_engine* E, int prox_w, double lam_w, int prox_h, double lam_h, int admm_iter, int64_t min_iter,
                               double tol1, double tol2, int64_t first, int64_t count) {
    E->gxb_img_ready = false;                          // (this solver rewrites W and H without the images of the k > 128 MUR loop)
    int rc;
    if ((rc = gx_admm_buffers(E))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    float* W = E->W[0];
    float* xG = E->xf32 + kp * np;
    if (first == 0 && count > 0 && (rc = gx_kl_objective_partial(E))) return rc;   // obj[0] (ao_admm.py:256)
    for (int64_t j = first; j < first + count; ++j) {
        hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                           E->state, E->obj_hist);
        NMFX_HIP(hipGetLastError());
        { ProfScope ps(E, "gram_tn");
          if ((rc = gx_split_product<false, false>(E, W, kp, W, kp, xG, kp, kp, mp, 64))) return rc; }
        if ((rc = gx_ao_kl_subproblem(E, true, xG, prox_h, (float)lam_h, admm_iter, E->inner_hist + j * 2))) return rc;
        { ProfScope ps(E, "gram_nt");
          if ((rc = gx_split_product<true, true>(E, E->H, np, E->H, np, E->HHt, kp, kp, np, 64))) return rc; }
        if ((rc = gx_ao_kl_subproblem(E, false, E->HHt, prox_w, (float)lam_w, admm_iter, E->inner_hist + j * 2 + 1))) return rc;
        if ((rc = gx_kl_objective_partial(E))) return rc;
    }
    return NMFX_OK;
}

// The KL-loss iteration beyond 128 components in the pieces a ROW-SHARDED run needs (r5; nmfx_aoadmm_kl_phase_* dispatch here): the
// V-sized products sit inside the inner rounds (ao_admm.py:85), so the H sub-problem exchanges once per round and the W sub-problem
// its four norm sums per round --
//   0 h_products(j, r): r = 0: (j = 0: the KL objective partial of the initial pair -> xf64[0]) W^T W of the rank's rows -> xf32 behind
//                       the k n part; every r: W^T S of the rank's rows -> xf32                  -> all-reduce f32 (r = 0: and f64[:8])
//   1 h_round(r):       r = 0: obj[j] + stop rule, (W^T W + rho I)^-1; then the replicated solve / prox / dual step, the rank-local
//                       v_aux / dual_v step and the stop decision of the round (H is replicated: its norms are whole)
//   2 h_close:          inner-round bookkeeping, H H^T and its inverse (replicated)
//   3 w_round(r):       r > 0: the stop decision of round r - 1 from the ALL-REDUCED norm sums xf64[1..4]; S H^T, solve, prox, duals and
//                       v_aux / dual_v on the rank's rows; the rank's norm sums of the round -> xf64[1..4]   -> all-reduce f64[1:5]
//   4 w_close:          the decision of the last round, bookkeeping, the KL objective partial of the new pair -> xf64[0]
int nmfx_generic_aoadmm_kl_phase(nmfx_engine* E, int phase, int prox, double lam, int admm_iter, int64_t min_iter, double tol1, double tol2,
                                 int64_t j, int round) {
    E->gxb_img_ready = false;
    int rc;
    if ((rc = gx_admm_buffers(E))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    float* W = E->W[0];
    float* xG = E->xf32 + kp * np;
    const int* stop = &E->state->inner_stop;
    const bool hside = phase <= 2;
    const int64_t cnt4 = (hside ? kp * np : mp * kp) / 4;
    const int nblk = (int)((cnt4 + 255) / 256);
    float* X = hside ? E->H : W;
    float* U = hside ? E->dualH : E->dualW;
    switch (phase) {
    case 0:
        if (round == 0) {
            if (j == 0 && (rc = gx_kl_objective_partial(E))) return rc;            // obj[0] (ao_admm.py:256)
            ProfScope ps(E, "gram_tn");
            if ((rc = gx_split_product<false, false>(E, W, kp, W, kp, xG, kp, kp, mp, 64))) return rc;
        }
        { ProfScope ps(E, "hphase");
          return gx_split_product<false, false>(E, W, kp, E->S, np, E->xf32, kp, np, mp, 8, round > 0 ? stop : nullptr); }
    case 3:
        if (round > 0) {
            hipLaunchKernelGGL(gx_decide_global_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)(E->xf64 + 1), E->state);
            NMFX_HIP(hipGetLastError());
        }
        { ProfScope ps(E, "wphase");
          if ((rc = gx_split_product<true, true>(E, E->S, np, E->H, np, E->A_part, mp, kp, np, 1, stop))) return rc; }
        [[fallthrough]];
    case 1: {
        if (phase == 1 && round == 0) {
            hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                               E->state, E->obj_hist);
            NMFX_HIP(hipGetLastError());
            if ((rc = gx_prepare(E, xG, -1.0))) return rc;
        }
        ProfScope ps(E, hside ? "inner_h" : "inner_w");
        hipLaunchKernelGGL(gx_rhs_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, hside ? (const float*)E->xf32 : (const float*)E->A_part,
                           (const float*)X, (const float*)U, E->gx_r, cnt4, (const DevState*)E->state);
        NMFX_HIP(hipGetLastError());
        if (hside) rc = gx_launch<true, false>(E, GX_STORE, E->Minv, kp, E->gx_r, np, E->gx_d, np, 0, kp, np, kp, 1, nullptr, 0, nullptr, stop);
        else rc = gx_launch<true, false>(E, GX_STORE, E->gx_r, kp, E->Minv, kp, E->gx_d, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr, stop);
        if (rc) return rc;
        hipLaunchKernelGGL(gx_prox_kernel, dim3((unsigned)nblk), dim3(256), 0, E->stream, (const float*)E->gx_d, X, U, prox, (float)lam, cnt4, E->gx_nrm,
                           (const DevState*)E->state);
        NMFX_HIP(hipGetLastError());
        if (hside) rc = gx_launch<true, false>(E, GX_VAUX, W, kp, E->gx_d, np, E->DV, np, 0, mp, np, kp, 1, E->V, np, nullptr, stop, E->S);
        else rc = gx_launch<true, false>(E, GX_VAUX, E->gx_d, kp, E->H, np, E->DV, np, 0, mp, np, kp, 1, E->V, np, nullptr, stop, E->S);
        if (rc) return rc;
        if (hside) hipLaunchKernelGGL(gx_decide_kernel, dim3(1), dim3(256), 0, E->stream, (const double*)E->gx_nrm, nblk, E->state);
        else hipLaunchKernelGGL(gx_gather_norms_kernel, dim3(1), dim3(256), 0, E->stream, (const double*)E->gx_nrm, nblk, E->xf64 + 1, (const DevState*)E->state);
        NMFX_HIP(hipGetLastError());
        return NMFX_OK; }
    case 2:
        hipLaunchKernelGGL(gx_close_kernel, dim3(1), dim3(1), 0, E->stream, E->state, E->inner_hist + j * 2);
        NMFX_HIP(hipGetLastError());
        { ProfScope ps(E, "gram_nt");
          if ((rc = gx_split_product<true, true>(E, E->H, np, E->H, np, E->HHt, kp, kp, np, 64))) return rc; }
        return gx_prepare(E, E->HHt, -1.0);
    case 4:
        if (admm_iter > 0) {
            hipLaunchKernelGGL(gx_decide_global_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)(E->xf64 + 1), E->state);
            NMFX_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(gx_close_kernel, dim3(1), dim3(1), 0, E->stream, E->state, E->inner_hist + j * 2 + 1);
        NMFX_HIP(hipGetLastError());
        return gx_kl_objective_partial(E);
    default:
        E->err = "generic AO-ADMM-KL phase: 0 .. 4"; return NMFX_E_ARG;
    }
}

// ADMM (one level, fixed rho): aux_update twice, prox twice, dual updates, KL: v_aux / dual_v from w_aux h_aux (admm.py:292-324).
// The caller (nmfx_admm_run) has allocated the ADMM state and, for the first iteration, set w_aux = w, h_aux = h.
// One ADMM iteration beyond 128 components in its two halves (the row-sharded form exchanges between them, r4):
//   products: [w_aux^T data | w_aux^T w_aux] of this rank's rows into the f32 exchange buffer (xf64[0] holds the objective partial of the
//             pair the update before left -- or, at j = 0, of the initial pair)
//   update:   obj[j] and the stop rule, then everything else of admm.py:292-334 (h_aux and the H half replicated, w_aux, both prox
//             operators, the duals and v_aux row-local), and the objective partial of the new (w, h) -> xf64[0]
struct GxAdmm { bool kl, bf; const float* data; float *xB, *xG; };
static int gx_admm_setup(nmfx_engine* E, int distance, int prox_w, int prox_h, GxAdmm* c) {
    E->gxb_img_ready = false;                          // (this solver rewrites W and H without the images of the k > 128 MUR loop)
    int rc;
    if ((rc = gx_admm_buffers(E))) return rc;
    c->kl = distance == NMFX_KL;
    c->xB = E->xf32;
    c->xG = E->xf32 + (int64_t)E->kp * E->np;
    c->data = c->kl ? E->S : E->V;                     // (KL: v_aux + dual_v, admm.py:224)
    // Euclidean loss, split-bf16 runs: the V-sized products from the V planes and images of the auxiliaries (FOUR terms: the Gram
    // systems carry the caller's fixed rho, kernels_bf16.hip top), the objective's W H from the images of (w, h) (three terms)
    c->bf = !c->kl && gxb_on(E);
    if (c->bf) { rc = gxb_prepare(E, E->W[0]); if (rc == GXB_NOFIT) c->bf = false; else if (rc) return rc; }
    E->gxb_img_ready = false;
    return NMFX_OK;
}

static int gx_admm_products(nmfx_engine* E, const GxAdmm& c) {
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    // h_aux = (w_aux^T w_aux + rho I)^-1 (w_aux^T data + rho (h + dual_h)): the two sums over rows
    { ProfScope ps(E, "gram_tn");
      if ((rc = gx_split_product<false, false>(E, E->auxW, kp, E->auxW, kp, c.xG, kp, kp, mp, 64))) return rc; }
    { ProfScope ps(E, "hphase");
      if (c.bf) {
          if ((rc = gxb_images_w(E, E->auxW))) return rc;
          rc = gxt_split_product(E, E->WThi, E->WTlo, E->gxb_v[2], E->gxb_v[3], c.xB, kp, np, mp, 8, 4);
      } else rc = gx_split_product<false, false>(E, E->auxW, kp, c.data, np, c.xB, kp, np, mp, 8);
      if (rc) return rc; }
    return NMFX_OK;
}

static int gx_admm_update(nmfx_engine* E, const GxAdmm& c, double rho, int prox_w, double lam_w, int prox_h, double lam_h, int64_t min_iter,
                          double tol1, double tol2, int64_t j) {
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    float* W = E->W[0];
    hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                       E->state, E->obj_hist);
    NMFX_HIP(hipGetLastError());
    if ((rc = gx_prepare(E, c.xG, rho))) return rc;
    { ProfScope ps(E, "inner_h");
      const int64_t c4 = kp * np / 4;
      hipLaunchKernelGGL(gx_rhs_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, E->stream, (const float*)c.xB, (const float*)E->H,
                         (const float*)E->dualH, E->gx_r, c4, (const DevState*)E->state);
      NMFX_HIP(hipGetLastError());
      if ((rc = gx_launch<true, false>(E, GX_STORE, E->Minv, kp, E->gx_r, np, E->auxH, np, 0, kp, np, kp, 1, nullptr, 0, nullptr))) return rc; }
    // w_aux^T = (h_aux h_aux^T + rho I)^-1 (h_aux data^T + rho (w^T + dual_w^T)), from the NEW h_aux
    { ProfScope ps(E, "gram_nt");
      if ((rc = gx_split_product<true, true>(E, E->auxH, np, E->auxH, np, E->HHt, kp, kp, np, 64))) return rc; }
    { ProfScope ps(E, "wphase");
      if (c.bf) {
          if ((rc = gxb_images_h(E, E->auxH))) return rc;
          rc = gxt_split_product(E, E->gxb_v[0], E->gxb_v[1], E->Hhi, E->Hlo, E->A_part, mp, kp, np, 4, 4);
      } else rc = gx_split_product<true, true>(E, c.data, np, E->auxH, np, E->A_part, mp, kp, np, 1);
      if (rc) return rc; }
    if ((rc = gx_prepare(E, E->HHt, rho))) return rc;
    { ProfScope ps(E, "inner_w");
      const int64_t c4 = mp * kp / 4;
      hipLaunchKernelGGL(gx_rhs_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, E->stream, (const float*)E->A_part, (const float*)W,
                         (const float*)E->dualW, E->gx_r, c4, (const DevState*)E->state);
      NMFX_HIP(hipGetLastError());
      if ((rc = gx_launch<true, false>(E, GX_STORE, E->gx_r, kp, E->Minv, kp, E->auxW, kp, 0, mp, kp, kp, 1, nullptr, 0, nullptr))) return rc; }
    // h = prox(h_aux, dual_h), w = prox(w_aux, dual_w), duals += x - x_aux
    if ((rc = gx_admm_prox(E, true, prox_h, rho, lam_h))) return rc;
    if ((rc = gx_admm_prox(E, false, prox_w, rho, lam_w))) return rc;
    if (c.kl) {                                        // v_aux, dual_v from w_aux h_aux (admm.py:312-315)
        ProfScope ps(E, "kl_vaux");
        if ((rc = gx_launch<true, false>(E, GX_VAUX, E->auxW, kp, E->auxH, np, E->DV, np, 0, mp, np, kp, 1, E->V, np, nullptr, nullptr, E->S))) return rc;
    }
    if (c.bf) { ProfScope ps(E, "images");            // images of the new (w, h) for the objective
                if ((rc = gxb_images_w(E, W))) return rc;
                if ((rc = gxb_images_h(E, E->H))) return rc; }
    return c.kl ? gx_kl_objective_partial(E) : gx_objective_partial(E, c.bf);
}

int nmfx_generic_admm_run(nmfx_engine* E, int distance, double rho, int prox_w, double lam_w, int prox_h, double lam_h, int64_t min_iter,
                          double tol1, double tol2, int64_t first, int64_t count) {
    GxAdmm c;
    int rc = gx_admm_setup(E, distance, prox_w, prox_h, &c); if (rc) return rc;
    if (first == 0 && count > 0 && (rc = c.kl ? gx_kl_objective_partial(E) : gx_objective_partial(E, c.bf))) return rc;     // (admm.py:289)
    for (int64_t j = first; j < first + count; ++j) {
        if ((rc = gx_admm_products(E, c))) return rc;
        if ((rc = gx_admm_update(E, c, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j))) return rc;
    }
    return NMFX_OK;
}

// row-sharded ADMM beyond 128 components (nmfx_admm_phase_products / _update dispatch here; the W-side prox must be row-local)
int nmfx_generic_admm_phase(nmfx_engine* E, int phase, int distance, double rho, int prox_w, double lam_w, int prox_h, double lam_h,
                            int64_t min_iter, double tol1, double tol2, int64_t j) {
    GxAdmm c;
    int rc = gx_admm_setup(E, distance, prox_w, prox_h, &c); if (rc) return rc;
    if (phase == 0) {
        if (j == 0 && (rc = c.kl ? gx_kl_objective_partial(E) : gx_objective_partial(E, c.bf))) return rc;
        return gx_admm_products(E, c);
    }
    return gx_admm_update(E, c, rho, prox_w, lam_w, prox_h, lam_h, min_iter, tol1, tol2, j);
}

// ---- ANLS (nmf/anls.py:18-47, 111-122) for k > 128 -------------------------------------------------------------------------------------
// The tuned NNLS kernels (kernels_anls.hip) keep one k x k system per wavefront in registers / LDS, which ends at k = 128.  Here one
// WORKGROUP owns one right-hand side and the passive-set system lives in a global f64 work area (L2 / Infinity-Cache resident):
// the same exact active-set method -- block principal pivoting with the back-up rule, warm-started from the support of the previous
// iterate, the same rounding-level infeasibility threshold and pivot guard -- with the passive-set solve done by a right-looking
// Cholesky factorisation in float64 on the gathered system [G_PP; r_P^T] (the extra row makes the forward substitution part of the
// factorisation) and a back substitution through LDS.  Persistent blocks loop over the right-hand sides.
namespace {

#define GX_NNLS_TOL 1e-6
#define GX_NNLS_PIVOT_EPS 1e-6

__device__ __forceinline__ double gx_block_max(double v, double* red, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__device__ __forceinline__ int gx_block_sum_i(int v, int* red, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ int gx_block_max_i(int v, int* red, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return max(max(red[0], red[1]), max(red[2], red[3]));
}

// X[:, c] = argmin_{x >= 0} 1/2 x^T (G + diag_add I) x - r_c^T x for every right-hand side c < nprob; variable i of problem c at
// [i * sj + c * sc] in R and X; X on entry = the warm start (its support).  G is [kp][kp] with zero padding beyond k.
__global__ __launch_bounds__(256) void gx_nnls_kernel(const float* __restrict__ G, int kp, double diag_add, const float* __restrict__ R,
                                                      float* __restrict__ X, int64_t sj, int64_t sc, int64_t nprob, int k,
                                                      double* __restrict__ work, DevState* __restrict__ st)
{
    if (st->flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char gx_nnls_smem[];
    double* zs = reinterpret_cast<double*>(gx_nnls_smem);      // [kp] right-hand side of the back substitution
    double* xs = zs + kp;                                      // [kp] solution by POSITION in the passive list
    double* ldiag = xs + kp;                                   // [kp] diagonal of the Cholesky factor (0: dropped pivot)
    float* rs = reinterpret_cast<float*>(ldiag + kp);          // [kp] right-hand side by variable
    int* plist = reinterpret_cast<int*>(rs + kp);              // [kp] passive variables, ascending
    unsigned char* inF = reinterpret_cast<unsigned char*>(plist + kp);
    unsigned char* dead = inF + kp;
    unsigned char* badf = dead + kp;
    __shared__ double redd[4];
    __shared__ int redi[4];
    __shared__ int sh_p;
    const int tid = threadIdx.x, lane = tid & 63;
    double* M = work + (int64_t)blockIdx.x * ((int64_t)kp + 1) * kp;
    const int cap = 8 * k + 64;
    for (int64_t c = blockIdx.x; c < nprob; c += gridDim.x) {
        __syncthreads();
        double ar = 0.0;
        for (int i = tid; i < k; i += 256) {
            const float rv = R[(int64_t)i * sj + c * sc];
            rs[i] = rv; ar = fmax(ar, fabs((double)rv));
            inF[i] = X[(int64_t)i * sj + c * sc] > 0.f ? 1 : 0;
            dead[i] = 0;
        }
        const double toly = GX_NNLS_TOL * gx_block_max(ar, redd, tid);
        int best = k + 1, spare = 3, p = 0;
        bool capped = true;
        for (int iter = 0; iter < cap; ++iter) {
            // ---- passive list (wave 0: ballot compaction in ascending order) ----
            __syncthreads();
            if (tid < 64) {
                int base = 0;
                for (int i0 = 0; i0 < k; i0 += 64) {
                    const int i = i0 + lane;
                    const bool on = i < k && inF[i] && !dead[i];
                    const unsigned long long m = __ballot(on);
                    if (on) plist[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
                    base += __popcll(m);
                }
                if (lane == 0) sh_p = base;
            }
            __syncthreads();
            p = sh_p;
            // ---- gather [G_PP + diag_add I; r_P^T] (lower triangle and the extra row) ----
            for (int64_t e = tid; e < (int64_t)(p + 1) * p; e += 256) {
                const int i = (int)(e / p), l = (int)(e % p);
                double v;
                if (i == p) v = (double)rs[plist[l]];
                else if (l > i) continue;
                else v = (double)G[(int64_t)plist[i] * kp + plist[l]] + (i == l ? diag_add : 0.0);
                M[e] = v;
            }
            __syncthreads();
            // ---- right-looking Cholesky; row p carries the forward substitution ----
            for (int j = 0; j < p; ++j) {
                const double d = M[(int64_t)j * p + j];
                const int vj = plist[j];
                const double gjj = (double)G[(int64_t)vj * kp + vj] + diag_add;
                if (!(d > GX_NNLS_PIVOT_EPS * gjj)) {              // (block-uniform) dropped pivot: the variable keeps x = 0 for the rest of this solve
                    for (int i = j + 1 + tid; i <= p; i += 256) M[(int64_t)i * p + j] = 0.0;
                    if (tid == 0) { ldiag[j] = 0.0; dead[vj] = 1; atomicAdd(&st->nnls_evicted, 1); }
                    __syncthreads();
                    continue;
                }
                const double sq = sqrt(d), inv = 1.0 / sq;
                for (int i = j + 1 + tid; i <= p; i += 256) M[(int64_t)i * p + j] *= inv;
                if (tid == 0) ldiag[j] = sq;
                __syncthreads();
                const int nr = p - j, nc = p - 1 - j;              // rows j + 1 .. p, columns j + 1 .. p - 1
                for (int64_t e = tid; e < (int64_t)nr * nc; e += 256) {
                    const int i = j + 1 + (int)(e / nc), l = j + 1 + (int)(e % nc);
                    if (l <= i) M[(int64_t)i * p + l] -= M[(int64_t)i * p + j] * M[(int64_t)l * p + j];
                }
                __syncthreads();
            }
            // ---- back substitution L^T x = z ----
            for (int l = tid; l < p; l += 256) zs[l] = M[(int64_t)p * p + l];
            __syncthreads();
            for (int j = p - 1; j >= 0; --j) {
                const double dj = ldiag[j];
                const double xj = dj > 0.0 ? zs[j] / dj : 0.0;
                for (int l = tid; l < j; l += 256) zs[l] -= M[(int64_t)j * p + l] * xj;
                if (tid == 0) xs[j] = xj;
                __syncthreads();
            }
            // ---- infeasible variables: x_i < 0 in the passive set, y_i = (G x - r)_i < 0 outside it ----
            double ax = 0.0;
            for (int l = tid; l < p; l += 256) ax = fmax(ax, fabs(xs[l]));
            const double tolx = GX_NNLS_TOL * gx_block_max(ax, redd, tid);
            for (int i = tid; i < k; i += 256) badf[i] = 0;
            __syncthreads();
            int ninf = 0, hi = -1;
            for (int l = tid; l < p; l += 256) {
                const int v = plist[l];
                if (!dead[v] && xs[l] < -tolx) { badf[v] = 1; ++ninf; hi = max(hi, v); }
            }
            for (int i = tid; i < k; i += 256) {
                if (inF[i] || dead[i]) continue;
                double y = -(double)rs[i];
                const float* grow = G + (int64_t)i * kp;
                for (int l = 0; l < p; ++l) y += (double)grow[plist[l]] * xs[l];
                if (y < -toly) { badf[i] = 1; ++ninf; hi = max(hi, i); }
            }
            const int total = gx_block_sum_i(ninf, redi, tid);
            if (total == 0) { capped = false; break; }
            bool full = true;
            if (total < best) { best = total; spare = 3; }
            else if (spare > 0) --spare;
            else full = false;
            const int top = gx_block_max_i(hi, redi, tid);        // (also orders the badf writes before the reads below)
            for (int i = tid; i < k; i += 256)
                if (badf[i] && (full || i == top)) inF[i] ^= 1;
        }
        __syncthreads();
        // ---- write the solution (the passive list of the last solve; everything else is zero) ----
        for (int i = tid; i < k; i += 256) X[(int64_t)i * sj + c * sc] = 0.f;
        __syncthreads();
        for (int l = tid; l < p; l += 256) {
            const double xv = xs[l];
            X[(int64_t)plist[l] * sj + c * sc] = xv > 0.0 ? (float)xv : 0.f;
        }
        if (capped && tid == 0) atomicAdd(&st->nnls_capped, 1);
    }
}

int gx_nnls(nmfx_engine* E, const float* G, double diag_add, const float* R, float* X, int64_t sj, int64_t sc, int64_t nprob) {
    ProfScope ps(E, "nnls");
    const int64_t kp = E->kp, per = (kp + 1) * kp;
    int64_t nb = std::min<int64_t>(std::min<int64_t>(4 * (int64_t)E->ncu, nprob), std::max<int64_t>(1, ((int64_t)1 << 30) / (per * 8)));
    nb = std::max<int64_t>(nb, 1);
    if (E->gx_nnls_cap < nb * per) {
        if (E->gx_nnls_work) { NMFX_HIP(hipStreamSynchronize(E->stream)); hipFree(E->gx_nnls_work); E->gx_nnls_work = nullptr; }
        NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->gx_nnls_work), (size_t)(nb * per) * sizeof(double)));
        E->gx_nnls_cap = nb * per;
    }
    const size_t shm = (size_t)kp * (3 * sizeof(double) + sizeof(float) + sizeof(int) + 3);
    int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(gx_nnls_kernel), (int)shm); if (rc) return rc;
    hipLaunchKernelGGL(gx_nnls_kernel, dim3((unsigned)nb), dim3(256), shm, E->stream, G, (int)kp, diag_add, R, X, sj, sc, nprob, E->k,
                       E->gx_nnls_work, E->state);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

}  // namespace

// ANLS outer iterations (anls.py:111-126): rows of W from G = H H^T + 2 lw I, r = rows of V H^T; columns of H from G = W^T W + 2 lh I,
// r = columns of W^T V; the objective (Euclidean or KL, E->anls_dist) of the new pair
int nmfx_generic_anls_run(nmfx_engine* E, double lam_w, double lam_h, int64_t min_iter, double tol1, double tol2, int64_t first, int64_t count) {
    E->gxb_img_ready = false;                          // (this solver rewrites W and H without the images of the k > 128 MUR loop)
    int rc;
    if ((rc = gx_buffers(E, false))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    const bool kl = E->anls_dist == NMFX_KL;
    float* W = E->W[0];
    float* xB = E->xf32;
    float* xG = E->xf32 + kp * np;
    // r4, split-bf16 runs: the two V-sized products from the V planes and the factor images with FOUR terms (like the k <= 128 ANLS:
    // the right-hand sides of the NNLS systems), the Euclidean objective by the persistent residual kernel; the Gram matrices,
    // whose conditioning decides the active sets, and the KL objective stay exact f32
    bool bf = gxb_on(E) && gx_anls_bf16();
    if (bf) { rc = gxb_prepare(E, W); if (rc == GXB_NOFIT) bf = false; else if (rc) return rc; }
    E->gxb_img_ready = false;                          // (valid inside this call only)
    if (first == 0 && count > 0 && (rc = kl ? gx_kl_objective_partial(E) : gx_objective_partial(E, bf))) return rc;     // obj[0] (anls.py:108)
    for (int64_t j = first; j < first + count; ++j) {
        hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                           E->state, E->obj_hist);
        NMFX_HIP(hipGetLastError());
        { ProfScope ps(E, "gram_nt");
          if ((rc = gx_split_product<true, true>(E, E->H, np, E->H, np, E->HHt, kp, kp, np, 64))) return rc; }
        { ProfScope ps(E, "wphase");
          if (bf) rc = gxt_split_product(E, E->gxb_v[0], E->gxb_v[1], E->Hhi, E->Hlo, E->A_part, mp, kp, np, 4, 4);
          else rc = gx_split_product<true, true>(E, E->V, np, E->H, np, E->A_part, mp, kp, np, 1);
          if (rc) return rc; }
        if ((rc = gx_nnls(E, E->HHt, 2.0 * lam_w, E->A_part, W, 1, kp, E->m))) return rc;
        if (bf) { ProfScope ps(E, "images");
                  if ((rc = gxb_images_w(E, W))) return rc; }
        { ProfScope ps(E, "gram_tn");
          if ((rc = gx_split_product<false, false>(E, W, kp, W, kp, xG, kp, kp, mp, 64))) return rc; }
        { ProfScope ps(E, "hphase");
          if (bf) rc = gxt_split_product(E, E->WThi, E->WTlo, E->gxb_v[2], E->gxb_v[3], xB, kp, np, mp, 8, 4);
          else rc = gx_split_product<false, false>(E, W, kp, E->V, np, xB, kp, np, mp, 8);
          if (rc) return rc; }
        if ((rc = gx_nnls(E, xG, 2.0 * lam_h, xB, E->H, np, 1, E->n))) return rc;
        if (bf) { ProfScope ps(E, "images");
                  if ((rc = gxb_images_h(E, E->H))) return rc; }
        if ((rc = kl ? gx_kl_objective_partial(E) : gx_objective_partial(E, bf))) return rc;
    }
    return NMFX_OK;
}

// row-sharded ANLS beyond 128 components (nmfx_anls_phase_* dispatch here, r4): 0 objective partial of the current pair -> xf64[0];
// 1 obj[j] + stop rule, H H^T (replicated), V H^T and the NNLS rows of W (rank-local), then [W^T V | W^T W] of the rank's rows into
// the f32 exchange buffer; 2 the NNLS columns of H from the all-reduced sums (replicated)
int nmfx_generic_anls_phase(nmfx_engine* E, int phase, double lam, int64_t min_iter, double tol1, double tol2, int64_t j) {
    E->gxb_img_ready = false;
    int rc;
    if ((rc = gx_buffers(E, false))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    const bool kl = E->anls_dist == NMFX_KL;
    float* W = E->W[0];
    float* xB = E->xf32;
    float* xG = E->xf32 + kp * np;
    if (phase == 0) return kl ? gx_kl_objective_partial(E) : gx_objective_partial(E);
    if (phase == 1) {
        hipLaunchKernelGGL(gx_record_kernel, dim3(1), dim3(1), 0, E->stream, (const double*)E->xf64, (long long)j, (long long)min_iter, tol1, tol2,
                           E->state, E->obj_hist);
        NMFX_HIP(hipGetLastError());
        { ProfScope ps(E, "gram_nt");
          if ((rc = gx_split_product<true, true>(E, E->H, np, E->H, np, E->HHt, kp, kp, np, 64))) return rc; }
        { ProfScope ps(E, "wphase");
          if ((rc = gx_split_product<true, true>(E, E->V, np, E->H, np, E->A_part, mp, kp, np, 1))) return rc; }
        if ((rc = gx_nnls(E, E->HHt, 2.0 * lam, E->A_part, W, 1, kp, E->m))) return rc;
        { ProfScope ps(E, "gram_tn");
          if ((rc = gx_split_product<false, false>(E, W, kp, W, kp, xG, kp, kp, mp, 64))) return rc; }
        { ProfScope ps(E, "hphase");
          return gx_split_product<false, false>(E, W, kp, E->V, np, xB, kp, np, mp, 8); }
    }
    return gx_nnls(E, xG, 2.0 * lam, xB, E->H, np, 1, E->n);
}

// The Euclidean objective of the CURRENT pair (W as nmfx_get_factors would return it, H) evaluated entirely in float64 on the
// device: the referee of the stop rule near the stop (see gx_resid64_kernel).  Synchronises.  Any k.
extern "C" int nmfx_objective_f64(nmfx_handle_t E, double* out) {
    if (!E || !out) { if (E) E->err = "objective_f64: out is NULL"; return NMFX_E_ARG; }
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if ((rc = nmfx_need_v(E))) return rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp, nblk = (mp / GX_T) * (np / GX_T);
    if ((rc = gx_alloc(E, &E->gx_part, nblk + 64))) return rc;
    const float* W = E->W[E->wsel];
    hipLaunchKernelGGL(gx_resid64_kernel, dim3((unsigned)(np / GX_T), (unsigned)(mp / GX_T)), dim3(256), 0, E->stream, W, kp, (const float*)E->H, np, kp,
                       (const float*)E->V, np, E->gx_part);
    hipLaunchKernelGGL(gx_sum64_kernel, dim3(1), dim3(256), 0, E->stream, (const double*)E->gx_part, nblk, E->gx_part + nblk);
    NMFX_HIP(hipGetLastError());
    NMFX_HIP(hipMemcpyAsync(out, E->gx_part + nblk, sizeof(double), hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

int nmfx_preload_generic() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(gx_record_kernel)) == hipSuccess ? 0 : -1; }
