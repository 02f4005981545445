// MUR with the Kullback-Leibler divergence.
//   reference: w_update / h_update 'kl' branches nmf/mur.py:24-27, 40-43;
//   objective nmf/utils.py:21-26.
//
//   W <- 2a / (b + sqrt(b^2 + 4 lam a)),  a = W o ((V / (W H + 1e-9)) H^T),  b = 1 H^T
//   H <- 2c / (d + sqrt(d^2 + 4 lam c)),  c = H o (W^T (V / (W H + 1e-9))),  d = W^T 1
//
// The reference spends two full m*n*k GEMMs on the constants b and d
// (`np.ones_like(x) @ h.T`, `w.T @ np.ones_like(x)`); they are the row sums of
// H and the column sums of W, computed here by reductions.  The quotient
// V / (W H + 1e-9) is never written to memory: each kernel forms the W H tile
// with MFMA in the register layout its V slice already has, divides in
// registers and feeds the result straight into the second MFMA product.
#include "nmfx_internal.h"
#include "kernels_small.h"

static int launch_h_row_sums(nmfx_engine* E);      // (defined next to the split-bf16 path below)

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// --------------------------------------------------------------------------
// H phase: B = W^T (V / (W H + 1e-9)) over the rows of split sr, NE*16 columns
// per block.  Stage = 16 rows:
//   P tile e  = W(16 x KP) . Hpanel(KP x 16)      -> lane (x,q) reg r = P[4q+r][16e+x]
//   Q         = V / (P + 1e-9)  (V loaded in that same layout)
//   acc[jt][e] += W(4 rows, tile jt)^T . Q(4 rows, tile e)   for the 4 row groups r
// Blocks with blockIdx.x == 0 also accumulate the column sums of W (the d term).
// --------------------------------------------------------------------------
template <int KP, int NE>
__global__ __launch_bounds__(256) void hphase_kl_kernel(
    const float* __restrict__ V, int64_t ldv, const float* __restrict__ W,
    const float* __restrict__ H, float* __restrict__ Bpart, float* __restrict__ csum_part,
    int64_t np, int64_t mp, const int* __restrict__ flag)
{
    if (*flag) return;
    constexpr int JT = KP / 16;
    constexpr int LDH = 16 * NE + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int SR = gridDim.y, sr = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * 16 * NE;
    const int64_t n16 = mp / 16;
    const int64_t u0 = n16 * sr / SR, u1 = n16 * (sr + 1) / SR;
    const int64_t t0 = u0 + (u1 - u0) * wave / 4, t1 = u0 + (u1 - u0) * (wave + 1) / 4;

    for (int i = tid; i < KP * 16 * NE; i += 256) {             // H panel [KP][16 NE]
        const int r = i / (16 * NE), c = i % (16 * NE);
        lds[r * LDH + c] = H[(int64_t)r * np + c0 + c];
    }
    __syncthreads();

    f32x4 acc[JT][NE];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[j][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) cs[j] = 0.f;
    const bool do_cs = (blockIdx.x == 0);

    for (int64_t t = t0; t < t1; ++t) {
        const int64_t r0 = t * 16;
        float4 wf[JT];
#pragma unroll
        for (int u = 0; u < JT; ++u)
            wf[u] = *reinterpret_cast<const float4*>(W + (r0 + x) * KP + 16 * u + 4 * q);
        float vv[NE][4], wa[4][JT];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int e = 0; e < NE; ++e) vv[e][r] = V[(r0 + 4 * q + r) * ldv + c0 + 16 * e + x];
#pragma unroll
            for (int j = 0; j < JT; ++j) wa[r][j] = W[(r0 + 4 * q + r) * KP + 16 * j + x];
        }
        f32x4 pe[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) pe[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < JT; ++u) {
            const float wv[4] = {wf[u].x, wf[u].y, wf[u].z, wf[u].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* hrow = lds + (16 * u + 4 * q + s) * LDH + x;
#pragma unroll
                for (int e = 0; e < NE; ++e) pe[e] = MFMA(wv[s], hrow[16 * e], pe[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[e][r] = vv[e][r] / (pe[e][r] + 1e-9f);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < JT; ++j) {
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[j][e] = MFMA(wa[r][j], vv[e][r], acc[j][e]);
                if (do_cs) cs[j] += wa[r][j];
            }
    }
    __syncthreads();                                            // H panel no longer needed

    // fixed-order cross-wave sum, then store (each lane owns its LDS words)
    f32x4* red = reinterpret_cast<f32x4*>(lds);                 // [JT*NE][64]
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int slot = (j * NE + e) * 64 + lane;
                    if (w == 0) red[slot] = acc[j][e];
                    else if (w < 3) red[slot] += acc[j][e];
                    else {
                        const f32x4 tt = red[slot] + acc[j][e];
                        float* out = Bpart + (int64_t)sr * KP * np + c0 + 16 * e + x;
#pragma unroll
                        for (int g = 0; g < 4; ++g) out[(int64_t)(16 * j + 4 * q + g) * np] = tt[g];
                    }
                }
        }
        __syncthreads();
    }
    if (do_cs) {
        // column sums of W over this split's rows: lanes (x, q=0..3) hold the four row
        // groups of factor 16j+x; sum over q, then over waves in order
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            float v = cs[j];
            v += __shfl_down(v, 32, 64);
            v += __shfl_down(v, 16, 64);
            cs[j] = v;
        }
        float* cred = lds;                                      // [4][KP]
        if (q == 0)
#pragma unroll
            for (int j = 0; j < JT; ++j) cred[wave * KP + 16 * j + x] = cs[j];
        __syncthreads();
        if (tid < KP)
            csum_part[(int64_t)sr * KP + tid] = ((cred[tid] + cred[KP + tid]) + cred[2 * KP + tid]) + cred[3 * KP + tid];
    }
}

// out[j] = sum_c X[j][c]   (one block per row, fixed-order tree)
__global__ __launch_bounds__(256) void row_sums_kernel(const float* __restrict__ X, int64_t cols, int64_t ld,
                                                       float* __restrict__ out, const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ float sh[4];
    const float* p = X + (int64_t)blockIdx.x * ld;
    float s = 0.f;
    for (int64_t c = threadIdx.x * 4; c < cols; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// W_new = 2a / (b + sqrt(b^2 + 4 lam a)), a = W * sum_sp A_part, b = rowsum_H[j]   (mur.py:25-27)
__global__ __launch_bounds__(256) void kl_w_update_kernel(
    const float* __restrict__ Apart, int wsplit, int64_t count, int kp, int k, const float* __restrict__ Wold,
    const float* __restrict__ rowsum, float lam, float* __restrict__ Wnew, const int* __restrict__ flag)
{
    if (*flag) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    if ((int)(i % kp) >= k) { Wnew[i] = 0.f; return; }      // padded factors: 0/0 must stay out
    float s = Apart[i];
    for (int p = 1; p < wsplit; ++p) s += Apart[(int64_t)p * count + i];
    const float a = Wold[i] * s;
    const float b = rowsum[i % kp];
    Wnew[i] = 2.f * a / (b + sqrtf(b * b + 4.f * lam * a));
}

// H_new = 2c / (d + sqrt(d^2 + 4 lam c)), c = H * B, d = colsum_W[j]   (mur.py:41-43),
// after the objective bookkeeping / convergence test (same protocol as MUR-eu).
__global__ __launch_bounds__(256) void kl_h_update_kernel(
    const float* __restrict__ xf32, const double* __restrict__ xf64, float* __restrict__ H, int64_t np,
    int kp, int k, float lam, long long j, long long min_iter, double tol1, double tol2,
    DevState* __restrict__ st, double* __restrict__ obj_hist)
{
    if (st->flag) return;
    const int rule = nmfx_record_objective(st, obj_hist, xf64[0], j, min_iter, tol1, tol2,
                                           blockIdx.x == 0 && threadIdx.x == 0);
    if (rule) return;
    const int64_t count = (int64_t)kp * np;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;
    if (i / np >= k) return;                                 // padded factor rows stay zero
    const float d = xf32[count + i / np];
    const float4 h = *reinterpret_cast<const float4*>(H + i);
    const float4 b = *reinterpret_cast<const float4*>(xf32 + i);
    float4 o;
    float c;
    c = h.x * b.x; o.x = 2.f * c / (d + sqrtf(d * d + 4.f * lam * c));
    c = h.y * b.y; o.y = 2.f * c / (d + sqrtf(d * d + 4.f * lam * c));
    c = h.z * b.z; o.z = 2.f * c / (d + sqrtf(d * d + 4.f * lam * c));
    c = h.w * b.w; o.w = 2.f * c / (d + sqrtf(d * d + 4.f * lam * c));
    *reinterpret_cast<float4*>(H + i) = o;
}

// pack for KL: xf32 = [ sum B slabs | sum colsum slabs ], xf64[0] = sum obj_part
__global__ __launch_bounds__(256) void kl_pack_kernel(
    const float* __restrict__ Bpart, int hsplit, int64_t bcount, const float* __restrict__ cpart, int kp,
    const double* __restrict__ objpart, int64_t nobj, float* __restrict__ xf32, double* __restrict__ xf64,
    int nb, const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ double sh[4];
    const int b = blockIdx.x;
    if (b < nb) {
        for (int64_t i4 = (int64_t)b * 256 + threadIdx.x; i4 * 4 < bcount; i4 += (int64_t)nb * 256) {
            float4 s = *reinterpret_cast<const float4*>(Bpart + i4 * 4);
            for (int p = 1; p < hsplit; ++p) {
                const float4 t = *reinterpret_cast<const float4*>(Bpart + (int64_t)p * bcount + i4 * 4);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            *reinterpret_cast<float4*>(xf32 + i4 * 4) = s;
        }
    } else if (b == nb) {
        if ((int)threadIdx.x < kp) {
            float s = cpart[threadIdx.x];
            for (int p = 1; p < hsplit; ++p) s += cpart[(int64_t)p * kp + threadIdx.x];
            xf32[bcount + threadIdx.x] = s;
        }
    } else {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < nobj; i += 256) s += objpart[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) xf64[0] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    }
}

// --------------------------------------------------------------------------
template <int KP, int NE>
static int launch_hphase_kl(nmfx_engine* E, const float* W) {
    dim3 grid((unsigned)(E->np / (16 * NE)), (unsigned)E->hsplit), block(256);
    const size_t panel = (size_t)KP * (16 * NE + 4) * sizeof(float);
    const size_t red = (size_t)(KP / 16) * NE * 64 * sizeof(f32x4);
    const size_t shm = std::max(std::max(panel, red), (size_t)4 * KP * sizeof(float));
    { int rc_ = nmfx_need_v(E); if (rc_) return rc_; }
    hipLaunchKernelGGL((hphase_kl_kernel<KP, NE>), grid, block, shm, E->stream, E->V, E->np, W, E->H,
                       E->B_part, E->G_part, E->np, E->mp, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_mur_kl_phase_a(nmfx_engine* E, double lambda_w, int64_t j) {
    const float* Wold = E->W[j & 1];
    float* Wnew = E->W[(j + 1) & 1];
    int rc;
    if ((rc = launch_h_row_sums(E))) return rc;      // b = 1 H^T  (HHt is unused by KL; its first kp floats hold the sums)
    if ((rc = nmfx_launch_wphase(E, Wold, true, true, true))) return rc;
    { ProfScope ps(E, "w_update");
      const int64_t count = E->mp * E->kp;
      hipLaunchKernelGGL(kl_w_update_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, E->stream,
                         E->A_part, E->wsplit, count, E->kp, E->k, Wold, E->HHt, (float)lambda_w, Wnew,
                         &E->state->flag);
      NMFX_HIP(hipGetLastError()); }
    { ProfScope ps(E, "hphase");
      switch (E->kp) {
        case 16: rc = launch_hphase_kl<16, 4>(E, Wnew); break;
        case 32: rc = launch_hphase_kl<32, 4>(E, Wnew); break;
        case 64: rc = launch_hphase_kl<64, 4>(E, Wnew); break;
        default: rc = launch_hphase_kl<128, 2>(E, Wnew); break;
      }
      if (rc) return rc; }
    { ProfScope ps(E, "pack");
      const int nb = 256;
      const int64_t nobj = (int64_t)(E->mp / 64) * E->wsplit;
      hipLaunchKernelGGL(kl_pack_kernel, dim3(nb + 2), dim3(256), 0, E->stream, E->B_part, E->hsplit,
                         (int64_t)E->kp * E->np, E->G_part, E->kp, E->obj_part, nobj, E->xf32, E->xf64, nb,
                         &E->state->flag);
      NMFX_HIP(hipGetLastError()); }
    return NMFX_OK;
}

int nmfx_mur_kl_phase_b(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j) {
    ProfScope ps(E, "h_update");
    const int64_t n4 = ((int64_t)E->kp * E->np) / 4;
    hipLaunchKernelGGL(kl_h_update_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, E->stream, E->xf32,
                       E->xf64, E->H, E->np, E->kp, E->k, (float)lambda_h, (long long)j, (long long)min_iter, tol1,
                       tol2, E->state, E->obj_hist);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- split-bf16 products (kp = 64 / 128): the two quotient products of an iteration on the bf16
// MFMA (xyt_bf16_kernel<.., KL = true>), everything else as above ----
// part[blk][j] = sum over the block's 128 rows of W[r][j]   (coalesced along j)
__global__ __launch_bounds__(256) void col_sums_part_kernel(const float* __restrict__ W, int kp, float* __restrict__ part,
                                                            const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ float sh[256];
    const int j = threadIdx.x % kp, rl = threadIdx.x / kp, nrl = 256 / kp;
    const float* p = W + (int64_t)blockIdx.x * 128 * kp;
    float s = 0.f;
    for (int r = rl; r < 128; r += nrl) s += p[(int64_t)r * kp + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        for (int t = 1; t < nrl; ++t) s += sh[t * kp + j];
        part[(int64_t)blockIdx.x * kp + j] = s;
    }
}

// out[j] = sum_b part[b][j]: one wavefront per j (lane-strided partial sums, then the shuffle tree: fixed order).  The
// single 128-thread block that walked all partials serially took 30 us of the 41 us "row_sums" of config 4.
__global__ __launch_bounds__(256) void col_sums_final_kernel(const float* __restrict__ part, int nblk, int kp,
                                                             float* __restrict__ out, const int* __restrict__ flag)
{
    if (*flag) return;
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= kp) return;
    float s = 0.f;
    for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * kp + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[j] = s;
}

// part[chunk][j] = sum of 4096 columns of row j of X (row sums of H in two stages: kp blocks of one row each streamed
// 64 KiB through 256 threads -- 64 blocks, latency bound, 40 us at n = 16384)
__global__ __launch_bounds__(256) void row_sums_part_kernel(const float* __restrict__ X, int64_t cols, int64_t ld, int kp,
                                                            float* __restrict__ part, const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ float sh[4];
    const int64_t c0 = (int64_t)blockIdx.x * 4096;
    const float* p = X + (int64_t)blockIdx.y * ld;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t c = c0 + (threadIdx.x + 256 * u) * 4;
        if (c < cols) { const float4 v = *reinterpret_cast<const float4*>(p + c); s += (v.x + v.y) + (v.z + v.w); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * kp + blockIdx.y] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// HHt[0 .. kp) = row sums of H (b = 1 H^T, mur.py:26): two small launches; scratch: the head of B_part (dead between the
// pack of one iteration and the H phase of the next)
static int launch_h_row_sums(nmfx_engine* E) {
    ProfScope ps(E, "row_sums");
    const int nchunk = (int)((E->np + 4095) / 4096);
    hipLaunchKernelGGL(row_sums_part_kernel, dim3((unsigned)nchunk, (unsigned)E->kp), dim3(256), 0, E->stream, E->H, E->np, E->np,
                       E->kp, E->B_part, &E->state->flag);
    hipLaunchKernelGGL(col_sums_final_kernel, dim3((unsigned)((E->kp + 3) / 4)), dim3(256), 0, E->stream, E->B_part, nchunk, E->kp,
                       E->HHt, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_mur_kl_phase_a_bf16(nmfx_engine* E, double lambda_w, int64_t j) {
    int rc;
    if (!E->bf_ready) E->wsel = (int)(j & 1);
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    const int cur = (int)(j & 1), nxt = cur ^ 1;
    const float* Wold = E->W[cur];
    float* Wnew = E->W[nxt];
    // the images of H in both layouts and b = 1 H^T (HHt is unused by KL; its first kp floats hold the sums): left by the
    // H epilogue of iteration j - 1 (images as they are, the sums as partials: one small launch), or built from H itself
    // (first iteration, new factors, another solver in between)
    const bool fresh = E->kl_part && E->kl_h_iter == j - 1;
    if (!fresh && (rc = nmfx_bf16_images_h(E, true))) return rc;
    if (fresh) {
        ProfScope ps(E, "row_sums");
        hipLaunchKernelGGL(col_sums_final_kernel, dim3((unsigned)((E->kp + 3) / 4)), dim3(256), 0, E->stream, E->kl_part,
                           (int)(E->np / 64), E->kp, E->HHt, &E->state->flag);
        NMFX_HIP(hipGetLastError());
    } else if ((rc = launch_h_row_sums(E))) return rc;
    if ((rc = nmfx_bf16_vht(E, true, cur, "wphase", true, 3))) return rc;
    // W_new + its images + the column-sum partials in one launch (kl_w_epilogue_kernel)
    if ((rc = nmfx_bf16_kl_w_epilogue(E, Wold, Wnew, nxt, (float)lambda_w, E->HHt))) return rc;
    if ((rc = nmfx_bf16_vtw(E, false, "hphase", true, 3))) return rc;
    { ProfScope ps(E, "row_sums");        // d = W^T 1 into the first kp floats of G_part
      hipLaunchKernelGGL(col_sums_final_kernel, dim3((unsigned)((E->kp + 3) / 4)), dim3(256), 0, E->stream,
                         E->kl_part + (int64_t)(E->np / 64) * E->kp, (int)(E->mp / 64), E->kp, E->G_part, &E->state->flag);
      NMFX_HIP(hipGetLastError()); }
    return nmfx_bf16_pack_t(E, E->G_part, 1, E->obj_count);            // (the W phase's launch has set it: (mp / 128) * bf_wsplit, or mp / 64 row blocks with NMFX_KL_NW4)
}

int nmfx_mur_kl_phase_b_bf16(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j) {
    // H_new + Hhi/Hlo (next W phase) + H^T images (next H phase) + the row-sum partials of the next W update in one launch
    return nmfx_bf16_kl_h_epilogue(E, (float)lambda_h, j, min_iter, tol1, tol2);
}

int nmfx_mur_kl_finish_a(nmfx_engine* E, int64_t j) {
    int rc;
    if ((rc = nmfx_launch_wphase(E, E->W[j & 1], false, true, true))) return rc;
    return nmfx_launch_obj_reduce(E);
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_kl() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(row_sums_kernel)) == hipSuccess ? 0 : -1; }
