// Small (k-sized) kernels of the MUR iteration: multiplicative update epilogues,
// deterministic reductions of split partials, the device-side convergence test.
//   reference: nmf/mur.py:20-49 (update rules), nmf/utils.py:4-15 (stop rule).
#include "nmfx_internal.h"
#include "kernels_small.h"

// out[i] = sum_s part[s][i], fixed order (bit-stable).
__global__ __launch_bounds__(256) void sum_partials_kernel(
    const float* __restrict__ part, int splits, int64_t count, float* __restrict__ out,
    const int* __restrict__ flag)
{
    if (*flag) return;
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= count) return;
    float4 s = *reinterpret_cast<const float4*>(part + i4 * 4);
    int p = 1;
    for (; p + 8 <= splits; p += 8) {                  // eight slabs in flight together (a Gram sum is 16 blocks: pure latency), added in slab order
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(part + (int64_t)(p + u) * count + i4 * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
    }
    for (; p < splits; ++p) {
        const float4 t = *reinterpret_cast<const float4*>(part + (int64_t)p * count + i4 * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4*>(out + i4 * 4) = s;
}

// Block-wide deterministic sum of `count` doubles (tree over a fixed layout).
__device__ double block_sum_f64(const double* __restrict__ src, int64_t count, double* sh) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) s += src[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;   // valid in thread 0
}

// pack: xf32 = [ sum_sr B_part | sum_s G_part ], xf64[0] = sum obj_part.
// Blocks [0, nb) do B, the next ceil(gcount/256) do G, the last one the objective.
__global__ __launch_bounds__(256) void mur_pack_kernel(
    const float* __restrict__ Bpart, int hsplit, int64_t bcount,
    const float* __restrict__ Gpart, int gsplit, int64_t gcount,
    const double* __restrict__ objpart, int64_t nobj,
    float* __restrict__ xf32, double* __restrict__ xf64, int nb, const int* __restrict__ flag,
    const int* __restrict__ flag2, float* __restrict__ xtail = nullptr, int xrank = 0, int xworld = 0)
{
    if (*flag || (flag2 && *flag2)) return;
    __shared__ double sh[4];
    const int b = blockIdx.x;
    if (b < nb) {
        // four grid strides per trip, every slab load of the four in flight together (k = 128 at n = 16384: eight strides per
        // thread, i.e. eight dependent round trips otherwise); slab order per element as before
        const int64_t stride = (int64_t)nb * 256, n4 = bcount / 4;
        int64_t i4 = (int64_t)b * 256 + threadIdx.x;
        for (; i4 + 3 * stride < n4; i4 += 4 * stride) {
            float4 s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = *reinterpret_cast<const float4*>(Bpart + (i4 + u * stride) * 4);
            for (int p = 1; p < hsplit; ++p) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(Bpart + (int64_t)p * bcount + (i4 + u * stride) * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) { s[u].x += t[u].x; s[u].y += t[u].y; s[u].z += t[u].z; s[u].w += t[u].w; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(xf32 + (i4 + u * stride) * 4) = s[u];
        }
        for (; i4 < n4; i4 += stride) {
            float4 s = *reinterpret_cast<const float4*>(Bpart + i4 * 4);
            for (int p = 1; p < hsplit; ++p) {
                const float4 t = *reinterpret_cast<const float4*>(Bpart + (int64_t)p * bcount + i4 * 4);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            *reinterpret_cast<float4*>(xf32 + i4 * 4) = s;
        }
    } else if (b < nb + (int)((gcount + 255) / 256)) {
        const int64_t i = (int64_t)(b - nb) * 256 + threadIdx.x;
        if (i < gcount) {
            float s = Gpart[i];
            int p = 1;
            for (; p + 8 <= gsplit; p += 8) {           // eight slabs in flight together, added in slab order
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = Gpart[(int64_t)(p + u) * gcount + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            for (; p < gsplit; ++p) s += Gpart[(int64_t)p * gcount + i];
            xf32[bcount + i] = s;
        }
    } else {
        const double t = block_sum_f64(objpart, nobj, sh);
        if (threadIdx.x == 0) xf64[0] = t;
        if (xtail) {                                   // NMFX_XTAIL: own slot = the four 16-bit digits of t, zeros elsewhere
            __shared__ unsigned long long tbits;
            if (threadIdx.x == 0) tbits = (unsigned long long)__double_as_longlong(t);
            __syncthreads();
            for (int i = threadIdx.x; i < 4 * xworld; i += blockDim.x)
                xtail[i] = (i >> 2) == xrank ? (float)((tbits >> (16 * (i & 3))) & 0xffffull) : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void obj_reduce_kernel(
    const double* __restrict__ objpart, int64_t nobj, double* __restrict__ xf64,
    const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ double sh[4];
    const double t = block_sum_f64(objpart, nobj, sh);
    if (threadIdx.x == 0) xf64[0] = t;
}

// W_new = W * A / (W HHt + lam W + 1e-9)        (nmf/mur.py:29, reassociated:
// (W H) H^T == W (H H^T)).  Block = 16 rows; thread (row = t/16, j = t%16 + 16 jj).
template <int KP>
__global__ __launch_bounds__(256) void mur_w_update_kernel(
    const float* __restrict__ Apart, int wsplit, int64_t mp, const float* __restrict__ Wold,
    const float* __restrict__ HHt, float lam, float* __restrict__ Wnew, const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ float hs[KP * KP];
    __shared__ float ws[16 * KP];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    for (int i = tid; i < KP * KP; i += 256) hs[i] = HHt[i];
    for (int i = tid; i < 16 * KP; i += 256) ws[i] = Wold[r0 * KP + i];
    __syncthreads();
    const int row = tid >> 4, jl = tid & 15;
#pragma unroll
    for (int jj = 0; jj < KP / 16; ++jj) {
        const int j = jl + 16 * jj;
        float d = 0.f;
#pragma unroll 8
        for (int l = 0; l < KP; ++l) d = fmaf(ws[row * KP + l], hs[l * KP + j], d);
        const int64_t idx = (r0 + row) * KP + j;
        float a = Apart[idx];
        for (int p = 1; p < wsplit; ++p) a += Apart[(int64_t)p * mp * KP + idx];
        const float w = ws[row * KP + j];
        Wnew[idx] = w * a / (d + lam * w + 1e-9f);
    }
}

// H_new = H * B / (G H + lam H + 1e-9)           (nmf/mur.py:45, reassociated:
// W^T (W H) == (W^T W) H), preceded by the bookkeeping of the objective that
// phase A produced: obj[j] is recorded and the reference's convergence test for
// loop index i = j-1 is evaluated by every block identically; when it fires the
// update is skipped so (W_j, H_j) -- the reference's return value -- survive.
template <int KP>
__global__ __launch_bounds__(256) void mur_h_update_kernel(
    const float* __restrict__ xf32, const double* __restrict__ xf64, float* __restrict__ H,
    int64_t np, float lam, long long j, long long min_iter, double tol1, double tol2,
    DevState* __restrict__ st, double* __restrict__ obj_hist)
{
    if (st->flag) return;
    const int rule = nmfx_record_objective(st, obj_hist, xf64[0], j, min_iter, tol1, tol2,
                                           blockIdx.x == 0 && threadIdx.x == 0);
    if (rule) return;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* gs = lds;                 // [KP][KP]
    float* hs = lds + KP * KP;       // [KP][64]
    const int tid = threadIdx.x, c = tid & 63, jq = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const float* G = xf32 + (int64_t)KP * np;
    for (int i = tid; i < KP * KP; i += 256) gs[i] = G[i];
    for (int i = tid; i < KP * 64; i += 256) hs[i] = H[(int64_t)(i >> 6) * np + c0 + (i & 63)];
    __syncthreads();
    constexpr int NJ = KP / 4;       // outputs per thread: j in [jq*NJ, jq*NJ + NJ)
    float acc[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) acc[t] = 0.f;
    for (int l = 0; l < KP; ++l) {
        const float hv = hs[l * 64 + c];
        const float* grow = gs + l * KP + jq * NJ;        // G symmetric: G[j][l] == G[l][j]
#pragma unroll
        for (int t = 0; t < NJ; t += 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(grow + t);
            acc[t] = fmaf(g4.x, hv, acc[t]);
            acc[t + 1] = fmaf(g4.y, hv, acc[t + 1]);
            acc[t + 2] = fmaf(g4.z, hv, acc[t + 2]);
            acc[t + 3] = fmaf(g4.w, hv, acc[t + 3]);
        }
    }
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
        const int jrow = jq * NJ + t;
        const float h = hs[jrow * 64 + c];
        const float b = xf32[(int64_t)jrow * np + c0 + c];
        H[(int64_t)jrow * np + c0 + c] = h * b / (acc[t] + lam * h + 1e-9f);
    }
}

__global__ void finalize_kernel(const double* __restrict__ xf64, long long j, long long min_iter,
                                double tol1, double tol2, DevState* __restrict__ st,
                                double* __restrict__ obj_hist)
{
    if (st->flag) return;
    nmfx_record_objective(st, obj_hist, xf64[0], j, min_iter, tol1, tol2, threadIdx.x == 0);
}

// --------------------------------------------------------------------------
// host-side sequencing of one MUR-eu iteration
// --------------------------------------------------------------------------
int nmfx_launch_sum_partials(nmfx_engine* E, const float* part, int splits, int64_t count, float* out) {
    const int64_t n4 = (count + 3) / 4;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, E->stream,
                       part, splits, count, out, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int KP>
static int launch_w_update(nmfx_engine* E, const float* Wold, float* Wnew, float lam, int splits = 0) {
    hipLaunchKernelGGL((mur_w_update_kernel<KP>), dim3((unsigned)(E->mp / 16)), dim3(256), 0, E->stream,
                       E->A_part, splits > 0 ? splits : E->wsplit, E->mp, Wold, E->HHt, lam, Wnew, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int KP>
static int launch_h_update(nmfx_engine* E, float lam, int64_t j, int64_t min_iter, double tol1, double tol2) {
    const size_t shm = (size_t)(KP * KP + KP * 64) * sizeof(float);
    auto kern = mur_h_update_kernel<KP>;
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(E->np / 64)), dim3(256), shm, E->stream, E->xf32, E->xf64,
                       E->H, E->np, lam, (long long)j, (long long)min_iter, tol1, tol2, E->state,
                       E->obj_hist);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_launch_pack(nmfx_engine* E, const int* flag2) {
    ProfScope ps(E, "pack");
    const int nb = 256;
    const int64_t nobj = (int64_t)(E->mp / 64) * E->wsplit;
    const int ngb = (int)(((int64_t)E->kp * E->kp + 255) / 256);
    hipLaunchKernelGGL(mur_pack_kernel, dim3(nb + ngb + 1), dim3(256), 0, E->stream, E->B_part, E->hsplit,
                       (int64_t)E->kp * E->np, E->G_part, nmfx_g_slabs(E), (int64_t)E->kp * E->kp,
                       E->obj_part, nobj, E->xf32, E->xf64, nb, &E->state->flag, flag2);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_launch_pack_from(nmfx_engine* E, const float* Bpart, int bsplit, const float* Gpart, int gsplit,
                          int64_t nobj) {
    ProfScope ps(E, "pack");
    const int nb = 256;
    const int ngb = (int)(((int64_t)E->kp * E->kp + 255) / 256);
    float* tail = E->xworld > 0 ? E->xf32 + (int64_t)E->kp * E->np + (int64_t)E->kp * E->kp + E->kp : nullptr;
    hipLaunchKernelGGL(mur_pack_kernel, dim3(nb + ngb + 1), dim3(256), 0, E->stream, Bpart, bsplit,
                       (int64_t)E->kp * E->np, Gpart, gsplit, (int64_t)E->kp * E->kp, E->obj_part, nobj,
                       E->xf32, E->xf64, nb, &E->state->flag, (const int*)nullptr, tail, E->xrank, E->xworld);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_launch_obj_reduce(nmfx_engine* E, int64_t nobj, const double* src) {
    ProfScope ps(E, "small");
    if (nobj <= 0) nobj = E->obj_count > 0 ? E->obj_count : (int64_t)(E->mp / 64) * E->wsplit;
    hipLaunchKernelGGL(obj_reduce_kernel, dim3(1), dim3(256), 0, E->stream, src ? src : (const double*)E->obj_part, nobj, E->xf64,
                       &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// W / H epilogues of the exact-f32 path, callable from the split-bf16 path for k = 128
int nmfx_launch_w_update(nmfx_engine* E, const float* Wold, float* Wnew, float lam, int splits) {
    ProfScope ps(E, "w_update");
    switch (E->kp) {
        case 16: return launch_w_update<16>(E, Wold, Wnew, lam, splits);
        case 32: return launch_w_update<32>(E, Wold, Wnew, lam, splits);
        case 64: return launch_w_update<64>(E, Wold, Wnew, lam, splits);
        default: return launch_w_update<128>(E, Wold, Wnew, lam, splits);
    }
}

int nmfx_launch_h_update(nmfx_engine* E, float lam, int64_t j, int64_t min_iter, double tol1, double tol2) {
    ProfScope ps(E, "h_update");
    switch (E->kp) {
        case 16: return launch_h_update<16>(E, lam, j, min_iter, tol1, tol2);
        case 32: return launch_h_update<32>(E, lam, j, min_iter, tol1, tol2);
        case 64: return launch_h_update<64>(E, lam, j, min_iter, tol1, tol2);
        default: return launch_h_update<128>(E, lam, j, min_iter, tol1, tol2);
    }
}

int nmfx_mur_eu_phase_a(nmfx_engine* E, double lambda_w, int64_t j) {
    const float* Wold = E->W[j & 1];
    float* Wnew = E->W[(j + 1) & 1];
    int rc;
    // HHt of the current H (partials were produced by the previous phase B / set_factors)
    { ProfScope ps(E, "sum_hht");
      if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, (int64_t)E->kp * E->kp, E->HHt))) return rc; }
    if ((rc = nmfx_launch_wphase(E, Wold, true, true))) return rc;
    { ProfScope ps(E, "w_update");
      switch (E->kp) {
        case 16: rc = launch_w_update<16>(E, Wold, Wnew, (float)lambda_w); break;
        case 32: rc = launch_w_update<32>(E, Wold, Wnew, (float)lambda_w); break;
        case 64: rc = launch_w_update<64>(E, Wold, Wnew, (float)lambda_w); break;
        default: rc = launch_w_update<128>(E, Wold, Wnew, (float)lambda_w); break;
      }
      if (rc) return rc; }
    const bool fuse_g = nmfx_hphase_can_fuse_gram(E);
    if (!fuse_g && (rc = nmfx_launch_gram_tn(E, Wnew, E->mp, E->G_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_hphase(E, Wnew, fuse_g))) return rc;
    if ((rc = nmfx_launch_pack(E))) return rc;
    return NMFX_OK;
}

int nmfx_mur_eu_phase_b(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    { ProfScope ps(E, "h_update");
      switch (E->kp) {
        case 16: rc = launch_h_update<16>(E, (float)lambda_h, j, min_iter, tol1, tol2); break;
        case 32: rc = launch_h_update<32>(E, (float)lambda_h, j, min_iter, tol1, tol2); break;
        case 64: rc = launch_h_update<64>(E, (float)lambda_h, j, min_iter, tol1, tol2); break;
        default: rc = launch_h_update<128>(E, (float)lambda_h, j, min_iter, tol1, tol2); break;
      }
      if (rc) return rc; }
    return nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit);
}

int nmfx_mur_eu_finish_a(nmfx_engine* E, int64_t j) {
    int rc;
    if ((rc = nmfx_launch_wphase(E, E->W[j & 1], false, true))) return rc;
    return nmfx_launch_obj_reduce(E);
}

int nmfx_finish_b(nmfx_engine* E, int64_t min_iter, double tol1, double tol2, int64_t j) {
    ProfScope ps(E, "small");
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, E->stream, E->xf64, (long long)j,
                       (long long)min_iter, tol1, tol2, E->state, E->obj_hist);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_mur() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(obj_reduce_kernel)) == hipSuccess ? 0 : -1; }
