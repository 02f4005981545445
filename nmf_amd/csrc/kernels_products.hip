// V-sized contractions of the NMF hot path, hand-written for gfx950.
//
//   wphase : A = V H^T  (m x k, contraction over n)  [+ fused residual objective]
//            reference: `x @ h.T` nmf/mur.py:29, `w.T @ y` on transposes
//            nmf/ao_admm.py:56,265; objective nmf/utils.py:29.
//   hphase : B = W^T V  (k x n, contraction over m)
//            reference: `w.T @ x` nmf/mur.py:45, nmf/ao_admm.py:56.
//   gram   : W^T W, H H^T (k x k)   reference: nmf/ao_admm.py:53, and the
//            reassociated `wh @ h.T` = W (H H^T), `w.T @ wh` = (W^T W) H of
//            nmf/mur.py:29,45.
//
// All use v_mfma_f32_16x16x4_f32 (exact f32 FMA chains).  The operand maps are
//   A-operand lane l: A[i = l&15][kk = l>>4],  B-operand: B[kk = l>>4][j = l&15],
//   C/D: D[row = 4*(l>>4) + reg][col = l&15].
// Because a contraction may visit its index in any order, each lane feeds the
// four elements of ONE 16-byte load to four consecutive MFMA k-steps (wphase) or
// to four different output tiles (hphase); both operands use the same
// permutation, so every global/LDS access is a full dwordx4.
#include "nmfx_internal.h"
#include <mutex>
#include <utility>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Make a loaded fragment opaque so hipcc keeps it in registers instead of
// re-issuing the (restrict, read-only) load inside the main loop, where its
// s_waitcnt vmcnt(0) would drain the software-pipelined prefetch.
__device__ __forceinline__ void pin(float4& v) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// LDS-DMA: 16 bytes per lane from a per-lane global address to wave-uniform LDS base
// + lane*16 (global_load_lds_dwordx4), asynchronous, tracked by vmcnt.  Issued through
// inline asm on purpose: with the builtin, hipcc treats the DMA as a pending LDS write
// and puts `s_waitcnt vmcnt(0)` in front of the next ds_read of the (single) LDS array,
// which serialises prefetch and compute.  The asm form is invisible to that
// bookkeeping, so the caller owns the wait: dma_wait_all() before the barrier that
// publishes the tile (cdna_hip_programming.md 5.7: M0 is written in the same statement).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_dst_wave_base) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// XOR swizzle of the 16-byte slot inside a 64-float LDS row; conflict-free for
// both read patterns of wphase (derivation in DESIGN.md, "H tile image").
__device__ __forceinline__ int swz(int row) {
    const int r = row & 15;
    return r ^ ((((r >> 2) ^ (r >> 3)) & 1) << 2);
}

// --------------------------------------------------------------------------
// wphase: one block = 64 rows (4 waves x 16 rows) x column groups [g0, g1).
// Per 64-column group a wave holds its 16x64 slice of V in 16 VGPRs.  V is read
// from HBM exactly once, with coalesced loads (4 rows x 256 B per wave
// instruction, prefetched one group ahead) into a wave-private swizzled LDS tile
// from which the MFMA fragments are read (fragment-shaped global loads -- 16
// rows x 16 B per quarter wave -- ran at 2 TB/s; this form does not).  The H
// tile [KP][64] is shared through LDS (double buffered).
//   A-product : acc[jt] += V(16x64) . Htile^T         -> A[16][KP]
//   D-product : d[e]    = (W Htile) transposed tiles  -> residual V - W H
// The D tiles come out of the MFMA in exactly the register layout the V slice
// already has (row on lane&15, column 16q+4reg+e), so the residual needs no
// data movement.
// --------------------------------------------------------------------------
template <int KP, bool WITH_A, bool WITH_OBJ, bool KL>
__global__ __launch_bounds__(256) void wphase_kernel(
    const float* __restrict__ V, int64_t ldv, const float* __restrict__ W,
    const float* __restrict__ H, int64_t ldh, float* __restrict__ Apart,
    double* __restrict__ objpart, int64_t mp, int ngroups, const int* __restrict__ flag,
    const int* __restrict__ flag2)
{
    if (*flag || (flag2 && *flag2)) return;
    constexpr int JT = KP / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x [KP][64] H tiles + 4 x [16][64] V tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, q = lane >> 4;
    const int S = gridDim.y, sp = blockIdx.y;
    const int g0 = (int)((int64_t)ngroups * sp / S);
    const int g1 = (int)((int64_t)ngroups * (sp + 1) / S);
    const int64_t r0 = (int64_t)blockIdx.x * 64 + wave * 16;

    float4 wf[JT];
#pragma unroll
    for (int u = 0; u < JT; ++u)
        wf[u] = *reinterpret_cast<const float4*>(W + (r0 + x) * KP + 16 * u + 4 * q);
#pragma unroll
    for (int u = 0; u < JT; ++u) pin(wf[u]);

    f32x4 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double osum = 0.0;

    // Staging by LDS-DMA.  One wave instruction fills 1 KiB of LDS linearly (4 tile rows
    // of 256 B); the XOR swizzle is applied to the per-lane SOURCE address: lane
    // (row q of the 4, position x) fetches chunk x ^ swz(row), so position p of a row
    // holds chunk p ^ swz(row) -- the image the swizzled fragment reads expect.
    //   H tile [KP][64]: wave w fills rows 4*(JT*w + p) .. +3, p < JT
    //   V tile [16][64] (wave-private): rows 4t .. 4t+3, t < 4
    float* vt = lds + 2 * KP * 64 + wave * (16 * 64);
    const float* hsrc[JT];
    const float* vsrc[4];
#pragma unroll
    for (int p = 0; p < JT; ++p) {
        const int row = 4 * (JT * wave + p) + q;
        hsrc[p] = H + (int64_t)row * ldh + 4 * (x ^ swz(row));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = 4 * t + q;
        vsrc[t] = V + (r0 + row) * ldv + 4 * (x ^ swz(row));
    }
    // read maps
    const int gx = swz(x);                 // rows with (row & 15) == x: H tile A-product, V tile
    int dslot[4];                          // D-product: rows 16u + 4q + s
#pragma unroll
    for (int s = 0; s < 4; ++s) dslot[s] = (4 * q + s) * 64 + 4 * (x ^ swz(4 * q + s));

    float4 vf[4];
    if (g0 < g1) {
#pragma unroll
        for (int p = 0; p < JT; ++p) dma16(hsrc[p] + (int64_t)g0 * 64, lds_addr(lds + (JT * wave + p) * 256));
#pragma unroll
        for (int t = 0; t < 4; ++t) dma16(vsrc[t] + (int64_t)g0 * 64, lds_addr(vt + t * 256));
    }
    dma_wait_all();
    __syncthreads();

    int cur = 0;
    for (int g = g0; g < g1; ++g) {
        // this wave's 16 x 64 slice of V: lane (x, q) holds V[x][16q + 4i + e] in vf[i].e
#pragma unroll
        for (int i = 0; i < 4; ++i)
            vf[i] = *reinterpret_cast<const float4*>(vt + x * 64 + 4 * ((4 * q + i) ^ gx));
#pragma unroll
        for (int i = 0; i < 4; ++i) pin(vf[i]);       // reads retired before the tile is refilled
        if (g + 1 < g1) {
            float* nb = lds + (cur ^ 1) * (KP * 64);
#pragma unroll
            for (int p = 0; p < JT; ++p) dma16(hsrc[p] + (int64_t)(g + 1) * 64, lds_addr(nb + (JT * wave + p) * 256));
#pragma unroll
            for (int t = 0; t < 4; ++t) dma16(vsrc[t] + (int64_t)(g + 1) * 64, lds_addr(vt + t * 256));
        }
        const float* buf = lds + cur * (KP * 64);

        if (WITH_OBJ || KL) {
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
#pragma unroll
            for (int u = 0; u < JT; ++u) {
                const float wv[4] = {wf[u].x, wf[u].y, wf[u].z, wf[u].w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 ha = *reinterpret_cast<const float4*>(buf + u * 16 * 64 + dslot[s]);
                    d0 = MFMA(ha.x, wv[s], d0);
                    d1 = MFMA(ha.y, wv[s], d1);
                    d2 = MFMA(ha.z, wv[s], d2);
                    d3 = MFMA(ha.w, wv[s], d3);
                }
            }
            float part = 0.f;
            if (!KL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float rx = vf[i].x - d0[i], ry = vf[i].y - d1[i];
                    const float rz = vf[i].z - d2[i], rw = vf[i].w - d3[i];
                    part += rx * rx + ry * ry + rz * rz + rw * rw;
                }
            } else {
                // KL: objective term v log(v/wh) [inf, nan -> 0] - v + wh (utils.py:23-26) and
                // the quotient v / (wh + 1e-9) that replaces V in the A-product (mur.py:25)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float vv[4] = {vf[i].x, vf[i].y, vf[i].z, vf[i].w};
                    const float pp[4] = {d0[i], d1[i], d2[i], d3[i]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (WITH_OBJ) {
                            float t = vv[e] * __logf(vv[e] / pp[e]);
                            t = (t != t || t == __builtin_inff()) ? 0.f : t;
                            part += (t - vv[e]) + pp[e];
                        }
                        vv[e] = vv[e] / (pp[e] + 1e-9f);
                    }
                    vf[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
                }
            }
            if (WITH_OBJ) osum += (double)part;
        }
        if (WITH_A) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 hb[JT];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    hb[jt] = *reinterpret_cast<const float4*>(
                        buf + (jt * 16 + x) * 64 + 4 * ((4 * q + i) ^ gx));
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[jt] = MFMA(vf[i].x, hb[jt].x, acc[jt]);
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[jt] = MFMA(vf[i].y, hb[jt].y, acc[jt]);
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[jt] = MFMA(vf[i].z, hb[jt].z, acc[jt]);
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[jt] = MFMA(vf[i].w, hb[jt].w, acc[jt]);
            }
        }
        dma_wait_all();          // next tiles have landed (issued a whole group of MFMA work ago)
        __syncthreads();
        cur ^= 1;
    }

    if (WITH_A) {
        float* out = Apart + ((int64_t)sp * mp + r0) * KP;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)(4 * q + r) * KP + jt * 16 + x] = acc[jt][r];
    }
    if (WITH_OBJ) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) osum += __shfl_down(osum, off, 64);
        double* red = reinterpret_cast<double*>(lds);      // all tile reads are done
        if (lane == 0) red[wave] = osum;
        __syncthreads();
        if (tid == 0)
            objpart[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] =
                (KL ? 1.0 : 0.5) * (((red[0] + red[1]) + red[2]) + red[3]);
    }
}

// --------------------------------------------------------------------------
// KL-loss ADMM: elementwise update of the m x n auxiliaries (nmf/ao_admm.py:90-95,
// nmf/admm.py:312-315) fused with the product it needs:
//   P = Wsrc Hsrc;  v_bar = P - dual_v;  v_aux = ((v_bar-1) + sqrt((v_bar-1)^2 + 4 V)) / 2
//   dual_v += v_aux - P
// Only dual_v and S = v_aux + dual_v (the matrix the next Gram right-hand side
// multiplies, ao_admm.py:85) are stored.  Same tiling as wphase; the P tile comes out
// of the MFMA in the register layout of the V / dual_v slices.
// --------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void kl_vaux_kernel(
    const float* __restrict__ V, float* __restrict__ DV, float* __restrict__ S, int64_t ldv,
    const float* __restrict__ W, const float* __restrict__ H, int64_t ldh, int ngroups,
    const int* __restrict__ flag, const int* __restrict__ flag2)
{
    if (*flag || (flag2 && *flag2)) return;
    constexpr int JT = KP / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 H tiles + 4 V tiles + 4 dual_v tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, q = lane >> 4;
    const int Sg = gridDim.y, sp = blockIdx.y;
    const int g0 = (int)((int64_t)ngroups * sp / Sg);
    const int g1 = (int)((int64_t)ngroups * (sp + 1) / Sg);
    const int64_t r0 = (int64_t)blockIdx.x * 64 + wave * 16;
    float4 wf[JT];
#pragma unroll
    for (int u = 0; u < JT; ++u)
        wf[u] = *reinterpret_cast<const float4*>(W + (r0 + x) * KP + 16 * u + 4 * q);
#pragma unroll
    for (int u = 0; u < JT; ++u) pin(wf[u]);
    float* vt = lds + 2 * KP * 64 + wave * (16 * 64);
    float* dt = lds + 2 * KP * 64 + 4 * (16 * 64) + wave * (16 * 64);
    const float* hsrc[JT];
    int64_t voff[4];
#pragma unroll
    for (int p = 0; p < JT; ++p) {
        const int row = 4 * (JT * wave + p) + q;
        hsrc[p] = H + (int64_t)row * ldh + 4 * (x ^ swz(row));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = 4 * t + q;
        voff[t] = (r0 + row) * ldv + 4 * (x ^ swz(row));
    }
    const int gx = swz(x);
    int dslot[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) dslot[s] = (4 * q + s) * 64 + 4 * (x ^ swz(4 * q + s));
    if (g0 < g1) {
#pragma unroll
        for (int p = 0; p < JT; ++p) dma16(hsrc[p] + (int64_t)g0 * 64, lds_addr(lds + (JT * wave + p) * 256));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dma16(V + voff[t] + (int64_t)g0 * 64, lds_addr(vt + t * 256));
            dma16(DV + voff[t] + (int64_t)g0 * 64, lds_addr(dt + t * 256));
        }
    }
    dma_wait_all();
    __syncthreads();
    int cur = 0;
    for (int g = g0; g < g1; ++g) {
        float4 vf[4], df[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vf[i] = *reinterpret_cast<const float4*>(vt + x * 64 + 4 * ((4 * q + i) ^ gx));
            df[i] = *reinterpret_cast<const float4*>(dt + x * 64 + 4 * ((4 * q + i) ^ gx));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { pin(vf[i]); pin(df[i]); }
        if (g + 1 < g1) {
            float* nb = lds + (cur ^ 1) * (KP * 64);
#pragma unroll
            for (int p = 0; p < JT; ++p) dma16(hsrc[p] + (int64_t)(g + 1) * 64, lds_addr(nb + (JT * wave + p) * 256));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dma16(V + voff[t] + (int64_t)(g + 1) * 64, lds_addr(vt + t * 256));
                dma16(DV + voff[t] + (int64_t)(g + 1) * 64, lds_addr(dt + t * 256));
            }
        }
        const float* buf = lds + cur * (KP * 64);
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
#pragma unroll
        for (int u = 0; u < JT; ++u) {
            const float wv[4] = {wf[u].x, wf[u].y, wf[u].z, wf[u].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 ha = *reinterpret_cast<const float4*>(buf + u * 16 * 64 + dslot[s]);
                d0 = MFMA(ha.x, wv[s], d0);
                d1 = MFMA(ha.y, wv[s], d1);
                d2 = MFMA(ha.z, wv[s], d2);
                d3 = MFMA(ha.w, wv[s], d3);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float vv[4] = {vf[i].x, vf[i].y, vf[i].z, vf[i].w};
            const float dd[4] = {df[i].x, df[i].y, df[i].z, df[i].w};
            const float pp[4] = {d0[i], d1[i], d2[i], d3[i]};
            float dn[4], sn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (pp[e] - dd[e]) - 1.f;
                const float va = 0.5f * (t + sqrtf(t * t + 4.f * vv[e]));
                dn[e] = dd[e] + va - pp[e];
                sn[e] = va + dn[e];
            }
            const int64_t o = (r0 + x) * ldv + (int64_t)g * 64 + 16 * q + 4 * i;
            *reinterpret_cast<float4*>(DV + o) = make_float4(dn[0], dn[1], dn[2], dn[3]);
            *reinterpret_cast<float4*>(S + o) = make_float4(sn[0], sn[1], sn[2], sn[3]);
        }
        dma_wait_all();
        __syncthreads();
        cur ^= 1;
    }
}

// --------------------------------------------------------------------------
// hphase: one block = 64 columns x the rows of split sr; its 4 waves take a
// quarter of those rows each and are summed through LDS in a fixed order
// (run-to-run bit-stable).  Per k-step (4 rows) a lane loads 16 B of V
// (4 rows x 256 B per wave instruction) and KP/16 dwords of W (L2-resident).
// Output tile e of the MFMA holds columns 4x+e, so one lane owns 4 adjacent
// columns of a B row and stores them as one dwordx4.
// --------------------------------------------------------------------------
template <int KP, bool WITH_G>
__global__ __launch_bounds__(256) void hphase_kernel(
    const float* __restrict__ V, int64_t ldv, const float* __restrict__ W,
    float* __restrict__ Bpart, float* __restrict__ Gpart, int64_t np, int64_t mp,
    const int* __restrict__ flag, const int* __restrict__ flag2)
{
    if (*flag || (flag2 && *flag2)) return;
    constexpr int JT = KP / 16;
    // WITH_G: column block t < JT*JT also accumulates tile (t / JT, t % JT) of
    // W^T W from the W fragments it loads anyway (one extra MFMA per k-step).
    const int gt = WITH_G ? (int)blockIdx.x : JT * JT;
    const bool do_g = WITH_G && gt < JT * JT;
    const int ga = gt / JT, gb = gt % JT;
    f32x4 gacc = {0.f, 0.f, 0.f, 0.f};
    extern __shared__ __attribute__((aligned(16))) float lds[];   // KP*4*64 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, q = lane >> 4;
    const int SR = gridDim.y, sr = blockIdx.y;
    const int64_t n16 = mp / 16;                       // 16-row units
    const int64_t u0 = n16 * sr / SR, u1 = n16 * (sr + 1) / SR;
    const int64_t steps = (u1 - u0) * 4;               // 4-row k-steps in this split
    const int64_t s0 = steps * wave / 4, s1 = steps * (wave + 1) / 4;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const float* vp = V + (u0 * 16 + s0 * 4 + q) * ldv + c0 + 4 * x;
    const float* wp = W + (u0 * 16 + s0 * 4 + q) * KP + x;

    f32x4 acc[JT][4];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int UN = 4;                              // k-steps per software-pipeline stage
    float4 vb[UN], vbn[UN];
    float wa[UN][JT], wan[UN][JT];
    int64_t s = s0;
    const int64_t sfull = s0 + ((s1 - s0) / UN) * UN;
    if (s < sfull) {
#pragma unroll
        for (int t = 0; t < UN; ++t) {
            vb[t] = *reinterpret_cast<const float4*>(vp + (int64_t)t * 4 * ldv);
#pragma unroll
            for (int j = 0; j < JT; ++j) wa[t][j] = wp[(int64_t)t * 4 * KP + 16 * j];
        }
    }
    for (; s < sfull; s += UN) {
        {   // branch-free prefetch of the next stage (the last one re-reads its own rows)
            const int64_t adv = (s + UN < sfull) ? UN : 0;
            const float* vq = vp + adv * 4 * ldv;
            const float* wq = wp + adv * 4 * KP;
#pragma unroll
            for (int t = 0; t < UN; ++t) {
                vbn[t] = *reinterpret_cast<const float4*>(vq + (int64_t)t * 4 * ldv);
#pragma unroll
                for (int j = 0; j < JT; ++j) wan[t][j] = wq[(int64_t)t * 4 * KP + 16 * j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < UN; ++t) {
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                acc[j][0] = MFMA(wa[t][j], vb[t].x, acc[j][0]);
                acc[j][1] = MFMA(wa[t][j], vb[t].y, acc[j][1]);
                acc[j][2] = MFMA(wa[t][j], vb[t].z, acc[j][2]);
                acc[j][3] = MFMA(wa[t][j], vb[t].w, acc[j][3]);
            }
            if (do_g) {
                float fa = 0.f, fb = 0.f;
#pragma unroll
                for (int j = 0; j < JT; ++j) { fa = (j == ga) ? wa[t][j] : fa; fb = (j == gb) ? wa[t][j] : fb; }
                gacc = MFMA(fa, fb, gacc);
            }
        }
        vp += (int64_t)UN * 4 * ldv;
        wp += (int64_t)UN * 4 * KP;
#pragma unroll
        for (int t = 0; t < UN; ++t) {
            vb[t] = vbn[t];
#pragma unroll
            for (int j = 0; j < JT; ++j) wa[t][j] = wan[t][j];
        }
    }
    for (; s < s1; ++s) {                               // remainder k-steps
        const float4 v1 = *reinterpret_cast<const float4*>(vp);
        float fa = 0.f, fb = 0.f;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const float a = wp[16 * j];
            fa = (j == ga) ? a : fa; fb = (j == gb) ? a : fb;
            acc[j][0] = MFMA(a, v1.x, acc[j][0]);
            acc[j][1] = MFMA(a, v1.y, acc[j][1]);
            acc[j][2] = MFMA(a, v1.z, acc[j][2]);
            acc[j][3] = MFMA(a, v1.w, acc[j][3]);
        }
        if (do_g) gacc = MFMA(fa, fb, gacc);
        vp += 4 * ldv;
        wp += 4 * KP;
    }

    // cross-wave sum in wave order 0,1,2,3 (each lane owns its LDS words)
    float4* red = reinterpret_cast<float4*>(lds);      // [JT*4][64] float4
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            if (w == 0) {
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(j * 4 + r) * 64 + lane] =
                            make_float4(acc[j][0][r], acc[j][1][r], acc[j][2][r], acc[j][3][r]);
            } else if (w < 3) {
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float4 t = red[(j * 4 + r) * 64 + lane];
                        t.x += acc[j][0][r]; t.y += acc[j][1][r]; t.z += acc[j][2][r]; t.w += acc[j][3][r];
                        red[(j * 4 + r) * 64 + lane] = t;
                    }
            } else {
                float* out = Bpart + (int64_t)sr * KP * np + c0 + 4 * x;
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float4 t = red[(j * 4 + r) * 64 + lane];
                        t.x += acc[j][0][r]; t.y += acc[j][1][r]; t.z += acc[j][2][r]; t.w += acc[j][3][r];
                        *reinterpret_cast<float4*>(out + (int64_t)(16 * j + 4 * q + r) * np) = t;
                    }
            }
        }
        __syncthreads();
    }
    if (do_g) {                                         // same fixed-order sum for the Gram tile
        f32x4* gred = reinterpret_cast<f32x4*>(lds);
#pragma unroll 1
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
                if (w == 0) gred[lane] = gacc;
                else if (w < 3) gred[lane] += gacc;
                else {
                    const f32x4 t = gred[lane] + gacc;
                    float* o = Gpart + (int64_t)sr * KP * KP;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[(int64_t)(16 * ga + 4 * q + r) * KP + 16 * gb + x] = t[r];
                }
            }
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------
// Gram kernels (k x k outputs; tiny next to the V-sized products).  One wave
// per block; block (split s, tile row jt) writes rows [16 jt, 16 jt + 16) of
// its partial Gram matrix.
// --------------------------------------------------------------------------
// Fixed-order sum of the 4 waves' tile rows through LDS, then the store.
template <int KP>
__device__ __forceinline__ void gram_block_store(f32x4 (&acc)[KP / 16], float* __restrict__ o, int jt,
                                                 f32x4* red)
{
    constexpr int JT = KP / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                if (w == 0) red[j * 64 + lane] = acc[j];
                else if (w < 3) red[j * 64 + lane] += acc[j];
                else {
                    const f32x4 t = red[j * 64 + lane] + acc[j];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[(int64_t)(16 * jt + 4 * q + r) * KP + 16 * j + x] = t[r];
                }
            }
        }
        __syncthreads();
    }
}

template <int KP>
__global__ __launch_bounds__(256) void gram_tn_kernel(    // X^T X, X [rows][KP]
    const float* __restrict__ X, int64_t rows, float* __restrict__ out, const int* __restrict__ flag)
{
    if (*flag) return;
    constexpr int JT = KP / 16;
    __shared__ f32x4 red[JT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
    const int S = gridDim.x, s = blockIdx.x, jt = blockIdx.y;
    const int64_t n4 = rows / 4;
    const int64_t b0 = n4 * s / S, b1 = n4 * (s + 1) / S;
    const int64_t t0 = b0 + (b1 - b0) * wave / 4, t1 = b0 + (b1 - b0) * (wave + 1) / 4;
    f32x4 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* p = X + (t0 * 4 + q) * KP + x;
    int64_t t = t0;
    for (; t + 4 <= t1; t += 4, p += 16 * KP) {
        float a[4], b[4][JT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = p[u * 4 * KP + 16 * jt];
#pragma unroll
            for (int j = 0; j < JT; ++j) b[u][j] = p[u * 4 * KP + 16 * j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < JT; ++j) acc[j] = MFMA(a[u], b[u][j], acc[j]);
    }
    for (; t < t1; ++t, p += 4 * KP) {
        const float a = p[16 * jt];
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[j] = MFMA(a, p[16 * j], acc[j]);
    }
    gram_block_store<KP>(acc, out + (int64_t)s * KP * KP, jt, red);
}

template <int KP>
__global__ __launch_bounds__(256) void gram_nt_kernel(    // X X^T, X [KP][ld]
    const float* __restrict__ X, int64_t cols, int64_t ld, float* __restrict__ out,
    const int* __restrict__ flag)
{
    if (*flag) return;
    constexpr int JT = KP / 16;
    __shared__ f32x4 red[JT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
    const int S = gridDim.x, s = blockIdx.x, jt = blockIdx.y;
    const int64_t n16 = cols / 16;
    const int64_t b0 = n16 * s / S, b1 = n16 * (s + 1) / S;
    const int64_t t0 = b0 + (b1 - b0) * wave / 4, t1 = b0 + (b1 - b0) * (wave + 1) / 4;
    f32x4 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* p = X + (int64_t)x * ld + t0 * 16 + 4 * q;
#pragma unroll 2
    for (int64_t t = t0; t < t1; ++t, p += 16) {
        const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)16 * jt * ld);
        float4 b[JT];
#pragma unroll
        for (int j = 0; j < JT; ++j) b[j] = *reinterpret_cast<const float4*>(p + (int64_t)16 * j * ld);
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[j] = MFMA(a.x, b[j].x, acc[j]);
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[j] = MFMA(a.y, b[j].y, acc[j]);
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[j] = MFMA(a.z, b[j].z, acc[j]);
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[j] = MFMA(a.w, b[j].w, acc[j]);
    }
    gram_block_store<KP>(acc, out + (int64_t)s * KP * KP, jt, red);
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
template <int KP>
static int wphase_dispatch(nmfx_engine* E, const float* W, bool with_a, bool with_obj, bool kl, const float* Hsrc,
                           const float* Vsrc, const int* flag2) {
    if (with_obj) E->obj_count = (int64_t)(E->mp / 64) * E->wsplit;
    dim3 grid((unsigned)(E->mp / 64), (unsigned)E->wsplit), block(256);
    const size_t shm = (size_t)(2 * KP * 64 + 4 * 16 * 64) * sizeof(float);
    const int ng = (int)(E->np / 64);
#define NMFX_WLAUNCH(A, O, K)                                                                  \
    do {                                                                                       \
        int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(wphase_kernel<KP, A, O, K>), (int)shm); \
        if (rc_) return rc_;                                                                   \
        hipLaunchKernelGGL((wphase_kernel<KP, A, O, K>), grid, block, shm, E->stream, Vsrc, E->np, W, \
                           Hsrc, E->np, E->A_part, E->obj_part, E->mp, ng, &E->state->flag, flag2); \
    } while (0)
    if (kl) {
        if (with_a && with_obj) NMFX_WLAUNCH(true, true, true);
        else if (with_a) NMFX_WLAUNCH(true, false, true);
        else NMFX_WLAUNCH(false, true, true);
    } else {
        if (with_a && with_obj) NMFX_WLAUNCH(true, true, false);
        else if (with_a) NMFX_WLAUNCH(true, false, false);
        else NMFX_WLAUNCH(false, true, false);
    }
#undef NMFX_WLAUNCH
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// Resident workgroups per CU of the two V-sized kernels (register / LDS bound);
// nmfx_create sizes the grids to a whole number of resident waves of blocks so
// that no partially filled tail round is left.
template <int KP>
static void occupancy_of(int* wocc, int* hocc) {
    int a = 0, b = 0;
    if ((size_t)(2 * KP * 64 + 4 * 16 * 64) * sizeof(float) > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wphase_kernel<KP, true, true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)((2 * KP * 64 + 4 * 16 * 64) * sizeof(float)));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, wphase_kernel<KP, true, true, false>, 256,
                                                     (size_t)(2 * KP * 64 + 4 * 16 * 64) * sizeof(float)) != hipSuccess) a = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, hphase_kernel<KP, true>, 256,
                                                     (size_t)KP * 64 * sizeof(float)) != hipSuccess) b = 2;
    *wocc = std::max(1, a); *hocc = std::max(1, b);
}

void nmfx_phase_occupancy(int kp, int* wocc, int* hocc) {
    // (called by nmfx_create of every handle, possibly from several threads: the attribute + occupancy queries are
    // serialised, answered once per device and padded rank, and must not leave a stale error for the next
    // hipGetLastError of the calling thread -- two concurrent creates with kp = 128 did exactly that)
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::pair<int, int>> cache;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto it = cache.find(std::make_pair(dev, kp));
    if (it == cache.end()) {
        int a = 2, b = 2;
        switch (kp) {
            case 16: occupancy_of<16>(&a, &b); break;
            case 32: occupancy_of<32>(&a, &b); break;
            case 64: occupancy_of<64>(&a, &b); break;
            default: occupancy_of<128>(&a, &b); break;
        }
        (void)hipGetLastError();
        it = cache.emplace(std::make_pair(dev, kp), std::make_pair(a, b)).first;
    }
    *wocc = it->second.first; *hocc = it->second.second;
}

int nmfx_launch_wphase(nmfx_engine* E, const float* W, bool with_a, bool with_obj, bool kl, const float* Hsrc,
                       const float* Vsrc, const int* flag2) {
    ProfScope ps(E, with_a ? (with_obj ? "wphase" : "wphase_noobj") : "objective");
    if (!Hsrc) Hsrc = E->H;
    if (!Vsrc) { int rc_ = nmfx_need_v(E); if (rc_) return rc_; Vsrc = E->V; }
    switch (E->kp) {
        case 16: return wphase_dispatch<16>(E, W, with_a, with_obj, kl, Hsrc, Vsrc, flag2);
        case 32: return wphase_dispatch<32>(E, W, with_a, with_obj, kl, Hsrc, Vsrc, flag2);
        case 64: return wphase_dispatch<64>(E, W, with_a, with_obj, kl, Hsrc, Vsrc, flag2);
        case 128: return wphase_dispatch<128>(E, W, with_a, with_obj, kl, Hsrc, Vsrc, flag2);
    }
    E->err = "unsupported padded rank";
    return NMFX_E_ARG;
}

template <int KP>
static int hphase_dispatch(nmfx_engine* E, const float* W, bool with_g, const float* Vsrc, const int* flag2) {
    dim3 grid((unsigned)(E->np / 64), (unsigned)E->hsplit), block(256);
    const size_t shm = (size_t)KP * 64 * sizeof(float);
    if (with_g)
        hipLaunchKernelGGL((hphase_kernel<KP, true>), grid, block, shm, E->stream, Vsrc, E->np, W,
                           E->B_part, E->G_part, E->np, E->mp, &E->state->flag, flag2);
    else
        hipLaunchKernelGGL((hphase_kernel<KP, false>), grid, block, shm, E->stream, Vsrc, E->np, W,
                           E->B_part, E->G_part, E->np, E->mp, &E->state->flag, flag2);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// B_part[sr] = W^T V; with_g: also G_part[sr] = W^T W (needs np/64 >= (kp/16)^2,
// see nmfx_hphase_can_fuse_gram).
int nmfx_launch_hphase(nmfx_engine* E, const float* W, bool with_g, const float* Vsrc, const int* flag2) {
    ProfScope ps(E, "hphase");
    if (!Vsrc) { int rc_ = nmfx_need_v(E); if (rc_) return rc_; Vsrc = E->V; }
    switch (E->kp) {
        case 16: return hphase_dispatch<16>(E, W, with_g, Vsrc, flag2);
        case 32: return hphase_dispatch<32>(E, W, with_g, Vsrc, flag2);
        case 64: return hphase_dispatch<64>(E, W, with_g, Vsrc, flag2);
        case 128: return hphase_dispatch<128>(E, W, with_g, Vsrc, flag2);
    }
    E->err = "unsupported padded rank";
    return NMFX_E_ARG;
}

bool nmfx_hphase_can_fuse_gram(const nmfx_engine* E) {
    const int64_t jt = E->kp / 16;
    return E->np / 64 >= jt * jt;
}

int nmfx_launch_gram_tn(nmfx_engine* E, const float* X, int64_t rows, float* out, int splits) {
    ProfScope ps(E, "gram_tn");
    dim3 grid((unsigned)splits, (unsigned)(E->kp / 16)), block(256);
    switch (E->kp) {
        case 16: hipLaunchKernelGGL((gram_tn_kernel<16>), grid, block, 0, E->stream, X, rows, out, &E->state->flag); break;
        case 32: hipLaunchKernelGGL((gram_tn_kernel<32>), grid, block, 0, E->stream, X, rows, out, &E->state->flag); break;
        case 64: hipLaunchKernelGGL((gram_tn_kernel<64>), grid, block, 0, E->stream, X, rows, out, &E->state->flag); break;
        case 128: hipLaunchKernelGGL((gram_tn_kernel<128>), grid, block, 0, E->stream, X, rows, out, &E->state->flag); break;
        default: E->err = "unsupported padded rank"; return NMFX_E_ARG;
    }
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_launch_gram_nt(nmfx_engine* E, const float* X, int64_t cols, int64_t ld, float* out, int splits) {
    ProfScope ps(E, "gram_nt");
    dim3 grid((unsigned)splits, (unsigned)(E->kp / 16)), block(256);
    switch (E->kp) {
        case 16: hipLaunchKernelGGL((gram_nt_kernel<16>), grid, block, 0, E->stream, X, cols, ld, out, &E->state->flag); break;
        case 32: hipLaunchKernelGGL((gram_nt_kernel<32>), grid, block, 0, E->stream, X, cols, ld, out, &E->state->flag); break;
        case 64: hipLaunchKernelGGL((gram_nt_kernel<64>), grid, block, 0, E->stream, X, cols, ld, out, &E->state->flag); break;
        case 128: hipLaunchKernelGGL((gram_nt_kernel<128>), grid, block, 0, E->stream, X, cols, ld, out, &E->state->flag); break;
        default: E->err = "unsupported padded rank"; return NMFX_E_ARG;
    }
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int KP>
static int kl_vaux_dispatch(nmfx_engine* E, const float* Wsrc, const float* Hsrc, const int* flag2) {
    dim3 grid((unsigned)(E->mp / 64), (unsigned)E->wsplit), block(256);
    const size_t shm = (size_t)(2 * KP * 64 + 8 * 16 * 64) * sizeof(float);
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(kl_vaux_kernel<KP>), (int)shm); if (rc_) return rc_; }
    { int rc_ = nmfx_need_v(E); if (rc_) return rc_; }
    hipLaunchKernelGGL((kl_vaux_kernel<KP>), grid, block, shm, E->stream, E->V, E->DV, E->S, E->np, Wsrc, Hsrc,
                       E->np, (int)(E->np / 64), &E->state->flag, flag2);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// dual_v, S <- KL-ADMM update with P = Wsrc Hsrc (see kl_vaux_kernel)
int nmfx_launch_kl_vaux(nmfx_engine* E, const float* Wsrc, const float* Hsrc, const int* flag2) {
    ProfScope ps(E, "kl_vaux");
    switch (E->kp) {
        case 16: return kl_vaux_dispatch<16>(E, Wsrc, Hsrc, flag2);
        case 32: return kl_vaux_dispatch<32>(E, Wsrc, Hsrc, flag2);
        case 64: return kl_vaux_dispatch<64>(E, Wsrc, Hsrc, flag2);
        default: return kl_vaux_dispatch<128>(E, Wsrc, Hsrc, flag2);
    }
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_products() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(gram_nt_kernel<16>)) == hipSuccess ? 0 : -1; }
