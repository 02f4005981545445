// ANLS: alternating non-negative least squares.
//   reference: nmf/anls.py:18-47 (w_update / h_update: one NNLS per row of W /
//   column of H on the stacked system [F; sqrt(2 lambda) I]), driver :111-122;
//   scipy.optimize.nnls (Lawson-Hanson) or nmf/fcnnls.py (Van Benthem-Keenan).
//
// Every NNLS problem of one half-step shares the Gram matrix G = F^T F + 2 lambda I
// and differs in its right-hand side r = F^T b (a column of W^T V, or a row of
// V H^T): those come from the same MFMA kernels as the other solvers.  The
// constrained solve itself is an exact active-set method, block principal pivoting
// (Kim & Park), which reaches the same KKT point as Lawson-Hanson / FCNNLS (the
// minimiser is unique for positive definite G):
//     F = passive set;  x_F = G_FF^-1 r_F, x_G = 0;  y_G = G_GF x_F - r_G, y_F = 0
//     infeasible = {i in F: x_i < 0} u {i in G: y_i < 0};  exchange them (all at once,
//     with the usual back-up rule: after 3 non-improving full exchanges only the
//     largest infeasible index) until none is left.
// One wavefront owns one right-hand side, so the data-dependent control flow is
// wave-uniform; variable i lives on lane i (and i+64 for k > 64).  The k x k solve
// is a Gauss-Jordan elimination over the passive pivots in a per-wave LDS
// workspace (no pivoting: G_FF is SPD).
#include "nmfx_internal.h"
#include "kernels_small.h"

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Infeasibility is tested against a rounding-level threshold (1e-6 of the largest |x| / |r| of the
// problem): a variable whose exact x_i (or dual y_i) is zero otherwise flips sign with the f32
// rounding of every solve and the exchange rule cycles until the iteration cap (seen on 0.03 % of
// the right-hand sides at 16384 x 8192 -- they set the run time of the whole launch).
#define NMFX_NNLS_TOL 1e-6f
// A passive pivot that has shrunk to this fraction of the variable's own diagonal entry is treated as zero: the
// variable is dropped from the passive set for the rest of the solve and keeps x = 0.  That is what happens to a
// dead component (a zero column of the fixed factor: G_pp = 0) or to the second of two collinear ones at
// lambda = 0 -- the Gram matrix is then singular on the passive set, the unguarded 1 / pivot made the whole
// right-hand side NaN and the solve silently returned zeros.  Lawson-Hanson and FCNNLS leave such a variable
// at zero as well (its dual is zero, it is never selected).  (NaN pivots fail the comparison and are dropped too.)
#define NMFX_NNLS_PIVOT_EPS 1e-6f

template <int KP>
__global__ __launch_bounds__(KP <= 64 ? 256 : 128) void nnls_bpp_kernel(
    const float* __restrict__ G, float diag_add, const float* __restrict__ R, float* __restrict__ X,
    int64_t sj, int64_t sc, int64_t nprob, int k, DevState* __restrict__ st, const int* __restrict__ todo,
    const int* __restrict__ inv_bad)
{
    if (st->flag) return;
    constexpr int NV = KP <= 64 ? 1 : KP / 64;          // variables per lane
    constexpr int NW = KP <= 64 ? 4 : 2;                // waves (problems) per block
    constexpr int LDM = KP + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * NW + wave;
    if (c >= nprob) return;                              // whole wave leaves together
    if (todo && !*inv_bad && !todo[c]) return;           // solved by nnls_cinv_kernel
    float* M = lds + (size_t)wave * (KP * LDM + KP);     // [KP][KP+1] augmented rows
    float* xs = M + KP * LDM;                            // [KP] broadcast copy of x

    int idx[NV]; bool valid[NV], inF[NV], dead[NV];
    float r[NV], x[NV], y[NV];
    // WARM START: the passive set starts as the support of the previous solution (X on entry: the
    // factor of the last outer iteration, or the initial factor).  Block principal pivoting reaches the
    // unique KKT point from any start; once the supports settle a half-step needs one or two solves
    // per right-hand side instead of a cold start's five to ten.
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        idx[t] = lane + 64 * t;
        valid[t] = idx[t] < k;
        r[t] = valid[t] ? R[(int64_t)idx[t] * sj + c * sc] : 0.f;
        x[t] = 0.f; y[t] = -r[t]; dead[t] = false;
        inF[t] = valid[t] && X[(int64_t)idx[t] * sj + c * sc] > 0.f;
    }
    float toly;
    {
        float ar = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) ar = fmaxf(ar, fabsf(r[t]));
        toly = NMFX_NNLS_TOL * wave_max(ar);
    }
    int best = k + 1, spare = 3, iter = 0;
    for (; iter < 8 * KP + 64; ++iter) {
        unsigned long long Im[NV];
        if (iter > 0) {                                  // (iteration 0 solves for the warm-start set first)
        float ax = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) ax = fmaxf(ax, fabsf(x[t]));
        const float tolx = NMFX_NNLS_TOL * wave_max(ax);
        int n_inf = 0;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const bool bad = valid[t] && !dead[t] && (inF[t] ? (x[t] < -tolx) : (y[t] < -toly));
            Im[t] = __ballot(bad);
            n_inf += __popcll(Im[t]);
        }
        if (n_inf == 0) break;
        bool full = true;
        if (n_inf < best) { best = n_inf; spare = 3; }
        else if (spare > 0) { --spare; }
        else full = false;
        if (!full) {                                     // back-up rule: largest infeasible index only
#pragma unroll
            for (int t = NV - 1; t >= 0; --t) {
                if (Im[t]) {
                    const int hi = 63 - __clzll((long long)Im[t]);
                    Im[t] = 1ull << hi;
#pragma unroll
                    for (int u = 0; u < t; ++u) Im[u] = 0ull;
                    break;
                }
            }
        }
        } else {
#pragma unroll
            for (int t = 0; t < NV; ++t) Im[t] = 0ull;
        }
        unsigned long long Fm[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if ((Im[t] >> lane) & 1ull) inF[t] = !inF[t];
            Fm[t] = __ballot(inF[t]);
        }
        // augmented rows of the passive variables (other rows are never touched).  ALL columns are
        // copied and eliminated: the non-passive ones multiply x = 0 and never serve as pivots, and
        // plain full-width loops keep the LDS operations independent and pipelined (the earlier
        // bit-scan over the passive columns made every read-modify-write wait for the previous one).
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (inF[t]) {
                const float* grow = G + (int64_t)idx[t] * KP;
                float* mrow = M + idx[t] * LDM;
#pragma unroll 8
                for (int cc = 0; cc < KP; ++cc) mrow[cc] = grow[cc] + (cc == idx[t] ? diag_add : 0.f);
                mrow[KP] = r[t];
            }
        }
        // Gauss-Jordan over the passive pivots (no normalisation: x_i = rhs_i / diag_i at the end)
        for (int tp = 0; tp < NV; ++tp) {
            unsigned long long left = Fm[tp];
            while (left) {
                const int b = __ffsll((long long)left) - 1;
                left &= left - 1;
                const int p = b + 64 * tp;
                const float* prow = M + p * LDM;
                if (!(prow[p] > NMFX_NNLS_PIVOT_EPS * (G[(int64_t)p * KP + p] + diag_add))) {   // vanished pivot: drop the variable
#pragma unroll
                    for (int t = 0; t < NV; ++t) if (idx[t] == p) { inF[t] = false; dead[t] = true; }
                    if (lane == 0) atomicAdd(&st->nnls_evicted, 1);
                    continue;
                }
                const float inv = 1.f / prow[p];
                const int c0 = (p + 1) & ~3;                 // columns <= p of the pivot row are already zero
#pragma unroll
                for (int t = 0; t < NV; ++t) {
                    if (inF[t] && idx[t] != p) {
                        float* mrow = M + idx[t] * LDM;
                        const float f = mrow[p] * inv;
                        for (int cc = p + 1; cc < c0; ++cc) mrow[cc] = fmaf(-f, prow[cc], mrow[cc]);
#pragma unroll 4
                        for (int cc = c0; cc < KP; cc += 4) {
                            const float p0 = prow[cc], p1 = prow[cc + 1], p2 = prow[cc + 2], p3 = prow[cc + 3];
                            const float m0 = mrow[cc], m1 = mrow[cc + 1], m2 = mrow[cc + 2], m3 = mrow[cc + 3];
                            mrow[cc] = fmaf(-f, p0, m0); mrow[cc + 1] = fmaf(-f, p1, m1);
                            mrow[cc + 2] = fmaf(-f, p2, m2); mrow[cc + 3] = fmaf(-f, p3, m3);
                        }
                        mrow[KP] = fmaf(-f, prow[KP], mrow[KP]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            x[t] = inF[t] ? M[idx[t] * LDM + KP] / M[idx[t] * LDM + idx[t]] : 0.f;
            if (idx[t] < KP) xs[idx[t]] = x[t];          // lanes beyond KP own no variable
        }
        // dual variables of the active set: y = G x - r   (x = 0 outside the passive set)
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            float acc = 0.f;
            if (valid[t] && !inF[t]) {
                const float* grow = G + (int64_t)idx[t] * KP;
#pragma unroll 8
                for (int cc = 0; cc < KP; ++cc) acc = fmaf(grow[cc], xs[cc], acc);
                acc -= r[t];
            }
            y[t] = acc;
        }
    }
    if (iter == 8 * KP + 64 && lane == 0) atomicAdd(&st->nnls_capped, 1);
#pragma unroll
    for (int t = 0; t < NV; ++t)
        if (idx[t] < KP) X[(int64_t)idx[t] * sj + c * sc] = (valid[t] && x[t] > 0.f) ? x[t] : 0.f;
}

#ifdef NMFX_NNLS_STATS          // experiment builds only (tools/anls_perf.py --stats), like the other nmfx_debug_* exports: not part of include/nmfx.h
__device__ unsigned long long nnls_dbg[8];     // [sum of iterations, max, problems, exchanges in back-up mode, pivots, final support]
#endif

// Register-resident variant for k <= 64 (one variable per lane, one right-hand side per wave):
// lane i keeps row i of G in registers, the elimination works on a register copy with the pivot
// row BROADCAST through the LDS crossbar (ds_bpermute), pivots in static order.
// No LDS storage, no data-dependent addressing: the LDS version above spends its time in serial
// read-modify-write chains (34 ms per half-step at 16384 x 8192, k = 64).
// The elimination runs over the passive pivots only but over ALL columns: the non-passive columns
// multiply x = 0, so the passive rows still end with x_i = rhs_i (pivot rows are normalised).
template <int KP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void nnls_bpp_reg_kernel(
    const float* __restrict__ G, float diag_add, const float* __restrict__ R, float* __restrict__ X,
    int64_t sj, int64_t sc, int64_t nprob, int k, DevState* __restrict__ st, const int* __restrict__ todo,
    const int* __restrict__ inv_bad)
{
    if (st->flag) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 4 + wave;
    if (c >= nprob) return;                              // whole wave leaves together
    if (todo && !*inv_bad && !todo[c]) return;           // solved by nnls_cinv_kernel
    const bool owner = lane < KP, valid = lane < k;
    float g[KP];
#pragma unroll
    for (int cc = 0; cc < KP; ++cc) g[cc] = owner ? G[(int64_t)lane * KP + cc] + (cc == lane ? diag_add : 0.f) : 0.f;
    float gd = 0.f;                                      // this variable's own diagonal entry (pivot guard)
#pragma unroll
    for (int cc = 0; cc < KP; ++cc) gd = (cc == lane) ? g[cc] : gd;
    bool dead = false;
    const float r = valid ? R[(int64_t)lane * sj + c * sc] : 0.f;
    bool inF = valid && X[(int64_t)(owner ? lane : 0) * sj + c * sc] > 0.f;      // warm start: previous support
    float x = 0.f, y = -r;
    const float toly = NMFX_NNLS_TOL * wave_max(fabsf(r));
    int best = k + 1, spare = 3;
    int iter = 0, nbackup = 0;
    for (; iter < 8 * KP + 64; ++iter) {
        if (iter > 0) {
            const float tolx = NMFX_NNLS_TOL * wave_max(fabsf(x));
            const bool bad = valid && !dead && (inF ? (x < -tolx) : (y < -toly));
            unsigned long long Im = __ballot(bad);
            const int n_inf = __popcll(Im);
            if (n_inf == 0) break;
            bool full = true;
            if (n_inf < best) { best = n_inf; spare = 3; }
            else if (spare > 0) { --spare; }
            else full = false;
            if (!full) { Im = 1ull << (63 - __clzll((long long)Im)); ++nbackup; }   // back-up rule: largest infeasible index only
            if ((Im >> lane) & 1ull) inF = !inF;
        }
        const unsigned long long Fm = __ballot(inF);
#ifdef NMFX_NNLS_STATS
        if (lane == 0) atomicAdd(&nnls_dbg[4], (unsigned long long)__popcll(Fm));
#endif
        float m[KP], rhs = r;
#pragma unroll
        for (int cc = 0; cc < KP; ++cc) m[cc] = g[cc];
        // Pivots = the passive indices, in a RUNTIME loop: unrolling the 64 pivot bodies statically made
        // ~80 KiB of code and the kernel instruction-fetch bound (230 cycles per instruction measured).
        // The row's own entry in the pivot column, m[p] with a wave-uniform runtime p, is picked by a
        // compare-select sweep; the pivot row comes through the LDS crossbar (ds_bpermute).
        unsigned long long left = Fm;
        while (left) {
            const int p = __ffsll((long long)left) - 1;
            left &= left - 1;
            float f = 0.f;
#pragma unroll
            for (int cc = 0; cc < KP; ++cc) f = (cc == p) ? m[cc] : f;
            const float piv = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * p, __float_as_int(f)));
            if (!(piv > NMFX_NNLS_PIVOT_EPS * __int_as_float(__builtin_amdgcn_ds_bpermute(4 * p, __float_as_int(gd))))) {
                if (lane == p) { inF = false; dead = true; atomicAdd(&st->nnls_evicted, 1); }      // vanished pivot: drop the variable
                continue;
            }
            const float inv = 1.f / piv;
            const bool me = lane == p;
#pragma unroll
            for (int c0 = 0; c0 < KP; c0 += 16) {
                float pr[16];
#pragma unroll
                for (int cc = 0; cc < 16; ++cc)
                    pr[cc] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * p, __float_as_int(m[c0 + cc])));
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    const float prs = pr[cc] * inv;
                    m[c0 + cc] = me ? prs : fmaf(-f, prs, m[c0 + cc]);
                }
            }
            const float prr = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * p, __float_as_int(rhs))) * inv;
            rhs = me ? prr : fmaf(-f, prr, rhs);
        }
        x = inF ? rhs : 0.f;
        float acc = -r;
#pragma unroll
        for (int cc = 0; cc < KP; ++cc)
            acc = fmaf(g[cc], __int_as_float(__builtin_amdgcn_ds_bpermute(4 * cc, __float_as_int(x))), acc);
        y = (valid && !inF) ? acc : 0.f;
    }
    if (iter == 8 * KP + 64 && lane == 0) atomicAdd(&st->nnls_capped, 1);
    if (owner) X[(int64_t)lane * sj + c * sc] = (valid && x > 0.f) ? x : 0.f;
#ifdef NMFX_NNLS_STATS
    const int nfinal = __popcll(__ballot(valid && x > 0.f));
    if (lane == 0) {
        atomicAdd(&nnls_dbg[5], (unsigned long long)nfinal);
        atomicAdd(&nnls_dbg[0], (unsigned long long)iter); atomicMax(&nnls_dbg[1], (unsigned long long)iter);
        atomicAdd(&nnls_dbg[2], 1ull); atomicAdd(&nnls_dbg[3], (unsigned long long)nbackup);
    }
#endif
}

#ifdef NMFX_NNLS_STATS
extern "C" int nmfx_debug_nnls_stats(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nnls_dbg), sizeof(nnls_dbg)) == hipSuccess ? 0 : -1;
}
#endif

// k = 128: two waves per right-hand side (thread = variable), rows in registers, the pivot row
// handed over through a double-buffered LDS row (one barrier per pivot), the masks of the two
// waves exchanged through LDS.  Same algorithm as nnls_bpp_reg_kernel.
__device__ __forceinline__ void nnls_bpp_reg128_body(
    const float* __restrict__ G, float diag_add, const float* __restrict__ R, float* __restrict__ X,
    int64_t sj, int64_t sc, int64_t nprob, int k, DevState* __restrict__ st, const int* __restrict__ todo,
    const int* __restrict__ inv_bad)
{
    if (st->flag) return;
    if (todo && !*inv_bad && !todo[blockIdx.x]) return;  // (block-uniform) solved by nnls_cinv_kernel
    constexpr int KP = 128;
    __shared__ __attribute__((aligned(16))) float prow_s[2][KP + 4];   // pivot row, [KP] = its right-hand side
    __shared__ __attribute__((aligned(16))) float xs[KP];
    __shared__ unsigned long long mask_s[2][2];
    __shared__ float red_s[2][2];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t c = blockIdx.x;
    const bool valid = tid < k;
    const float* grow = G + (int64_t)tid * KP;
    const float r = valid ? R[(int64_t)tid * sj + c * sc] : 0.f;
    bool inF = valid && X[(int64_t)tid * sj + c * sc] > 0.f;        // warm start: previous support
    bool dead = false;
    float x = 0.f, y = -r;
    int ph = 0;                                                      // parity of the small exchange buffers
    auto block_max = [&](float v) {
        v = wave_max(v);
        if ((tid & 63) == 0) red_s[ph][wave] = v;
        __syncthreads();
        const float o = fmaxf(red_s[ph][0], red_s[ph][1]);
        ph ^= 1;
        return o;
    };
    auto block_masks = [&](bool bit, unsigned long long out[2]) {
        const unsigned long long b = __ballot(bit);
        if ((tid & 63) == 0) mask_s[ph][wave] = b;
        __syncthreads();
        out[0] = mask_s[ph][0]; out[1] = mask_s[ph][1];
        ph ^= 1;
    };
    const float toly = NMFX_NNLS_TOL * block_max(fabsf(r));
    int best = k + 1, spare = 3, pbuf = 0, iter = 0;
    for (; iter < 8 * KP + 64; ++iter) {
        if (iter > 0) {
            const float tolx = NMFX_NNLS_TOL * block_max(fabsf(x));
            unsigned long long Im[2];
            block_masks(valid && !dead && (inF ? (x < -tolx) : (y < -toly)), Im);
            const int n_inf = __popcll(Im[0]) + __popcll(Im[1]);
            if (n_inf == 0) break;
            bool full = true;
            if (n_inf < best) { best = n_inf; spare = 3; }
            else if (spare > 0) { --spare; }
            else full = false;
            if (!full) {                                 // back-up rule: largest infeasible index only
                if (Im[1]) { Im[1] = 1ull << (63 - __clzll((long long)Im[1])); Im[0] = 0ull; }
                else Im[0] = 1ull << (63 - __clzll((long long)Im[0]));
            }
            if ((Im[wave] >> (tid & 63)) & 1ull) inF = !inF;
        }
        unsigned long long Fm[2];
        block_masks(inF, Fm);
        float m[KP], rhs = r;
#pragma unroll
        for (int c4 = 0; c4 < KP / 4; ++c4) {
            const float4 gv = *reinterpret_cast<const float4*>(grow + 4 * c4);
            m[4 * c4] = gv.x + (4 * c4 == tid ? diag_add : 0.f); m[4 * c4 + 1] = gv.y + (4 * c4 + 1 == tid ? diag_add : 0.f);
            m[4 * c4 + 2] = gv.z + (4 * c4 + 2 == tid ? diag_add : 0.f); m[4 * c4 + 3] = gv.w + (4 * c4 + 3 == tid ? diag_add : 0.f);
        }
        for (int tp = 0; tp < 2; ++tp) {
            unsigned long long left = Fm[tp];
            while (left) {
                const int p = __ffsll((long long)left) - 1 + 64 * tp;
                left &= left - 1;
                float* pw = prow_s[pbuf];
                if (tid == p) {
#pragma unroll
                    for (int c4 = 0; c4 < KP / 4; ++c4)
                        *reinterpret_cast<float4*>(pw + 4 * c4) = make_float4(m[4 * c4], m[4 * c4 + 1], m[4 * c4 + 2], m[4 * c4 + 3]);
                    pw[KP] = rhs;
                }
                __syncthreads();
                float f = 0.f;
#pragma unroll
                for (int cc = 0; cc < KP; ++cc) f = (cc == p) ? m[cc] : f;
                if (!(pw[p] > NMFX_NNLS_PIVOT_EPS * (G[(int64_t)p * KP + p] + diag_add))) {       // vanished pivot: drop the variable
                    if (tid == p) { inF = false; dead = true; atomicAdd(&st->nnls_evicted, 1); }
                    pbuf ^= 1;
                    continue;
                }
                const float inv = 1.f / pw[p];
                const bool me = tid == p;
#pragma unroll
                for (int c8 = 0; c8 < KP / 32; ++c8) {           // 8 x 16 bytes of the pivot row in flight at a time
#pragma unroll
                    for (int c4 = 8 * c8; c4 < 8 * c8 + 8; ++c4) {
                        const float4 pv = *reinterpret_cast<const float4*>(pw + 4 * c4);
                        const float ps[4] = {pv.x * inv, pv.y * inv, pv.z * inv, pv.w * inv};
#pragma unroll
                        for (int e = 0; e < 4; ++e) m[4 * c4 + e] = me ? ps[e] : fmaf(-f, ps[e], m[4 * c4 + e]);
                    }
                    __builtin_amdgcn_sched_barrier(0);           // keeps the row's live range short (register budget)
                }
                const float prr = pw[KP] * inv;
                rhs = me ? prr : fmaf(-f, prr, rhs);
                pbuf ^= 1;
            }
        }
        x = inF ? rhs : 0.f;
        __syncthreads();                                 // the previous iteration's readers of xs are done
        xs[tid] = x;
        __syncthreads();
        float acc = -r;
#pragma unroll 8
        for (int c4 = 0; c4 < KP / 4; ++c4) {
            const float4 gv = *reinterpret_cast<const float4*>(grow + 4 * c4);
            const float4 xv = *reinterpret_cast<const float4*>(xs + 4 * c4);
            acc = fmaf(gv.x, xv.x, acc); acc = fmaf(gv.y, xv.y, acc); acc = fmaf(gv.z, xv.z, acc); acc = fmaf(gv.w, xv.w, acc);
        }
        acc = fmaf(diag_add, x, acc);                    // (G + diag) x; x = 0 for the active variables anyway
        y = (valid && !inF) ? acc : 0.f;
    }
    if (iter == 8 * KP + 64 && tid == 0) atomicAdd(&st->nnls_capped, 1);
    X[(int64_t)tid * sj + c * sc] = (valid && x > 0.f) ? x : 0.f;
}

// Two register budgets for the same body: 256 registers (two waves per SIMD; hipcc parks ~40 dwords of the row in
// scratch) or 512 (one wave per SIMD, no scratch).  NMFX_NNLS128_OCC=1 selects the second; measured in tools/anls_perf.py.
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 3))) void nnls_bpp_reg128_kernel(
    const float* __restrict__ G, float diag_add, const float* __restrict__ R, float* __restrict__ X,
    int64_t sj, int64_t sc, int64_t nprob, int k, DevState* __restrict__ st, const int* __restrict__ todo,
    const int* __restrict__ inv_bad)
{ nnls_bpp_reg128_body(G, diag_add, R, X, sj, sc, nprob, k, st, todo, inv_bad); }
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void nnls_bpp_reg128_wide_kernel(
    const float* __restrict__ G, float diag_add, const float* __restrict__ R, float* __restrict__ X,
    int64_t sj, int64_t sc, int64_t nprob, int k, DevState* __restrict__ st, const int* __restrict__ todo,
    const int* __restrict__ inv_bad)
{ nnls_bpp_reg128_body(G, diag_add, R, X, sj, sc, nprob, k, st, todo, inv_bad); }

// ---------------------------------------------------------------------------------------------------------------------
// NNLS through the INVERSE of the Gram matrix and the COMPLEMENT of the passive set.
// The elimination kernels above solve G_FF x_F = r_F per right-hand side: |F| ~ 57 pivots over 64-entry rows at k = 64,
// 1.05 ms per half-step at 16384 x 8192 -- the largest item of an ANLS iteration.  With u = G^-1 r (the unconstrained
// solution) and C the ACTIVE set (x_C = 0; ~6 indices once the supports have settled) the same KKT point is
//     z = ((G^-1)_CC)^-1 u_C,    x_F = u_F - (G^-1)_FC z,    y_C = (G x - r)_C = -z
// (from G x - r = mu, mu_F = 0, x_C = 0), i.e. a |C| x |C| solve and |C| rank-1 corrections per exchange instead of a
// |F| x |F| elimination.  A first version with an f32 inverse was too inaccurate (G^-1 squares the condition number of
// the data matrix); here G^-1 is formed ONCE per half-step in f64 (nnls_inverse_kernel: in-place Gauss-Jordan of
// G + diag_add I in LDS, one workgroup) and every per-problem quantity is f64.  Same exchange rules, tolerances, warm
// start and iteration cap as nnls_bpp_kernel.  What this path does not take it leaves to the elimination kernels, which
// run afterwards on exactly those problems: a half-step whose Gram matrix is singular or too ill-conditioned for an
// explicit inverse (`inv_bad`: a dead or collinear component at lambda = 0 -- their pivot guard handles it), and
// single problems whose complement outgrows the workspace (todo[c] = 1).
// ---------------------------------------------------------------------------------------------------------------------
#define NMFX_NNLS_INV_EPS 1e-9            // pivot below this fraction of the mean diagonal: no explicit inverse

template <int KP>
__global__ __launch_bounds__(256) void nnls_inverse_kernel(const float* __restrict__ G, float diag_add, int k,
                                                           double* __restrict__ Ginv, int* __restrict__ inv_bad,
                                                           const DevState* __restrict__ st)
{
    if (st->flag) return;
    constexpr int LD = KP + 1;
    extern __shared__ __attribute__((aligned(16))) double inv_lds[];
    double* A = inv_lds;                                // [KP][LD]
    double* colp = A + KP * LD;                         // [KP] pivot column of the step
    __shared__ double scal[2];
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int i = tid; i < KP * KP; i += 256) {
        const int r = i / KP, c = i % KP;
        double v = (double)G[(int64_t)r * KP + c];
        if (r == c) v += (r < k) ? (double)diag_add : 1.0;       // padded variables: decoupled unit block
        A[r * LD + c] = v;
    }
    __syncthreads();
    if (tid < 64) {                                     // mean diagonal: the scale of the pivot guard
        double t = 0.0;
        for (int i = tid; i < k; i += 64) t += A[i * LD + i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (tid == 0) scal[0] = t / (double)k;
    }
    __syncthreads();
    const double floor_ = NMFX_NNLS_INV_EPS * scal[0];
    for (int p = 0; p < KP; ++p) {
        const double piv = A[p * LD + p];
        if (!(piv > floor_)) { if (tid == 0) bad = 1; }
        for (int i = tid; i < KP; i += 256) colp[i] = A[i * LD + p];
        __syncthreads();
        if (bad) break;
        const double inv = 1.0 / piv;
        // row p: scaled; every other row i: A[i][j] -= A[i][p] * A[p][j] / piv (j != p), A[i][p] = -A[i][p] / piv
        for (int i = tid; i < KP * KP; i += 256) {
            const int r = i / KP, c = i % KP;
            if (r == p) continue;
            const double f = colp[r];
            A[r * LD + c] = (c == p) ? -f * inv : fma(-f * inv, A[p * LD + c], A[r * LD + c]);
        }
        __syncthreads();
        for (int c = tid; c < KP; c += 256) A[p * LD + c] = (c == p) ? inv : A[p * LD + c] * inv;
        __syncthreads();
    }
    if (tid == 0) *inv_bad = bad;
    if (bad) return;
    for (int i = tid; i < KP * KP; i += 256) Ginv[i] = A[(i / KP) * LD + (i % KP)];
}

// SECOND = true (k = 128 only): a second pass over the problems the first one left (todo[c] = 1), one wave per block with
// the whole remaining LDS as its workspace (complements up to 56): at k = 128 a fifth of the problems of the first
// iterations have complements above 36, and the elimination kernel needs ~280 us for each of them.
template <int KP, bool SECOND = false>
__global__ __launch_bounds__(SECOND ? 64 : (KP <= 64 ? 384 : 128)) void nnls_cinv_kernel(
    const double* __restrict__ Ginv, const int* __restrict__ inv_bad, const float* __restrict__ R, float* __restrict__ X,
    int64_t sj, int64_t sc, int64_t nprob, int k, DevState* __restrict__ st, int* __restrict__ todo)
{
    if (st->flag) return;
    if (*inv_bad) { if (!SECOND && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&st->nnls_noinv, 1); return; }
    constexpr int NV = KP <= 64 ? 1 : KP / 64;          // variables per lane
    constexpr int NW = SECOND ? 1 : (KP <= 64 ? 6 : 2); // waves per block (k = 128: G^-1 alone takes 128 KiB of LDS)
    constexpr int MC = SECOND ? 56 : (KP <= 64 ? 28 : 36);   // largest complement solved here
    constexpr int LDS_ = MC + 1;
    constexpr int WSZ = MC * LDS_ + MC + KP;            // per wave: S [MC][MC + 1] | z [MC] | r broadcast [KP]
    extern __shared__ __attribute__((aligned(16))) double cinv_lds[];
    double* gi = cinv_lds;                              // G^-1 [KP][KP] (symmetric: row j is read for column j)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* S = gi + KP * KP + wave * WSZ;
    double* zv = S + MC * LDS_;
    double* rv = zv + MC;
    for (int i = threadIdx.x; i < KP * KP; i += 64 * NW) gi[i] = Ginv[i];
    __syncthreads();
    for (int64_t c = (int64_t)blockIdx.x * NW + wave; c < nprob; c += (int64_t)gridDim.x * NW) {
        if (SECOND && !todo[c]) continue;               // (wave-uniform) solved by the first pass
        int idx[NV]; bool valid[NV], inF[NV];
        double r[NV], u[NV], x[NV], y[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            idx[t] = lane + 64 * t;
            valid[t] = idx[t] < k;
            r[t] = valid[t] ? (double)R[(int64_t)idx[t] * sj + c * sc] : 0.0;
            inF[t] = valid[t] && X[(int64_t)idx[t] * sj + c * sc] > 0.f;      // warm start: the previous support
            if (idx[t] < KP) rv[idx[t]] = r[t];
        }
        __builtin_amdgcn_wave_barrier();
        float ar = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) ar = fmaxf(ar, fabsf((float)r[t]));
        const double toly = (double)(NMFX_NNLS_TOL * wave_max(ar));
#pragma unroll
        for (int t = 0; t < NV; ++t) {                  // u = G^-1 r
            double a0 = 0.0, a1 = 0.0;
            if (idx[t] < KP) {
#pragma unroll 8
                for (int j = 0; j < KP; j += 2) {
                    a0 = fma(gi[j * KP + idx[t]], rv[j], a0);
                    a1 = fma(gi[(j + 1) * KP + idx[t]], rv[j + 1], a1);
                }
            }
            u[t] = a0 + a1; x[t] = 0.0; y[t] = 0.0;
        }
        int best = k + 1, spare = 3, iter = 0;
        bool give_up = false;
        for (; iter < 8 * KP + 64; ++iter) {
            unsigned long long Im[NV];
            if (iter > 0) {
                float ax = 0.f;
#pragma unroll
                for (int t = 0; t < NV; ++t) ax = fmaxf(ax, fabsf((float)x[t]));
                const double tolx = (double)(NMFX_NNLS_TOL * wave_max(ax));
                int n_inf = 0;
#pragma unroll
                for (int t = 0; t < NV; ++t) {
                    const bool bad = valid[t] && (inF[t] ? (x[t] < -tolx) : (y[t] < -toly));
                    Im[t] = __ballot(bad);
                    n_inf += __popcll(Im[t]);
                }
                if (n_inf == 0) break;
                bool full = true;
                if (n_inf < best) { best = n_inf; spare = 3; }
                else if (spare > 0) { --spare; }
                else full = false;
                if (!full) {                            // back-up rule: largest infeasible index only
#pragma unroll
                    for (int t = NV - 1; t >= 0; --t) {
                        if (Im[t]) {
                            const int hi = 63 - __clzll((long long)Im[t]);
                            Im[t] = 1ull << hi;
#pragma unroll
                            for (int w2 = 0; w2 < t; ++w2) Im[w2] = 0ull;
                            break;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NV; ++t) Im[t] = 0ull;
            }
            unsigned long long Cm[NV];
            int m = 0, base[NV];
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                if ((Im[t] >> lane) & 1ull) inF[t] = !inF[t];
                Cm[t] = __ballot(valid[t] && !inF[t]);
                base[t] = m;
                m += __popcll(Cm[t]);
            }
            if (m > MC) { give_up = true; break; }
            // S = (G^-1)_CC with the right-hand side u_C: the owner of the s-th member fills row s
            int slot[NV];
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                slot[t] = base[t] + __popcll(Cm[t] & ((1ull << lane) - 1ull));
                const bool mine = (Cm[t] >> lane) & 1ull;
                int col = 0;
#pragma unroll
                for (int t2 = 0; t2 < NV; ++t2) {
                    unsigned long long left = Cm[t2];
                    while (left) {
                        const int j = __ffsll((long long)left) - 1 + 64 * t2;
                        left &= left - 1;
                        if (mine) S[slot[t] * LDS_ + col] = gi[j * KP + idx[t]];
                        ++col;
                    }
                }
                if (mine) S[slot[t] * LDS_ + m] = u[t];
            }
            __builtin_amdgcn_wave_barrier();
            // Gauss-Jordan on the m x m system, lane s = row s (no pivoting: a principal block of an SPD matrix)
            bool sing = false;
            for (int p = 0; p < m; ++p) {
                const double piv = S[p * LDS_ + p];
                if (!(piv > 0.0)) { sing = true; break; }
                const double inv = 1.0 / piv;
                if (lane < m && lane != p) {
                    const double f = S[lane * LDS_ + p] * inv;
                    for (int cc = p + 1; cc <= m; ++cc) S[lane * LDS_ + cc] = fma(-f, S[p * LDS_ + cc], S[lane * LDS_ + cc]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (sing) { give_up = true; break; }
            if (lane < m) zv[lane] = S[lane * LDS_ + m] / S[lane * LDS_ + lane];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const bool mine = (Cm[t] >> lane) & 1ull;
                double acc = u[t];
                int col = 0;
#pragma unroll
                for (int t2 = 0; t2 < NV; ++t2) {
                    unsigned long long left = Cm[t2];
                    while (left) {
                        const int j = __ffsll((long long)left) - 1 + 64 * t2;
                        left &= left - 1;
                        if (idx[t] < KP) acc = fma(-gi[j * KP + idx[t]], zv[col], acc);
                        ++col;
                    }
                }
                x[t] = (valid[t] && !mine) ? acc : 0.0;                  // passive: u_F - (G^-1)_FC z
                y[t] = mine ? -zv[slot[t]] : 0.0;                        // active: the dual G x - r
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (iter == 8 * KP + 64) { give_up = true; if (lane == 0) atomicAdd(&st->nnls_capped, 1); }
        if (lane == 0 && todo) todo[c] = give_up ? 1 : 0;
        if (lane == 0 && give_up && (SECOND || KP <= 64)) atomicAdd(&st->nnls_fallback, 1);      // (k = 128: counted where it is final)
        if (!give_up) {
#pragma unroll
            for (int t = 0; t < NV; ++t)
                if (idx[t] < KP) X[(int64_t)idx[t] * sj + c * sc] = (valid[t] && x[t] > 0.0) ? (float)x[t] : 0.f;
        }
    }
}

template <int KP>
static int launch_nnls_reg(nmfx_engine* E, const float* G, float diag_add, const float* R, float* X, int64_t sj,
                           int64_t sc, int64_t nprob, const int* todo, const int* bad) {
    hipLaunchKernelGGL((nnls_bpp_reg_kernel<KP>), dim3((unsigned)((nprob + 3) / 4)), dim3(256), 0, E->stream, G, diag_add,
                       R, X, sj, sc, nprob, E->k, E->state, todo, bad);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int KP>
static int launch_nnls(nmfx_engine* E, const float* G, float diag_add, const float* R, float* X, int64_t sj,
                       int64_t sc, int64_t nprob, const int* todo, const int* bad) {
    constexpr int NW = KP <= 64 ? 4 : 2;
    const size_t shm = (size_t)NW * (KP * (KP + 1) + KP) * sizeof(float);
    auto kern = nnls_bpp_kernel<KP>;
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)((nprob + NW - 1) / NW)), dim3(64 * NW), shm, E->stream, G, diag_add,
                       R, X, sj, sc, nprob, E->k, E->state, todo, bad);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <typename T>
static int lazy_alloc(nmfx_engine* E, T** p, int64_t count) {
    if (*p) return NMFX_OK;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T)));
    NMFX_HIP(hipMemsetAsync(*p, 0, (size_t)count * sizeof(T), E->stream));
    return NMFX_OK;
}

// the inverse-based pass (see nnls_cinv_kernel); leaves *todo / *bad for the elimination kernel that follows
template <int KP>
static int launch_nnls_cinv(nmfx_engine* E, const float* G, float diag_add, const float* R, float* X, int64_t sj, int64_t sc,
                            int64_t nprob, const int** todo, const int** bad) {
    int rc;
    if ((rc = lazy_alloc(E, &E->nnls_ginv, (int64_t)KP * KP + 2))) return rc;
    { const int64_t need = E->mp > E->np ? E->mp : E->np;
      if ((rc = lazy_alloc(E, &E->nnls_todo, need))) return rc; }
    int* inv_bad = reinterpret_cast<int*>(E->nnls_ginv + (int64_t)KP * KP);
    if constexpr (KP >= 64) {                           // the blocked f64-MFMA Gauss-Jordan of the AO-ADMM `prepare` (153 -> ~20 us at k = 64)
        if ((rc = nmfx_launch_inverse64(E, G, (double)diag_add, E->nnls_ginv, inv_bad))) return rc;
    } else {
        const size_t shm = (size_t)(KP * (KP + 1) + KP) * sizeof(double);
        auto kern = nnls_inverse_kernel<KP>;
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm))) return rc;
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), shm, E->stream, G, diag_add, E->k, E->nnls_ginv, inv_bad, E->state);
        NMFX_HIP(hipGetLastError());
    }
    {
        constexpr int NW = KP <= 64 ? 6 : 2, MC = KP <= 64 ? 28 : 36;
        const size_t shm = (size_t)(KP * KP + NW * (MC * (MC + 1) + MC + KP)) * sizeof(double);
        auto kern = nnls_cinv_kernel<KP>;
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm))) return rc;
        const int64_t blocks_needed = (nprob + NW - 1) / NW;
        const int64_t resident = (int64_t)E->ncu * (KP <= 64 ? 2 : 1);                     // blocks the LDS lets a CU hold
        const unsigned grid = (unsigned)(blocks_needed < resident ? blocks_needed : resident);   // one resident round: waves loop over the problems
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), shm, E->stream, E->nnls_ginv, inv_bad, R, X, sj, sc, nprob, E->k,
                           E->state, E->nnls_todo);
        NMFX_HIP(hipGetLastError());
    }
    if constexpr (KP == 128) {                          // second pass: larger workspace, one wave per block, the first pass's leftovers
        constexpr int MC2 = 56;
        const size_t shm = (size_t)(KP * KP + (MC2 * (MC2 + 1) + MC2 + KP)) * sizeof(double);
        auto kern = nnls_cinv_kernel<KP, true>;
        if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm))) return rc;
        const unsigned grid = (unsigned)(nprob < 512 ? nprob : 512);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), shm, E->stream, E->nnls_ginv, inv_bad, R, X, sj, sc, nprob, E->k,
                           E->state, E->nnls_todo);
        NMFX_HIP(hipGetLastError());
    }
    *todo = E->nnls_todo; *bad = inv_bad;
    return NMFX_OK;
}

static int nnls(nmfx_engine* E, const float* G, float diag_add, const float* R, float* X, int64_t sj, int64_t sc,
                int64_t nprob) {
    ProfScope ps(E, "nnls");
    static const bool lds_only = getenv("NMFX_NNLS_LDS") && atoi(getenv("NMFX_NNLS_LDS")) != 0;
    // NMFX_NNLS_CINV=0: elimination kernels only (the round-1 path; A/B runs and tests/test_gpu_knobs.py)
    static const bool cinv = !(getenv("NMFX_NNLS_CINV") && atoi(getenv("NMFX_NNLS_CINV")) == 0);
    const int* todo = nullptr; const int* bad = nullptr;
    int rc;
    if (cinv) {
        switch (E->kp) {
            case 16: rc = launch_nnls_cinv<16>(E, G, diag_add, R, X, sj, sc, nprob, &todo, &bad); break;
            case 32: rc = launch_nnls_cinv<32>(E, G, diag_add, R, X, sj, sc, nprob, &todo, &bad); break;
            case 64: rc = launch_nnls_cinv<64>(E, G, diag_add, R, X, sj, sc, nprob, &todo, &bad); break;
            default: rc = launch_nnls_cinv<128>(E, G, diag_add, R, X, sj, sc, nprob, &todo, &bad); break;
        }
        if (rc) return rc;
    }
    switch (E->kp) {
        case 16: return lds_only ? launch_nnls<16>(E, G, diag_add, R, X, sj, sc, nprob, todo, bad)
                                 : launch_nnls_reg<16>(E, G, diag_add, R, X, sj, sc, nprob, todo, bad);
        case 32: return lds_only ? launch_nnls<32>(E, G, diag_add, R, X, sj, sc, nprob, todo, bad)
                                 : launch_nnls_reg<32>(E, G, diag_add, R, X, sj, sc, nprob, todo, bad);
        case 64: return lds_only ? launch_nnls<64>(E, G, diag_add, R, X, sj, sc, nprob, todo, bad)
                                 : launch_nnls_reg<64>(E, G, diag_add, R, X, sj, sc, nprob, todo, bad);
        default:
            if (lds_only) return launch_nnls<128>(E, G, diag_add, R, X, sj, sc, nprob, todo, bad);
            { static const bool wide = getenv("NMFX_NNLS128_OCC") && atoi(getenv("NMFX_NNLS128_OCC")) == 1;
              hipLaunchKernelGGL(wide ? nnls_bpp_reg128_wide_kernel : nnls_bpp_reg128_kernel, dim3((unsigned)nprob), dim3(128), 0,
                                 E->stream, G, diag_add, R, X, sj, sc, nprob, E->k, E->state, todo, bad); }
            NMFX_HIP(hipGetLastError());
            return NMFX_OK;
    }
}

// One outer iteration (anls.py:112-126) in the pieces the row-sharded form needs:
//   objective of the current pair -> [all-reduce] -> stop rule, W rows (rank-local), products
//   for H -> [all-reduce] -> H columns (replicated) and the objective partials of the new pair.
static bool anls_bf16(const nmfx_engine* E) { return E->precision == 1 && nmfx_bf16_supported(E); }

// objective partials of (W, H) (anls.py:118)
static int anls_objective(nmfx_engine* E) {
    int rc;
    // distance_type = 'kl' (anls.py:108,118 + utils.py:21-26): the KL objective of the least-squares iterates,
    // by the exact-f32 objective pass in either arithmetic mode
    E->anls_a_ready = false;
    if (E->anls_dist == NMFX_KL) return nmfx_launch_wphase(E, E->W[0], false, true, true);
    if (!anls_bf16(E)) return nmfx_launch_wphase(E, E->W[0], false, true);
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if ((rc = nmfx_bf16_images_w(E, E->W[0], 1))) return rc;
    if ((rc = nmfx_bf16_images_h(E, false))) return rc;
    // The objective pass over V is the W-side product of the NEXT iteration as well: V H^T with the residual objective of
    // (W, H) in one launch (the MUR W phase), instead of an objective-only pass now and a product pass over the same V
    // then.  (NMFX_ANLS_FUSE_OBJ=0: the two separate passes.)
    static const bool fuse = !(getenv("NMFX_ANLS_FUSE_OBJ") && atoi(getenv("NMFX_ANLS_FUSE_OBJ")) == 0);
    if (!fuse) return nmfx_bf16_objective(E, 1, "objective");
    if ((rc = nmfx_bf16_vht(E, true, 1, "wphase", false, 4))) return rc;
    E->anls_a_ready = true;
    return NMFX_OK;
}

static int anls_w_and_products(nmfx_engine* E, double lam_w, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    float* W = E->W[0];
    const int64_t kk = (int64_t)E->kp * E->kp;
    if ((rc = nmfx_finish_b(E, min_iter, tol1, tol2, j))) return rc;      // obj[j] + stop rule
    if (anls_bf16(E)) {     // the same steps with the products on the split-bf16 kernels (kp = 64 / 128)
        const bool byprod = E->kp == 64;                                   // Gram matrices as by-products
        if ((rc = nmfx_bf16_prepare(E))) return rc;
        if (!E->anls_a_ready) {                                            // (else: A_part and the H H^T slabs come from the objective pass)
            if ((rc = nmfx_bf16_images_h(E, false))) return rc;
            if ((rc = nmfx_bf16_vht(E, false, 0, "wphase_noobj"))) return rc;
        }
        E->anls_a_ready = false;
        if (!byprod && (rc = nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
        if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, byprod ? nmfx_bf16_hht_slabs(E) : E->gsplit, kk, E->HHt))) return rc;
        if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->bf_wsplit, E->mp * E->kp, E->Asum))) return rc;
        if ((rc = nnls(E, E->HHt, (float)(2.0 * lam_w), E->Asum, W, 1, E->kp, E->m))) return rc;
        if ((rc = nmfx_bf16_images_w(E, W, 0))) return rc;
        if ((rc = nmfx_bf16_vtw(E, false, "hphase"))) return rc;
        if (byprod) return nmfx_bf16_pack_t(E, E->G_part, nmfx_bf16_g_slabs(E), E->obj_count);
        if ((rc = nmfx_launch_gram_tn(E, W, E->mp, E->G_part, E->gsplit))) return rc;
        return nmfx_bf16_pack_t(E, E->G_part, E->gsplit, E->obj_count);
    }
    // ---- W: rows of W from G = H H^T + 2 lam_w I, r = (V H^T)[i, :]  (anls.py:18-31) ----
    if ((rc = nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_wphase(E, W, true, false))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc;
    if ((rc = nmfx_launch_sum_partials(E, E->A_part, E->wsplit, E->mp * E->kp, E->Asum))) return rc;
    if ((rc = nnls(E, E->HHt, (float)(2.0 * lam_w), E->Asum, W, 1, E->kp, E->m))) return rc;
    // ---- products for H: [W^T V | W^T W] ----
    const bool fuse_g = nmfx_hphase_can_fuse_gram(E);
    if (!fuse_g && (rc = nmfx_launch_gram_tn(E, W, E->mp, E->G_part, E->gsplit))) return rc;
    if ((rc = nmfx_launch_hphase(E, W, fuse_g))) return rc;
    return nmfx_launch_pack(E);
}

static int anls_h(nmfx_engine* E, double lam_h) {
    int rc;
    // ---- H: columns of H from G = W^T W + 2 lam_h I, r = (W^T V)[:, c]  (anls.py:34-47) ----
    if ((rc = nnls(E, E->xf32 + (int64_t)E->kp * E->np, (float)(2.0 * lam_h), E->xf32, E->H, E->np, 1, E->n))) return rc;
    // ---- objective (anls.py:118) ----
    return anls_objective(E);
}

static int anls_iteration(nmfx_engine* E, double lam_w, double lam_h, int64_t min_iter, double tol1, double tol2,
                          int64_t j) {
    int rc;
    // objective of the current pair (filled by the previous pass) and the stop rule
    if ((rc = nmfx_launch_obj_reduce(E))) return rc;
    if ((rc = anls_w_and_products(E, lam_w, min_iter, tol1, tol2, j))) return rc;
    return anls_h(E, lam_h);
}

static int anls_ready(nmfx_engine* E, int64_t j, double lam) {
    if (!E) return NMFX_E_ARG;
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    if (j < 0 || lam < 0) { E->err = "bad range or lambda"; return NMFX_E_ARG; }
    { int rc_ = nmfx_enter_family(E, 4); if (rc_) return rc_; }      // (beyond 128 components the phases are composed from the generic kernels: r4)
    NMFX_HIP(hipSetDevice(E->device));
    if (!E->Asum) {
        NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->Asum), (size_t)E->mp * E->kp * sizeof(float)));
        NMFX_HIP(hipMemsetAsync(E->Asum, 0, (size_t)E->mp * E->kp * sizeof(float), E->stream));
    }
    E->wsel = 0;
    E->w_in_place = true;
    return nmfx_ensure_obj_capacity(E, j + 3);
}

// ---- row-sharded form: objective partial -> [all-reduce f64] -> phase_w -> [all-reduce f32]
// -> phase_h ----
extern "C" int nmfx_anls_phase_objective(nmfx_handle_t E, int64_t j) {
    if (E) { E->himg_both = false; E->kl_h_iter = -2; }
    int rc = anls_ready(E, j, 0.0); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_anls_phase(E, 0, 0.0, 0, 0.0, 0.0, j);
    if (j == 0 && (rc = anls_objective(E))) return rc;   // obj[0] partials
    return nmfx_launch_obj_reduce(E);
}

extern "C" int nmfx_anls_phase_w(nmfx_handle_t E, double lambda_w, int64_t min_iter, double tol1, double tol2,
                                 int64_t j) {
    if (E) { E->himg_both = false; E->kl_h_iter = -2; }
    int rc = anls_ready(E, j, lambda_w); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_anls_phase(E, 1, lambda_w, min_iter, tol1, tol2, j);
    return anls_w_and_products(E, lambda_w, min_iter, tol1, tol2, j);
}

extern "C" int nmfx_anls_phase_h(nmfx_handle_t E, double lambda_h, int64_t j) {
    if (E) { E->himg_both = false; E->kl_h_iter = -2; }
    int rc = anls_ready(E, j, lambda_h); if (rc) return rc;
    if (E->kp > 128) return nmfx_generic_anls_phase(E, 2, lambda_h, 0, 0.0, 0.0, j);
    return anls_h(E, lambda_h);
}

extern "C" int nmfx_anls_set_distance(nmfx_handle_t E, int distance) {
    if (!E || (distance != NMFX_EU && distance != NMFX_KL)) { if (E) E->err = "Unknown distance type."; return NMFX_E_ARG; }
    E->anls_dist = distance;
    E->anls_a_ready = false;
    return NMFX_OK;
}

extern "C" int nmfx_anls_run(nmfx_handle_t E, double lambda_w, double lambda_h, int64_t min_iter, double tol1,
                             double tol2, int64_t first, int64_t count) {
    if (E) { E->himg_both = false; E->kl_h_iter = -2; }
    if (!E) return NMFX_E_ARG;
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    if (first < 0 || count < 0 || lambda_w < 0 || lambda_h < 0) { E->err = "bad range or lambda"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if ((rc = nmfx_enter_family(E, 4))) return rc;
    if (E->kp > 128) {         // one workgroup per right-hand side, systems in a global f64 work area (kernels_generic.hip)
        if ((rc = nmfx_ensure_obj_capacity(E, first + count + 2))) return rc;
        E->wsel = 0;
        E->w_in_place = true;
        E->anls_a_ready = false;
        return nmfx_generic_anls_run(E, lambda_w, lambda_h, min_iter, tol1, tol2, first, count);
    }
    if (!E->Asum) {
        NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->Asum), (size_t)E->mp * E->kp * sizeof(float)));
        NMFX_HIP(hipMemsetAsync(E->Asum, 0, (size_t)E->mp * E->kp * sizeof(float), E->stream));
    }
    if ((rc = nmfx_ensure_obj_capacity(E, first + count + 2))) return rc;
    E->wsel = 0;
    E->w_in_place = true;
    if (first == 0 && count > 0 && (rc = anls_objective(E))) return rc;   // obj[0]
    for (int64_t j = first; j < first + count; ++j)
        if ((rc = anls_iteration(E, lambda_w, lambda_h, min_iter, tol1, tol2, j))) return rc;
    return NMFX_OK;
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_anls() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(nnls_bpp_reg128_kernel)) == hipSuccess ? 0 : -1; }
