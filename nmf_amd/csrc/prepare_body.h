// Blocked Gauss-Jordan inverse of a shifted Gram matrix on v_mfma_f64_16x16x4_f64 (AO-ADMM / ADMM `prepare`, the f64 inverse of the
// ANLS complement solver, the diagonal blocks of the inversion beyond k = 128): the body is shared by the stand-alone kernel
// (kernels_aoadmm.hip) and by the SIDE JOB of the split-bf16 product kernel (kernels_bf16.hip, r4: the one-workgroup inversion runs as
// one more block of the V-sized product's grid instead of as a launch of its own with 255 CUs idle).
#pragma once
#include "nmfx_internal.h"
#include "kernels_small.h"

// The same inverse for KP = 64 / 128 as a BLOCKED Gauss-Jordan on v_mfma_f64_16x16x4_f64: the scalar
// kernel above pays one workgroup barrier per pivot (128 of them at ~1 us), this one pays one per
// 16 x 16 block step.  NB = KP / 16 waves; wave w keeps block row w (NB tiles of 16 x 16 f64) in
// registers in the MFMA C/D layout (lane: column c = lane & 15, rows q + 4 r, q = lane >> 4).
// Block step kb:
//   (a) wave kb inverts its diagonal tile D in place (16 scalar pivots inside ONE wave, row broadcast by
//       ds_bpermute, column broadcast by DPP row_share, no barrier; these are the pivots of the unblocked
//       elimination, so "not positive definite" is detected on the same condition),
//       and publishes D^-1 and its OLD row panel T[kb][j],
//       -- one workgroup barrier --
//   (b) wave kb turns its own row panel into D^-1 T[kb][j] (MFMA),
//   (c) at the same time every other wave i: L = -T[i][kb] D^-1 (its new T[i][kb]), T[i][j] += L T_old[kb][j]  (MFMA) --
//       the reassociated form of T[i][j] -= T[i][kb] (D^-1 T[kb][j]), so that (b) is no longer between (a) and (c) on the
//       critical path of a step (r2: 52 -> 44 us at k = 128).
// Wave kb+1 goes on to (a) of the next step as soon as its own (c) is done; the published
// panels are double buffered, so the one barrier per step is enough.
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// value of lane P of the caller's row of 16 lanes (DPP row_share: a VALU move, no LDS round trip)
template <int P> __device__ __forceinline__ double row_share_f64(double v) {     // ONE v_mov_b64_dpp row_newbcast:P (a single wave issues
    long x = __double_as_longlong(v);                                             // a VALU instruction every ~5 cycles: the in-wave
    x = __builtin_amdgcn_update_dpp(x, x, 0x150 + P, 0xf, 0xf, true);             // pivots are bound by their instruction COUNT)
    return __longlong_as_double(x);
}
// the 16 lanes with lane >> 4 == QSRC, copied to all four rows of lanes: two gfx950 lane swaps per dword (v_permlane32_swap
// leaves lanes 0-31 of its first operand in both halves of it and lanes 32-63 in both halves of the second; v_permlane16_swap
// does the same with the even / odd rows of 16) -- VALU moves instead of the LDS round trip of a ds_bpermute
// (probe: tools/lab/permlane_probe.hip)
template <int QSRC> __device__ __forceinline__ int bcast_row_i32(int v) {
    auto h = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    const unsigned z = (QSRC >= 2) ? h[1] : h[0];
    auto g = __builtin_amdgcn_permlane16_swap(z, z, false, false);
    return (int)((QSRC & 1) ? g[1] : g[0]);
}
template <int QSRC> __device__ __forceinline__ double bcast_row_f64(double v) {
    return __hiloint2double(bcast_row_i32<QSRC>(__double2hiint(v)), bcast_row_i32<QSRC>(__double2loint(v)));
}
// One scalar pivot of the in-wave 16 x 16 Gauss-Jordan (tile in the C/D layout: a[r] = D[q + 4r][c]).
// Row P lives in the lanes with q == P & 3 (bcast_row brings it to every row of lanes), column P
// in lane P of every row of lanes (row_share).  Same adjusted-pivot update as the scalar kernel.  A pivot that is not
// positive (scipy cholesky would raise LinAlgError) is only NOTED in `ok` (no branch on the chain of 16 dependent pivots;
// what follows it is garbage nobody uses).
template <int P> __device__ __forceinline__ void gj16_pivot(f64x4& a, int c, int q, bool& ok) {
    const double rowp = bcast_row_f64<(P & 3)>(a[P >> 2]);              // D[P][c]
    const double piv = row_share_f64<P>(rowp);                          // D[P][P]
    ok = ok && (piv > 0.0);
    double inv = __builtin_amdgcn_rcp(piv);                             // + two Newton steps: full f64 accuracy
    inv = fma(fma(-piv, inv, 1.0), inv, inv);
    inv = fma(fma(-piv, inv, 1.0), inv, inv);
    const double rv = (c == P) ? 1.0 + inv : rowp * inv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double colv = row_share_f64<P>(a[r]);                     // D[q + 4r][P]
        const double cv = (q + 4 * r == P) ? piv - 1.0 : colv;
        a[r] = fma(-cv, rv, a[r]);
    }
}

#if defined(NMFX_EXP_STAMPS) && defined(NMFX_PREP_STAMPS_HERE)   // experiment (tools/lab/prep_stamps.py): the timeline of the last blocked Gauss-Jordan launch, 10 ns ticks (kernels_aoadmm.hip only)
__device__ unsigned long long nmfx_dbg_prep[9][40];            // [wave (8 = helper)][0 start, 1 loaded, 2 first tile inverted, 3 + 3 kb: step barrier | mid-step barrier | step done, 30 end]
extern "C" int nmfx_debug_prep_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nmfx_dbg_prep), sizeof(nmfx_dbg_prep)) == hipSuccess ? 0 : -1;
}
#define PREP_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); if (lane == 0) nmfx_dbg_prep[w][slot] = __builtin_amdgcn_s_memrealtime(); \
                              __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PREP_STAMP(slot) do { } while (0)
#endif
// HELPER: a ninth (fifth) wave that inverts the diagonal tiles one step ahead (the stand-alone kernel: KP * 4 + 64 threads).  Without
// it (the side job: the product kernel's 512 threads) wave kb + 1 inverts its own diagonal tile behind the mid-step barrier, before
// its other tiles -- the form of round 2, a few us slower and, beside a V-sized product, hidden anyway.  Waves beyond the ones the
// matrix needs only keep the barriers company.
// nslab: src holds that many slabs of KP * KP floats whose sum (slab order) is the Gram matrix.
// defer_notpd: a non-positive pivot is only noted in st->notpd_pending -- the launch that records the objective of the iteration
// decides (a stop rule that fires first wins, as in the reference, where the factorisation of the next iteration is never tried).
template <int KP, bool HELPER>
__device__ __forceinline__ void ao_prepare_body(
    double* __restrict__ prep_lds,
    const float* __restrict__ src, int nslab, int k, float* __restrict__ Minv, DevState* __restrict__ st,
    double fixed_rho, bool defer_notpd,
    double* __restrict__ out64 = nullptr, int* __restrict__ soft_bad = nullptr,
    const double* __restrict__ src64 = nullptr, int64_t ld64 = 0)
{
    // src64 != nullptr (with out64; a diagonal block of the blocked Gauss-Jordan inversion beyond k = 128, kernels_generic.hip): the
    // matrix is the KP x KP f64 block at src64 (row stride ld64), nothing is added to its diagonal, its f64 inverse goes to out64
    // [KP][KP]; a pivot <= 0 is the run's "not positive definite" unless soft_bad is given.
    // out64 != nullptr (ANLS, nnls_cinv_kernel): the f64 inverse of src + fixed_rho I goes to out64 [KP][KP] instead of the
    // f32 one to Minv, padded variables (index >= k) get a unit diagonal, the solver state is left alone, and
    // "not positive definite" or "too ill-conditioned for an explicit inverse" (max diag(inverse) x mean diag(matrix)
    // > 1e9) are reported in *soft_bad instead of stopping the run
    //
    // Waves 0 .. NB-1 own one block row of 16 rows each (NB tiles of f64x4 per lane); wave NB is the HELPER that inverts the
    // diagonal tiles one step ahead (end of r2).  Measured timeline of the form without it (tools/lab/prep_stamps.py, k = 128):
    // 4.45 us per step = 2.4 us for the 16 dependent pivots of the diagonal tile, with seven waves waiting, + 2.0 us for the
    // 36 f64 MFMAs per wave of the update (two waves per SIMD share the pipe: that part is bound by the matrix cores).  The
    // pivots are VALU work, so they can run BESIDE the update: in step kb every row wave first brings its tile of block column
    // kb + 1 up to date; behind a barrier the helper inverts T[kb+1][kb+1] while the row waves update their other tiles.
    constexpr int NB = KP / 16, LDP = KP + 2, LDD = 17;
    double* dinv = prep_lds;                            // [2][16][LDD]   D^-1 of the step (A and B operand source)
    double* rown = dinv + 2 * 16 * LDD;                 // [2][16][LDP]   OLD row panel of the step (through the step before)
    double* diag = rown + 2 * 16 * LDP;                 // [16][LDD]      next diagonal tile, row wave -> helper ([16][LDP] reserved)
    double* colp = diag + 16 * LDP;                     // [NB][16][LDD]  own column tile of every wave
    double* misc = colp + NB * 16 * LDD;                // [0] rho, [1] bad, [2] mean diagonal
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, q = lane >> 4;
    const bool helper = HELPER && w == NB;
    const bool idle = w > NB || (!HELPER && w == NB);       // (waves the matrix does not need)
    PREP_STAMP(0);

    if (w == 0) {                                       // rho = trace(G) / k (ao_admm.py:54), or the fixed one (ADMM)
        double tr = 0.0;
        if (src64) { for (int i = lane; i < k; i += 64) tr += src64[(int64_t)i * ld64 + i]; }
        else for (int i = lane; i < k; i += 64) {
            float d = 0.f;
            for (int p = 0; p < nslab; ++p) d += src[(int64_t)p * KP * KP + (int64_t)i * KP + i];
            tr += (double)d;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tr += __shfl_down(tr, off, 64);
        if (lane == 0) {
            const double rho = (fixed_rho >= 0.0) ? fixed_rho : tr / (double)k;
            misc[0] = rho; misc[1] = 0.0; misc[2] = tr / (double)k + rho;
            if (!out64) { st->rho = rho; st->inner_stop = 0; st->inner_count = 0; }
        }
    }
    f64x4 t[NB];
    if (!helper && !idle) {
        if (src64 || nslab == 1) {
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    t[jb][r] = src64 ? src64[(int64_t)(16 * w + q + 4 * r) * ld64 + 16 * jb + c] : (double)src[(int64_t)(16 * w + q + 4 * r) * KP + 16 * jb + c];
        } else {                                        // the Gram slabs summed on the way in, f32 and in slab order like the `sums` / `pack` launches
            float a[NB][4];
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[jb][r] = 0.f;
            int p = 0;
            for (; p + 4 <= nslab; p += 4) {            // four slabs' loads in flight together (one CU streams all of them), added in slab order
                float v[4][NB][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[u][jb][r] = src[(int64_t)(p + u) * KP * KP + (int64_t)(16 * w + q + 4 * r) * KP + 16 * jb + c];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[jb][r] += v[u][jb][r];
            }
            for (; p < nslab; ++p) {
                const float* sp = src + (int64_t)p * KP * KP;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[jb][r] += sp[(int64_t)(16 * w + q + 4 * r) * KP + 16 * jb + c];
            }
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) t[jb][r] = (double)a[jb][r];
        }
    }
    __syncthreads();
    PREP_STAMP(1);
    const double rho = misc[0];
    if (!helper && !idle) {
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
            if (jb == w) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (q + 4 * r == c) t[jb][r] += (out64 && 16 * w + c >= k) ? 1.0 : rho;
            }
    }
    // the helper's job: invert the tile in `diag` with the adjusted-pivot update of the scalar kernel, leave it in dinv[buf]
    auto invert_diag = [&](int buf) {
        f64x4 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = diag[(q + 4 * r) * LDD + c];
        bool ok = true;
        gj16_pivot<0>(a, c, q, ok); gj16_pivot<1>(a, c, q, ok); gj16_pivot<2>(a, c, q, ok); gj16_pivot<3>(a, c, q, ok);
        gj16_pivot<4>(a, c, q, ok); gj16_pivot<5>(a, c, q, ok); gj16_pivot<6>(a, c, q, ok); gj16_pivot<7>(a, c, q, ok);
        gj16_pivot<8>(a, c, q, ok); gj16_pivot<9>(a, c, q, ok); gj16_pivot<10>(a, c, q, ok); gj16_pivot<11>(a, c, q, ok);
        gj16_pivot<12>(a, c, q, ok); gj16_pivot<13>(a, c, q, ok); gj16_pivot<14>(a, c, q, ok); gj16_pivot<15>(a, c, q, ok);
        if (!ok) { if (lane == 0) misc[1] = 1.0; }
        else {
            double* dv = dinv + buf * 16 * LDD;
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[(q + 4 * r) * LDD + c] = a[r];
        }
    };
    // step 0 has nothing to hide behind: wave 0 publishes its row panel and its diagonal tile, the helper inverts it
    if (w == 0) {
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) rown[(q + 4 * r) * LDP + 16 * jb + c] = t[jb][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) diag[(q + 4 * r) * LDD + c] = t[0][r];
    }
    __syncthreads();
    if (HELPER ? helper : w == 0) invert_diag(0);
    PREP_STAMP(2);

#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const double* dv = dinv + (kb & 1) * 16 * LDD;
        const double* rn = rown + (kb & 1) * 16 * LDP;
        __syncthreads();                                // D^-1 and the OLD row panel of step kb are in place
        PREP_STAMP(3 + 3 * kb);
        if (misc[1] != 0.0) {
            if (tid == 0) { if (soft_bad) *soft_bad = 1; else if (defer_notpd) st->notpd_pending = 1; else { st->notpd = 1; st->flag = 3; } }
            return;
        }
        double af[4];
        if (helper || idle) {
            // nothing before the mid-step barrier
        } else if (w == kb) {
            // (a) the diagonal tile becomes D^-1; (b) own row panel: new = D^-1 * old (old tiles from the published panel) --
            //     off the critical path of the step: the other waves use the OLD panel, see (c)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[kb][r] = dv[(q + 4 * r) * LDD + c];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) af[s4] = dv[c * LDD + 4 * s4 + q];
        } else {
            // (c) L = -T[w][kb] D^-1 (the new T[w][kb]), then T[w][j] += L T_old[kb][j]: A operands through this wave's own LDS tile
            double* cp = colp + w * 16 * LDD;
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[(q + 4 * r) * LDD + c] = t[kb][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) af[s4] = -cp[c * LDD + 4 * s4 + q];
            f64x4 l = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) l = MFMA_F64(af[s4], dv[(4 * s4 + q) * LDD + c], l);
            t[kb] = l;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[(q + 4 * r) * LDD + c] = l[r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) af[s4] = cp[c * LDD + 4 * s4 + q];
        }
        // the tile of block column kb + 1 first: wave kb + 1 hands its (now current) diagonal tile to the helper
        auto update_tile = [&](int jb) {
            if (w == kb) {
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = MFMA_F64(af[s4], rn[(4 * s4 + q) * LDP + 16 * jb + c], acc);
                t[jb] = acc;
            } else {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t[jb] = MFMA_F64(af[s4], rn[(4 * s4 + q) * LDP + 16 * jb + c], t[jb]);
            }
        };
        if (kb + 1 < NB) {
            if (!helper && !idle) update_tile(kb + 1);
            if (w == kb + 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) diag[(q + 4 * r) * LDD + c] = t[kb + 1][r];
            }
            __syncthreads();                            // mid-step: the next diagonal tile is with the helper
            PREP_STAMP(4 + 3 * kb);
        }
        if (helper) {
            if (kb + 1 < NB) invert_diag((kb + 1) & 1);
        } else if (!idle) {
            if (!HELPER && w == kb + 1) invert_diag((kb + 1) & 1);
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                if (jb == kb || jb == kb + 1) continue;
                update_tile(jb);
            }
            if (w == kb + 1) {                          // the OLD row panel of the next step (its diagonal tile is not read by anybody)
                double* rnn = rown + ((kb + 1) & 1) * 16 * LDP;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rnn[(q + 4 * r) * LDP + 16 * jb + c] = t[jb][r];
            }
        }
        PREP_STAMP(5 + 3 * kb);
    }
    if (out64) {
        double dmax = 0.0;                              // largest diagonal entry of the inverse
        if (!helper && !idle) {
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
                if (jb == w) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (q + 4 * r == c && 16 * w + c < k) dmax = fmax(dmax, t[jb][r]);
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
        __syncthreads();                                // (dinv is free: every wave is behind the last step)
        if (lane == 0 && !helper && !idle) dinv[w] = dmax;
        __syncthreads();
        if (tid == 0 && soft_bad) {
            double mx = 0.0;
            for (int i = 0; i < NB; ++i) mx = fmax(mx, dinv[i]);
            *soft_bad = (mx * misc[2] > 1e9 || !(mx == mx)) ? 1 : 0;
        }
        if (helper || idle) return;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) out64[(int64_t)(16 * w + q + 4 * r) * KP + 16 * jb + c] = t[jb][r];
        return;
    }
    if (defer_notpd && tid == 0) st->notpd_pending = 0;
    if (helper || idle) return;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Minv[(int64_t)(16 * w + q + 4 * r) * KP + 16 * jb + c] = (float)t[jb][r];
    PREP_STAMP(30);
}

