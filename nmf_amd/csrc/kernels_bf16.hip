// Split-bf16 ("2 x bf16") form of the two V-sized products of MUR-Euclidean, k = 64.
//
// An f32 number x is written x = hi + lo + eps with hi = bf16(x), lo = bf16(x - hi),
// |eps| <= 2^-17 |x|.  A product of two such numbers is evaluated as the four bf16 MFMA
// terms hi*hi + hi*lo + lo*hi + lo*lo with f32 accumulation (v_mfma_f32_16x16x32_bf16,
// 16x the rate of the f32-input MFMA), i.e. every operand carries 16 significant bits
// and the accumulation is the same f32 chain as before.  On dot products of length
// 8192 / 16384 the result differs from the exact-f32 kernels by ~1e-7 relative (the
// parity tests run both).  With this form both products leave the MFMA roofline and
// become HBM bound (V is read once per phase).
//
// One kernel serves both phases because every contraction runs along the contiguous
// dimension:
//     W phase:  A   = V   H^T      X = V   [m][n],  Y = H   [64][n]   (+ fused residual)
//     H phase:  B^T = V^T W        X = V^T [n][m],  Y = W^T [64][m]
// so the engine keeps tile-major copies of V and V^T in HBM (built once after the upload)
// and the small update kernels emit the bf16 hi/lo images of the factors they have just
// produced (W: [m][64] for the residual and [64][m] for the H phase; H: [64][n]).
#include "nmfx_internal.h"
#include "prepare_body.h"
#include "kernels_small.h"
#include <cstdlib>
#include <type_traits>

// Terms of a split product x y with x = xh + xl + ex, y = yh + yl + ey (|xl| <= 2^-8 |x|, |ex| <= 2^-16 |x|):
//   4: xh yh + xh yl + xl yh + xl yl;
//   3: without xl yl.  The dropped term is <= 2^-16 |x y|, the size of the representation errors ex y + x ey
//      that the four-term form has anyway (worst case 3 x 2^-16 instead of 2 x 2^-16, random signs), and it is
//      25 % of the MFMAs: the W phase of config 2 takes 121-125 us instead of 139-141 on the same box.
// The multiplicative updates (MUR, both divergences) take 3: an update is a ratio of two such sums and the
// iteration is self-correcting.  AO-ADMM takes 3 as well: its systems are G + rho I with rho = trace(G) / k,
// condition number <= k + 1.  ANLS and ADMM keep 4: their systems use the unshifted Gram matrix or the
// caller's fixed rho, and the perturbation is amplified by the conditioning (ANLS 300 x 220, k = 40: objective
// history 8e-4 off the oracle with 3 terms, < 5e-4 with 4; ADMM 64 x 64, k = 64, rho = 1: W H 1.02e-4 off).
// The residual product Z Y of the Euclidean objective always takes 3: the objective is only recorded and
// compared, nothing is computed from it (3 vs 4 terms moves it by ~1e-8 relative).
// NMFX_BF16_TERMS=4 in the environment forces 4 for every product that is fed back.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// t_i = x_i * log2(q_i), i = 0..3, with v_mul_legacy_f32 (0 * anything, inf and nan included, = 0).  ONE asm block: the four
// logarithms first, then the products -- a transcendental's result must not be consumed by the very next VALU instruction
// on this part, and the hazard recogniser does not look inside inline asm (a lone `v_mul_legacy_f32` right behind a compiler
// generated `v_log_f32` read a stale register).
__device__ __forceinline__ void xlog2_legacy4(const float4& x, float q0, float q1, float q2, float q3,
                                              float& t0, float& t1, float& t2, float& t3) {
    asm("v_log_f32 %0, %4\n\tv_log_f32 %1, %5\n\tv_log_f32 %2, %6\n\tv_log_f32 %3, %7\n\t"
        "v_mul_legacy_f32 %0, %8, %0\n\tv_mul_legacy_f32 %1, %9, %1\n\tv_mul_legacy_f32 %2, %10, %2\n\tv_mul_legacy_f32 %3, %11, %3"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
}
union Frag8 { uint4 u; bf16x8 v; };
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a).v, (b).v, (c), 0, 0, 0)

__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// LDS-DMA runs: N consecutive 1 KiB pieces, 16 bytes per lane each, from (wave-uniform base
// + per-lane 32-bit byte offset) to LDS (wave-uniform dst + piece * 1 KiB + lane * 16).  One
// asm statement per run: m0 is saved once, bumped between the loads and restored, the lane
// offsets are loop invariant and the base advances on the scalar unit, so a run costs no
// vector instruction (cdna_hip_programming.md 5.7: M0 written in the statement that uses it).
__device__ __forceinline__ void dma_run4(unsigned long long base, unsigned dst, unsigned o0, unsigned o1,
                                         unsigned o2, unsigned o3) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(dst), "v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory", "scc");
}
__device__ __forceinline__ void dma_run4_nt(unsigned long long base, unsigned dst, unsigned o0, unsigned o1,
                                            unsigned o2, unsigned o3) {       // streaming (non-temporal) policy
#ifdef NMFX_EXP_TEMPORAL      // experiment: the tile stream with the default cache policy (does a small V stay in the Infinity Cache?)
    dma_run4(base, dst, o0, o1, o2, o3); return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(dst), "v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory", "scc");
}
__device__ __forceinline__ void dma_run2(unsigned long long base, unsigned dst, unsigned o0, unsigned o1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(dst), "v"(o0), "v"(o1) : "memory", "scc");
}
__device__ __forceinline__ void dma_run2_nt(unsigned long long base, unsigned dst, unsigned o0, unsigned o1) {
#ifdef NMFX_EXP_TEMPORAL
    dma_run2(base, dst, o0, o1); return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(dst), "v"(o0), "v"(o1) : "memory", "scc");
}
template <int N> __device__ __forceinline__ void dma_wait_le() {      // at most N DMAs still in flight
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void dma_wait_le_rt(int n) {                 // the same with a wave-uniform count (a multiple of 4, at most 52)
    switch (n >> 2) {
        case 0: dma_wait_le<0>(); break;   case 1: dma_wait_le<4>(); break;   case 2: dma_wait_le<8>(); break;   case 3: dma_wait_le<12>(); break;
        case 4: dma_wait_le<16>(); break;  case 5: dma_wait_le<20>(); break;  case 6: dma_wait_le<24>(); break;  case 7: dma_wait_le<28>(); break;
        case 8: dma_wait_le<32>(); break;  case 9: dma_wait_le<36>(); break;  case 10: dma_wait_le<40>(); break; case 11: dma_wait_le<44>(); break;
        case 12: dma_wait_le<48>(); break;
        default: dma_wait_le<52>(); break;
    }
}
__device__ __forceinline__ void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void pinu(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// two floats -> packed bf16 hi pair and lo pair (round to nearest even both times)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(a - ah), "v"(b - bh));    // (r5: the two subtractions as ONE v_pk_add_f32 -- 6 % fewer VALU
}                                                                                  //  instructions in the KL loops -- measured no faster, k = 128 1 % slower: not kept)
#ifdef NMFX_EXP_FP8D           // experiment (r5): the cross terms of the RESIDUAL product (k = 128 Euclidean form) on the block-scaled fp8 matrix pipe, operands
                               // converted in registers from the bf16 fragments (fixed scales: hi x 1, lo x 2^9).  Measured: config 5 W phase 2740 -> 2719 us,
                               // config 3 180.2 -> 178.8 (-0.8 %): the 64 conversions per group cost the issue port what the 12 saved MFMAs give back;
                               // recorded objective +1e-6 .. +1.4e-5 off the f64 one.  The fp8 pipe pays only with operands that arrive converted (LAB_NOTES R5)
typedef int fp8_i32x8 __attribute__((ext_vector_type(8)));
typedef short fp8_s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fp8_bf16x2 __attribute__((ext_vector_type(2)));
// eight bf16 (one 32x32x16 k-step's fragment) -> eight e4m3 bytes = registers 2 j, 2 j + 1 of a 32x32x64 operand; stored value = x / scale
__device__ __forceinline__ void frag_to_fp8(const Frag8& f, float scale, int& r0, int& r1) {
    union { unsigned u; fp8_bf16x2 v; } a, b, c, d;
    a.u = f.u.x; b.u = f.u.y; c.u = f.u.z; d.u = f.u.w;
    union { fp8_s16x2 v; int i; } o0, o1;
    o0.i = 0; o1.i = 0;
    o0.v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o0.v, a.v, scale, false);
    o0.v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o0.v, b.v, scale, true);
    o1.v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o1.v, c.v, scale, false);
    o1.v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o1.v, d.v, scale, true);
    r0 = o0.i; r1 = o1.i;
}
#endif
__device__ __forceinline__ void hi8(const float4& p, const float4& q, Frag8& hi) {       // the bf16 image alone (round to nearest even)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi.u.x) : "v"(p.x), "v"(p.y));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi.u.y) : "v"(p.z), "v"(p.w));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi.u.z) : "v"(q.x), "v"(q.y));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi.u.w) : "v"(q.z), "v"(q.w));
}
__device__ __forceinline__ void split8(const float4& p, const float4& q, Frag8& hi, Frag8& lo) {
    split2(p.x, p.y, hi.u.x, lo.u.x); split2(p.z, p.w, hi.u.y, lo.u.y);
    split2(q.x, q.y, hi.u.z, lo.u.z); split2(q.z, q.w, hi.u.w, lo.u.w);
}

// ---------------------------------------------------------------------------
// A_part[sp] = X(rows of the block, columns of split sp) * Y^T, optional residual
// objective 0.5 * sum (X - Z Y)^2.  Block = 128 rows (8 waves x 16 rows, two waves per
// SIMD), 64-column groups.  LDS (all of the 160 KiB):
//   Y side, double buffered: ONE image of the Y tile per group -- Yhi and Ylo, each 64 rows
//     (factors) x 128 B of bf16; 16-byte chunk c of row r sits at position c ^ yswz(r).  The
//     A-product takes its B operand from it by rows (ds_read_b128), the residual product takes
//     its A operand from the SAME image by columns (ds_read_b64_tr_b16, the gfx950 transposing
//     read): yswz is conflict free for both (searched exhaustively over XOR-linear maps against
//     the lane groups of MI355X_MICROARCH.md, LDS table).  A second, transposed copy of the tile
//     would double the L2 -> LDS traffic, which is what bounds the W phase.
//   V side, a 4-deep ring per wave: [16][64] f32, chunk c of row r at position c ^ r;
//     this is the HBM stream, requested four groups ahead (the slot of the tile being consumed is
//     refilled as soon as every wave holds its tile in registers: 128 KiB in flight per CU; the
//     stream rate follows the bytes in flight until HBM saturates).  X is either row-major (ldx) or
//     tile-major ([128 rows][64 cols] tiles, one contiguous 32 KiB read per block and group).
// Everything is filled by LDS-DMA and retired with a COUNTED s_waitcnt vmcnt.  The DMA work
// is split by wave (4 "Y loaders", 4 "V loaders" that each fetch the tiles of two waves) so
// that the deep V prefetch is not drained by the shallow Y prefetch.  One barrier per group
// publishes Y(grp) / V(grp) and frees the buffers read one group earlier.  All LDS reads are
// (loop-invariant lane offset) + immediate and conflict free.
// ---------------------------------------------------------------------------
#pragma clang fp contract(fast)
__device__ __forceinline__ int yswz(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

typedef short s16x4 __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16: within each group of 16 lanes, lane 4q+p supplies the address of 4
// consecutive bf16 of row q; lane i receives element i of the 16-wide rows 0..3.
__device__ __forceinline__ uint2 lds_read_tr(const unsigned char* p) {
    union { s16x4 s; uint2 u; } r;
    r.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(const __attribute__((address_space(3))) void*)p);
    return r.u;
}

// KP = 64 or 128 factors.  KP = 128: Y image 2 x 16 KiB per group (double buffered: 64 KiB), V ring
// 3 deep, accumulators for 8 factor tiles, the Gram by-product is left to the Gram kernels.
//
// KL = true (MUR with the KL divergence, nmf/mur.py:24-27, 40-43): A_part = (X / (Z Y + 1e-9)) Y^T.
// The product Z Y comes FIRST (same transposed reads), the quotient is formed in registers in the
// D layout (lane = row x, columns 16 e + 4 g + r) and is used as the A operand as it stands: the
// contraction index of the A-product is PERMUTED (k-step s runs over the columns of e = 2s and
// 2s + 1), and the B operand follows with two 8-byte reads of the same Y image instead of one
// 16-byte read.  With WITH_OBJ the objective term x log(x / zy) - x + zy (nmf/utils.py:23-26) is
// accumulated from the same registers.
#ifdef NMFX_EXP_BLOCKTIME      // experiment (tools/lab/block_times.py): start / end time of every block of the last product launches
__device__ unsigned long long nmfx_dbg_times[2][2][1024];      // [with objective][start | end][block]
extern "C" int nmfx_debug_block_times(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nmfx_dbg_times), sizeof(nmfx_dbg_times)) == hipSuccess ? 0 : -1;
}
#endif
// WITH_A = false (with WITH_OBJ, Euclidean): only the residual objective -- no A-product, no Gram by-product,
// nothing written but objpart (the objective passes of ADMM and ANLS).
template <int KP, bool WITH_OBJ, bool KL, int TERMS, bool WITH_A = true>
__global__ __launch_bounds__(512) void xyt_bf16_kernel(
    const float* __restrict__ X, int64_t ldx,
    const unsigned short* __restrict__ Yhi, const unsigned short* __restrict__ Ylo, int64_t ldy,
    const unsigned short* __restrict__ Zhi, const unsigned short* __restrict__ Zlo,
    float* __restrict__ Apart, double* __restrict__ objpart, float* __restrict__ gram_part, int64_t R,
    int ngroups, const int* __restrict__ flag, int tiled, int ng)
{
    if (*flag) return;
#ifdef NMFX_EXP_BLOCKTIME
    const int dbg_b = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && dbg_b < 1024) nmfx_dbg_times[WITH_OBJ][0][dbg_b] = wall_clock64();
#endif
    constexpr int NJT = KP / 16;                      // factor tiles of the A-product
    constexpr int YT = KP * 128;                      // bytes of one Y tile (KP rows x 64 bf16)
    constexpr int YBUF = 2 * YT;                      // Yhi tile, Ylo tile
    constexpr int VOFF = 2 * YBUF;                    // start of the V rings
    constexpr int VRING = (KP == 64) ? 4 : 3;         // V ring depth (LDS: 2*YBUF + 8*VRING*4 KiB = 160 KiB)
    static_assert(WITH_A || (WITH_OBJ && !KL), "without the A-product the launch must at least compute the objective");
    constexpr bool WITH_GRAM = (KP == 64) && !KL && WITH_A;
    constexpr bool WITH_D = WITH_OBJ || KL;           // the product Z Y is formed
    constexpr int YPW = 2 * (KP / 8) / 4;             // Y pieces (8 rows x 128 B) per loader wave and group
    constexpr int NA = WITH_A ? 2 * (NJT / 4) : 0;    // pipeline stages of the A-product: (k-step, half of the tiles)
    constexpr int ND = WITH_D ? KP / 32 : 0;          // stages of the product Z Y: k-steps over the factors
    constexpr int NS = NA + ND;                       // order: A.. then D.. (Euclidean), D.. then A.. (KL)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a PROVABLY uniform value: everything derived from it (DMA bases, LDS
    // destinations) then stays in SGPRs, which the "s" operands of the DMA statements need
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 15, g = lane >> 4;
    const int S = gridDim.y, sp = blockIdx.y;
    const int g0 = (int)((int64_t)ngroups * sp / S);
    const int g1 = (int)((int64_t)ngroups * (sp + 1) / S);
    const int64_t r0 = (int64_t)blockIdx.x * 128 + wave * 16;

    // ---- DMA plan: roles by wave, so that every wave's vmcnt queue is homogeneous ----
    // (vmcnt retires in order: a shallow Y request behind deep V requests would force
    // the V requests to complete too.)
    //   waves 4..7 ("Y loaders"): YPW of the pieces (8 rows x 128 B) of the Y tiles of group grp+1
    //   waves 0..3 ("V loaders"): the V tiles of TWO waves each (w and w+4), VRING-1 groups ahead
    const bool yrole = wave >= 4;
    const int lw = wave & 3;
    const int ytile = (lw * YPW) / (KP / 8), p0 = (lw * YPW) % (KP / 8);
    const unsigned short* ysrc = ytile == 0 ? Yhi : Ylo;
    unsigned long long ybase = (unsigned long long)ysrc + (unsigned long long)g0 * 128ull;
    unsigned yoffs[YPW];
#pragma unroll
    for (int i = 0; i < YPW; ++i) {
        const int row = 8 * (p0 + i) + (lane >> 3), pos = lane & 7, chunk = pos ^ yswz(row);
        yoffs[i] = (unsigned)(((int64_t)row * ldy + 8 * chunk) * 2);
    }
    const unsigned ydst = (unsigned)(ytile * YT + p0 * 1024);
    // V: rows 4t + g of a wave's 16, position x holds chunk x ^ row
    const int64_t rblk = (int64_t)blockIdx.x * 128;
    unsigned long long vbaseA = (unsigned long long)(X + (rblk + lw * 16) * ldx) + (unsigned long long)g0 * 256ull;
    unsigned long long vbaseB = (unsigned long long)(X + (rblk + (lw + 4) * 16) * ldx) + (unsigned long long)g0 * 256ull;
    unsigned long long vstep = 256ull;
    unsigned voffs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = 4 * t + g;
        voffs[t] = (unsigned)(((int64_t)row * ldx + 4 * (x ^ row)) * 4);
    }
    if (tiled) {       // X stored tile-major: [R/128][ldx/64] tiles of [128 rows][64 cols], 32 KiB contiguous each
        const unsigned long long tile0 = (unsigned long long)X + ((unsigned long long)blockIdx.x * (ldx / 64) + g0) * 32768ull;
        vbaseA = tile0 + lw * 16 * 256; vbaseB = tile0 + (lw + 4) * 16 * 256; vstep = 32768ull;
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int row = 4 * t + g; voffs[t] = (unsigned)((row * 64 + 4 * (x ^ row)) * 4); }
    }
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
    const unsigned vdstA = smem0 + VOFF + lw * (VRING * 4096);
    const unsigned vdstB = smem0 + VOFF + (lw + 4) * (VRING * 4096);
    int yq = 0, vq = 0;                               // next Y buffer / V ring slot to fill
    auto issue_y = [&]() {                            // Y loaders only
#pragma unroll
        for (int i = 0; i < YPW; i += 4)
            dma_run4(ybase, smem0 + yq * YBUF + ydst + i * 1024, yoffs[i], yoffs[i + 1], yoffs[i + 2], yoffs[i + 3]);
        ybase += 128ull; yq ^= 1;
    };
    auto issue_v = [&]() {                            // V loaders only
        dma_run4_nt(vbaseA, vdstA + vq * 4096, voffs[0], voffs[1], voffs[2], voffs[3]);
        dma_run4_nt(vbaseB, vdstB + vq * 4096, voffs[0], voffs[1], voffs[2], voffs[3]);
        vbaseA += vstep; vbaseB += vstep; vq = (vq == VRING - 1) ? 0 : vq + 1;
    };

    // ---- loop-invariant LDS read offsets ----
    int ylane[2], ykl[2][2], vaoff[2][2], vroff[4], tro[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ylane[s] = x * 128 + 16 * ((4 * s + g) ^ yswz(x));           // + 2048 * (factor tile) + YT * (lo image)
        // KL: columns 32 s + 4 g .. + 3 and 32 s + 16 + 4 g .. + 3 of row x (8 bytes each)
        ykl[s][0] = x * 128 + 16 * ((4 * s + (g >> 1)) ^ yswz(x)) + 8 * (g & 1);
        ykl[s][1] = x * 128 + 16 * ((4 * s + 2 + (g >> 1)) ^ yswz(x)) + 8 * (g & 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) vaoff[s][h] = x * 256 + 16 * ((8 * s + 2 * g + h) ^ x);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) vroff[e] = x * 256 + 16 * ((4 * e + g) ^ x);
    {   // transposed reads: lane (q = x >> 2, p = x & 3) of group g addresses row 8g + q (+ 32 s + 4 t),
        // columns 16 e + 4 p .. + 3; yswz of that row = 2 (q >> 1 & 1) + 4 (g & 1) whatever s and t are
        const int q = x >> 2, pp = x & 3;
        const int fz = (((q >> 1) & 1) << 1) | ((g & 1) << 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) tro[e] = 128 * (8 * g + q) + 8 * (pp & 1) + 16 * ((2 * e + (pp >> 1)) ^ fz);
    }
    const unsigned char* vring = smem + VOFF + wave * (VRING * 4096);

    f32x4 acc[NJT];
#pragma unroll
    for (int j = 0; j < NJT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Gram by-product (KP = 64): Y Y^T (H H^T in the W phase, W^T W in the H phase) from the Y tiles
    // the blocks fetch anyway.  The work is SPREAD over the first `ng` row blocks -- row block b takes
    // the groups with (grp - g0) % ng == b of every split -- because the 32 extra MFMAs per group made
    // the single row block that used to do all of it the tail of the whole launch (W phase 150 us
    // with it, 123 us without).  Slab (b, split) = gram_part[b * S + split].  Wave w owns tile row
    // w>>1 and tile columns 2(w&1), 2(w&1)+1.
    const bool do_gram = WITH_GRAM && ((int)blockIdx.x < ng);
    const int git = wave >> 1, gj0 = 2 * (wave & 1);
    f32x4 gacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    double osum = 0.0;
    // Launches without the objective are bound by the bytes in flight: they refill a V slot as soon as every
    // wave holds its tile in registers (VRING groups ahead, one more barrier per group).  The launches that also
    // carry the objective are bound by instruction issue and the extra barrier costs more than the deeper
    // prefetch brings (config 5 W phase 3014 -> 3168 us, MUR-KL W phase 748 -> 812): those keep VRING - 1.
    constexpr bool EARLY = !WITH_OBJ || !WITH_A;       // (also the KL H phase: 730 -> 695 us with it, and the objective-only pass)
    constexpr int VAHEAD = EARLY ? VRING : VRING - 1;  // groups requested ahead of the one being consumed
    if (yrole) { if (g0 < g1) issue_y(); }
    else {
#pragma unroll
        for (int a = 0; a < VAHEAD; ++a) if (g0 + a < g1) issue_v();
    }
    // The Z fragments (ordinary vector loads) go out BEHIND the first DMAs, so the block pays one memory round
    // trip at its start instead of two.  vmcnt retires in order: the wait below also lands the DMAs issued
    // above, which the first group needs anyway, and every later counted wait only sees DMAs again.
    Frag8 zh[WITH_D ? KP / 32 : 1], zl[WITH_D ? KP / 32 : 1];
    if (WITH_D) {
#pragma unroll
        for (int s = 0; s < KP / 32; ++s) {
            zh[s].u = *reinterpret_cast<const uint4*>(Zhi + (r0 + x) * KP + 32 * s + 8 * g);
            zl[s].u = *reinterpret_cast<const uint4*>(Zlo + (r0 + x) * KP + 32 * s + 8 * g);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < KP / 32; ++s) { pinu(zh[s].u); pinu(zl[s].u); }
    }
    int ycur = 0, vcur = 0;
    for (int grp = g0; grp < g1; ++grp) {
        // Y loaders: Y(grp) is their newest request.  V loaders: V(grp+1 .. grp+VRING-1) may
        // stay in flight (8 DMAs per group), V(grp) must have landed.
        if (yrole) dma_wait_le<0>();
        else {
            const int ahead = min(VAHEAD - 1, g1 - 1 - grp);
            if (ahead >= 3) dma_wait_le<24>(); else if (ahead == 2) dma_wait_le<16>(); else if (ahead == 1) dma_wait_le<8>(); else dma_wait_le<0>();
        }
        __syncthreads();
        if (yrole) { if (grp + 1 < g1) issue_y(); }
        else if (!EARLY && grp + VRING - 1 < g1) issue_v();
        const unsigned char* ybuf = smem + ycur * YBUF;
        const unsigned char* vt = vring + vcur * 4096;

        // Explicit software pipeline (hipcc otherwise pairs every ds_read with its own
        // s_waitcnt right in front of the MFMA that uses it): stage st first ISSUES the fragment
        // reads of stage st+1 into the other register set, then runs its own 16 MFMAs.
        // Stages: A-product (k-step, half of the factor tiles) ..., then the residual product's
        // k-steps over the factors.  sched_barrier(0) pins the stage boundaries.
#define NMFX_FENCE() __builtin_amdgcn_sched_barrier(0)
        float4 va[2][2], vr[4];
        Frag8 fh[2][4], fl[2][4];                      // [register set][fragment]: hi and lo images
        auto issue = [&](int st, int set) {
            const bool a_stage = KL ? st >= ND : st < NA;
            if (a_stage) {                             // Y tile rows 16 jt.., k-step ast / (NJT / 4)
                const int ast = KL ? st - ND : st;
                const int ks = ast / (NJT / 4), half = ast % (NJT / 4);
                if (KL) {
                    const unsigned char* y0 = ybuf + ykl[ks][0] + half * 4 * 2048;
                    const unsigned char* y1 = ybuf + ykl[ks][1] + half * 4 * 2048;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint2 a0 = *reinterpret_cast<const uint2*>(y0 + j * 2048);
                        const uint2 a1 = *reinterpret_cast<const uint2*>(y1 + j * 2048);
                        const uint2 b0 = *reinterpret_cast<const uint2*>(y0 + j * 2048 + YT);
                        const uint2 b1 = *reinterpret_cast<const uint2*>(y1 + j * 2048 + YT);
                        fh[set][j].u = make_uint4(a0.x, a0.y, a1.x, a1.y);
                        fl[set][j].u = make_uint4(b0.x, b0.y, b1.x, b1.y);
                    }
                } else {
                    const unsigned char* ys = ybuf + ylane[ks] + half * 4 * 2048;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        fh[set][j].u = *reinterpret_cast<const uint4*>(ys + j * 2048);
                        fl[set][j].u = *reinterpret_cast<const uint4*>(ys + j * 2048 + YT);
                    }
                }
            } else {                                   // columns 16 e.. of factors 32 s.. 32 s + 31, transposed
                const int s = KL ? st : st - NA;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned char* ts = ybuf + tro[e] + s * 4096;
                    const uint2 h0 = lds_read_tr(ts), h1 = lds_read_tr(ts + 512);
                    const uint2 l0 = lds_read_tr(ts + YT), l1 = lds_read_tr(ts + YT + 512);
                    fh[set][e].u = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    fl[set][e].u = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            }
        };
        if (!KL && WITH_A) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) va[s][h] = *reinterpret_cast<const float4*>(vt + vaoff[s][h]);
        }
        if (WITH_D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vr[e] = *reinterpret_cast<const float4*>(vt + vroff[e]);
        }
        issue(0, 0);
        NMFX_FENCE();
        Frag8 vh[2], vl[2];
        if (!KL && WITH_A) {
#pragma unroll
            for (int s = 0; s < 2; ++s) split8(va[s][0], va[s][1], vh[s], vl[s]);
        }
        if (EARLY) {
            // Every wave now has its V tile of this group in registers, so the slot is refilled HERE, VRING groups
            // ahead, instead of at the next group boundary (VRING - 1 ahead): the stream rate follows the bytes in
            // flight (H phase of config 2: 96 -> 128 KiB per CU, 96 -> 87 us = 6.2 TB/s).  Costs one more barrier
            // per group: the V loader waves must know that the OTHER waves have read their tiles too.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (!yrole && grp + VRING < g1) issue_v();
        }
        float klpart = 0.f;
        f32x4 d[4];                                    // D tiles: d[e][reg] = (Z Y)[row x][16 e + 4 g + reg]
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int set = st & 1;
            if (st + 1 < NS) issue(st + 1, set ^ 1);
            NMFX_FENCE();
            const bool a_stage = KL ? st >= ND : st < NA;
            if (a_stage) {
                const int ast = KL ? st - ND : st;
                const int ks = ast / (NJT / 4), j0 = 4 * (ast % (NJT / 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j0 + j] = MFMA_BF16(vh[ks], fh[set][j], acc[j0 + j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j0 + j] = MFMA_BF16(vl[ks], fh[set][j], acc[j0 + j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j0 + j] = MFMA_BF16(vh[ks], fl[set][j], acc[j0 + j]);
                if (TERMS >= 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j0 + j] = MFMA_BF16(vl[ks], fl[set][j], acc[j0 + j]);
                }
            } else {
                const int s = KL ? st : st - NA;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = MFMA_BF16(fh[set][e], zh[WITH_D ? s : 0], d[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = MFMA_BF16(fl[set][e], zh[WITH_D ? s : 0], d[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = MFMA_BF16(fh[set][e], zl[WITH_D ? s : 0], d[e]);
                if (TERMS >= 4 && KL) {               // (the Euclidean residual is only summed into the objective, which
#pragma unroll                                        //  nothing is computed from: three terms whatever TERMS is)
                    for (int e = 0; e < 4; ++e) d[e] = MFMA_BF16(fl[set][e], zl[WITH_D ? s : 0], d[e]);
                }
            }
            NMFX_FENCE();
            if (KL && st == ND - 1) {
                // quotient x / (zy + 1e-9) in the D layout = the A operand of the permuted k-steps;
                // objective term x log(x / zy) [inf, nan -> 0] - x + zy
                float qv[4][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float vv[4] = {vr[e].x, vr[e].y, vr[e].z, vr[e].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // v_rcp_f32 / v_log_f32 (1 ulp each) instead of IEEE division and logf: this
                        // phase is VALU bound, and 0 * log(0 / p), x * log(x / 0) and 0 / 0 still come
                        // out as nan / inf / nan and are zeroed exactly like utils.py:24 does
                        const float pv = d[e][r];
                        if (WITH_OBJ) {
                            float t = vv[r] * (__builtin_amdgcn_logf(vv[r] * __builtin_amdgcn_rcpf(pv)) * 0.69314718055994531f);
                            t = (t != t || t == __builtin_inff()) ? 0.f : t;
                            klpart += (t - vv[r]) + pv;
                        }
                        qv[e][r] = vv[r] * __builtin_amdgcn_rcpf(pv + 1e-9f);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    split2(qv[2 * ks][0], qv[2 * ks][1], vh[ks].u.x, vl[ks].u.x);
                    split2(qv[2 * ks][2], qv[2 * ks][3], vh[ks].u.y, vl[ks].u.y);
                    split2(qv[2 * ks + 1][0], qv[2 * ks + 1][1], vh[ks].u.z, vl[ks].u.z);
                    split2(qv[2 * ks + 1][2], qv[2 * ks + 1][3], vh[ks].u.w, vl[ks].u.w);
                }
                NMFX_FENCE();
            }
            if (WITH_GRAM && st == NA - 1 && do_gram && ((grp - g0) % ng) == (int)blockIdx.x) {
                // Gram by-product: operands straight from the LDS tiles at wave-uniform tile rows
                // (A = rows 16*git.., B = rows 16*(gj0+c)..); row block b < ng takes every ng-th group
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned char* ys = ybuf + ylane[s];
                    Frag8 ah, al;
                    ah.u = *reinterpret_cast<const uint4*>(ys + git * 2048);
                    al.u = *reinterpret_cast<const uint4*>(ys + git * 2048 + YT);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        Frag8 bh, bl;
                        bh.u = *reinterpret_cast<const uint4*>(ys + (gj0 + c) * 2048);
                        bl.u = *reinterpret_cast<const uint4*>(ys + (gj0 + c) * 2048 + YT);
                        gacc[c] = MFMA_BF16(ah, bh, gacc[c]);
                        gacc[c] = MFMA_BF16(al, bh, gacc[c]);
                        gacc[c] = MFMA_BF16(ah, bl, gacc[c]);
                        if (TERMS >= 4) gacc[c] = MFMA_BF16(al, bl, gacc[c]);
                    }
                }
            }
        }
        if (WITH_OBJ && !KL) {
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float rx = vr[e].x - d[e][0], ry = vr[e].y - d[e][1];
                const float rz = vr[e].z - d[e][2], rw = vr[e].w - d[e][3];
                part += rx * rx + ry * ry + rz * rz + rw * rw;
            }
            osum += (double)part;
        }
        if (WITH_OBJ && KL) osum += (double)klpart;
#undef NMFX_FENCE
        ycur ^= 1;
        vcur = (vcur == VRING - 1) ? 0 : vcur + 1;
    }

    if (WITH_A) {
        float* out = Apart + ((int64_t)sp * R + r0) * KP;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(int64_t)(4 * g + r) * KP + jt * 16 + x] = acc[jt][r];
    }
    if (do_gram) {
        float* go = gram_part + ((int64_t)blockIdx.x * S + sp) * KP * KP;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                go[(int64_t)(16 * git + 4 * g + r) * KP + 16 * (gj0 + c) + x] = gacc[c][r];
    }
#ifdef NMFX_EXP_BLOCKTIME
    __syncthreads();
    if (threadIdx.x == 0 && dbg_b < 1024) nmfx_dbg_times[WITH_OBJ][1][dbg_b] = wall_clock64();
#endif
    if (WITH_OBJ) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) osum += __shfl_down(osum, off, 64);
        __syncthreads();                                   // everybody is done with the LDS tiles
        double* red = reinterpret_cast<double*>(smem);
        if (lane == 0) red[wave] = osum;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < 8; ++w) t += red[w];
            objpart[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (KL ? 1.0 : 0.5) * t;
        }
    }
}
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------
// The 32-row form of the Euclidean product launch for KP = 64 (W phase and H phase of MUR, the
// products of AO-ADMM / ADMM / ANLS): same arguments, same LDS budget, same DMA plan and the same
// results layout as xyt_bf16_kernel<64, WITH_OBJ, false, TERMS>, but on v_mfma_f32_32x32x16_bf16
// with the block's 8 waves arranged as 4 row groups (32 rows) x 2 column halves (32 of the group's
// 64 columns).  Why: the 16-row form reads every Y fragment for 16 rows of V and each wave reads its
// V tile twice (operand layout + accumulator layout): 2176 LDS-array cycles per 64-column group and
// CU against 1536 MFMA cycles per SIMD -- the LDS array, not the matrix pipe or HBM, was the W phase's
// longest queue.  Here a Y fragment feeds 32 rows (half the fragment reads per flop), the MFMA holds
// the vector issue port for 8 of 32 cycles instead of 8 of 16, and per wave and group there are
// 28 LDS reads (4 V, 8 Y rows, 16 transposed) instead of 56.
//   wave (rg = w & 3, hh = w >> 2): rows 32 rg .. + 31 of the block, columns 32 hh .. + 31 of each group
//   A-product  A[32 rows][64 factors] += V[32][32 cols] Y^T : 2 k-steps (16 columns) x 2 factor tiles
//   residual   D^T[32 cols][32 rows]  = Y^T[32 cols][64 factors] Z^T : 4 k-steps (16 factors); the rows of
//              this tile are a PERMUTATION of the columns (chosen through the addresses of the transposed
//              reads) such that every accumulator register sits on the lane that holds the same element of
//              V in the A-operand layout: one LDS read of V serves both products, residual = va - d
//   the two column halves of a row group hold partial A tiles; they are exchanged through LDS once,
//   after the last group (each wave finishes one factor tile).
// LDS: Y images as before ([64 factors][128 B] hi, lo; double buffered) with the chunk swizzle yswz32
// (tools/lab/swizzle_search.py: conflict free for the b128 row reads of the 32x32x16 B operand, its
// transposed b64 reads AND the 16x16x32 row reads of the Gram by-product); V ring per row group,
// [32 rows][64 cols] f32, chunk c of row r at c ^ (r & 15), 4 deep.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).v, (b).v, (c), 0, 0, 0)
__device__ __forceinline__ int yswz32(int row) {
    const int b1 = (row >> 1) & 1;
    return b1 | ((b1 ^ ((row >> 2) & 1)) << 1) | ((b1 ^ ((row >> 3) & 1)) << 2);
}

#ifdef NMFX_EXP_STAMPS         // experiment (tools/lab/stamps.py): where a wave's cycles go, per segment of the group loop
__device__ unsigned long long nmfx_dbg_stamps[2][256][8][6];   // [with objective][block][wave][wait, head, early barrier, mfma, cycles, realtime ticks]
extern "C" int nmfx_debug_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nmfx_dbg_stamps), sizeof(nmfx_dbg_stamps)) == hipSuccess ? 0 : -1;
}
#define NMFX_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NMFX_STAMP(var) do { } while (0)
#endif
#pragma clang fp contract(fast)
// ABL (experiments only, NMFX_EXP_ABLATE builds; results are then WRONG): bit 0 no DMA in the loop, bit 1 no bf16 split,
// bit 2 no residual arithmetic, bit 3 no fragment reads after the first group, bit 4 no MFMAs
// KL = true (MUR with the KL divergence, nmf/mur.py:24-27, 40-43; W phase: X = V, Y = H, Z = W; H phase: X = V^T, Y = W^T,
// Z = H^T): A_part = (X / (Z Y + 1e-9)) Y^T.  The product Z Y comes first (the residual product's transposed reads), the
// quotient is formed in registers -- accumulator register 4 a + c sits on the lane that holds the same element of X in the
// operand layout (see above), so it is split to bf16 where it stands and used as the A operand of the second product -- and
// with WITH_OBJ the objective term x log(x / zy) [inf, nan -> 0] - x + zy (nmf/utils.py:21-26) comes from the same registers.
// KP = 128 (config 3 and 5): four factor tiles, eight k-steps of the residual product, Y images of 2 x 16 KiB per group
// (double buffered: 64 KiB), V ring 3 deep; stages of (k-step, pair of factor tiles) so that the fragment registers stay
// at 2 x 16; no cross-group pipeline (its second V register set does not fit next to 64 accumulator and 64 Z registers),
// no Gram by-product, Euclidean only.
// NPROB = 2 (pair mode, KP = 128 with the objective): the factor columns [0, 64) and [64, 128) belong to two independent problems
// on the same X -- the A-product is the k = 128 product as it stands (A = [X Y_0^T | X Y_1^T]); the residual product is closed
// after the first four k-steps (objective of problem 0), restarted, and closed again after the last four (problem 1):
// objpart[p][split][block].
#ifdef NMFX_EXP_REVERSE
// Experiment (tools/lab/rev_probe.py): every other launch walks its groups backwards, so that what the previous launch streamed
// LAST is read FIRST -- how much of a V-sized stream does the 256 MiB Infinity Cache serve on the turn-around?
__device__ int nmfx_rev_flag = 0;
__global__ void nmfx_set_rev_kernel(int v) { nmfx_rev_flag = v; }
extern "C" int nmfx_debug_set_reverse(void* stream, int v) {
    hipLaunchKernelGGL(nmfx_set_rev_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), v);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

// TEMPORAL (r3): the V tiles with the default cache policy instead of non-temporal.  A V-sized stream goes 6 % faster
// non-temporal (config 2: 212.8 vs 225.9 us per iteration) and leaves nothing in the 256 MiB Infinity Cache; but when V and V^T
// TOGETHER fit there -- a rank's shard of a strongly scaled problem: 2 x 64 MiB at config 2 over 8 GPUs -- the default policy keeps
// both resident from one iteration to the next: 62.7 -> 60.4 us per iteration at 2048 rows (tools/lab/ab_iter.py shard8), 81.0 -> 79.9
// at 4096 rows, 122 -> 134 (worse) at 8192.  Chosen per launch from the size of the two copies (nmfx_bf16_temporal).
// WITH_A = false (with WITH_OBJ, Euclidean): only the residual objective of (Z, Y) -- no A-product, no exchange, nothing stored but
// the objective partials (the closing objective of a run, ADMM's objective of (w, h), ANLS's un-fused objective).
// SK = true (r4; Euclidean products at k padded to 128 -- AO-ADMM's two V-sized products): a STREAM-K partition instead of the
// (row block) x (split) grid.  The (row block, group) units of the product, row block major, are dealt to `sk_workers` workgroups in
// contiguous runs of (nearly) equal length; a run that crosses a row block boundary is two SEGMENTS (seg[]: row block, first group,
// end group, slab), each with its own accumulators, epilogue and slab -- a static plan (nmfx_sk_plan), so the sums stay in a fixed
// order.  What it buys: the number of workers is free, and with one CU left over the grid carries a SIDE JOB in block `sk_workers`:
// the one-workgroup f64 inversion of the sub-problem's shifted Gram matrix (ao_prepare_body), which otherwise is a launch of its
// own -- 38 us twice per outer iteration of config 3 with 255 CUs idle, since nothing fits on a CU beside a block of this kernel.
// (Two streams were measured first, tools/lab/overlap_probe.hip: workgroups are bound to an (XCC, SE) slot round-robin at dispatch,
// so a second dispatch's single workgroup waits for ITS slot even with other CUs idle, and the fork / join events cost 15 us.)
struct XytSide {                 // the side job: Minv = (sum of gslabs slabs of gsrc + rho I)^-1, rho = trace / k (fixed_rho < 0) -- nmf/ao_admm.py:53-55
    const float* gsrc; int gslabs; int k; float* Minv; DevState* st; double fixed_rho;
    int defer;                   // 1: a non-positive pivot is noted in st->notpd_pending and the launch that records obj[j] decides (the H side, whose
};                               // objective may stop the run first); 0: it ends the run here, as nmf/ao_admm.py:55 does (the W side: no stop rule follows it)
// VAUX = true (r4; with WITH_OBJ, without the A-product): the m x n auxiliaries of the KL-loss ADMM variants instead of the residual --
// nmf/ao_admm.py:87-93, nmf/admm.py:306-314: with P = Z Y (= w h_aux, or its transpose) where the accumulator stands,
//   v_bar = P - dual_v;  v_aux = ((v_bar - 1) + sqrt((v_bar - 1)^2 + 4 v)) / 2;  dual_v += v_aux - P;  S = v_aux + dual_v
// dual_v (vaux_dv) and S (vaux_s: the only form in which v_aux enters the next right-hand side) are TILE-MAJOR like X, in the
// orientation of X.  They are streamed straight to / from registers: the loads of group g + 1 go out behind the barrier of group g,
// the stores of group g behind the barrier of group g + 1 (stores count in vmcnt as well: issued at the END of a group they would
// sit in front of the next group's counted waits).  The counted waits of the V loaders hold for the steady state only (groups
// with three predecessors and three successors: 24 / 20 younger loads); the first and last groups of a block drain the queue.
template <bool WITH_OBJ, int TERMS, int ABL = 0, bool KL = false, int KP = 64, int NPROB = 1, bool TEMPORAL = false, bool WITH_A = true, int NW = 8, bool SK = false,
          int VMODE = 0>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void xyt32_bf16_kernel(
    const float* __restrict__ X, int64_t ldx,
    const unsigned short* __restrict__ Yhi, const unsigned short* __restrict__ Ylo, int64_t ldy,
    const unsigned short* __restrict__ Zhi, const unsigned short* __restrict__ Zlo,
    float* __restrict__ Apart, double* __restrict__ objpart, float* __restrict__ gram_part, int64_t R,
    int ngroups, const int* __restrict__ flag, int ng,
    const int4* __restrict__ sk_seg = nullptr, const int* __restrict__ sk_first = nullptr, int sk_workers = 0, XytSide side = XytSide(),
    float* __restrict__ vaux_dv = nullptr, float* __restrict__ vaux_s = nullptr, const int* __restrict__ flag2 = nullptr,
    int xpriv = 0)               // r5: X (the S of the KL-loss ADMM variants) lies in the auxiliaries kernel's register order inside its tiles (kl_dv_pos)
{
#define MFMA32X(a, b, c) ((ABL & 16) ? (c) : MFMA32_BF16(a, b, c))
    if (*flag) return;
    if (flag2 && *flag2) return;                       // (the inner stop of a KL-ADMM sub-problem has fired: a no-op like every later round)
    // VMODE 1 = VAUX (above); VMODE 2 = the KL objective of (Z, Y) alone -- sum x log(x / zy) [inf, nan -> 0] - x + zy, nmf/utils.py:21-26 --
    // where the objective-only form evaluates the residual: the closing objective of the KL-loss ADMM variants (r4; the exact-f32 pass
    // took 400 us at 16384 x 8192, k = 128)
    // VMODE 3 = VAUXF (r5): the auxiliaries AND the next round's right-hand-side product in one pass -- the KL form at KP = 128 (product Z Y, the
    // element-wise step where the accumulator stands, its result split to bf16 as the A operand of the second product) with v_aux / dual_v in the
    // place of the quotient: S = v_aux + dual_v never goes to HBM between two rounds of a sub-problem (it is stored when the launch finds that its
    // round is the sub-problem's last: `ng` != 0, or the round's `terminate` norms -- `objpart` points at them, `sk_workers` = their block count --
    // pass the test that the next round's kernel will apply).  Per round X read + dual_v read and written: 3 V-sized streams instead of 5.
    // VMODE 4 = XGATHER (r5): the plain product form with X read from the tiles of the OTHER orientation (see the DMA plan).  A template mode, not
    // an argument: as a run-time branch in read_va and in the group step it cost EVERY instantiation 5-13 % (config 2 W phase 117 -> 128 us, H phase
    // 88 -> 100, config 5 H phase 1745 -> 1900: profiles/r05_gather_as_runtime_branch_regression.txt)
    constexpr bool VAUX = VMODE == 1, KLOBJ = VMODE == 2, VAUXF = VMODE == 3, XGATHER = VMODE == 4;
    static_assert(VMODE != 4 || (!WITH_OBJ && WITH_A && !KL && NPROB == 1 && NW == 8 && !SK && ABL == 0), "XGATHER: the objective-free Euclidean product form");
    static_assert(VMODE == 0 || VMODE == 3 || VMODE == 4 || (WITH_OBJ && !WITH_A && !KL && NPROB == 1 && NW == 8 && !SK && ABL == 0), "VMODE: the objective-only form of the kernel");
    static_assert(VMODE != 3 || (!WITH_OBJ && WITH_A && KL && NPROB == 1 && NW == 8 && !SK && ABL == 0), "VAUXF: the one-register-set KL form (kl128_group, at either KP)");
    static_assert(!SK || (KP == 128 && !KL && NPROB == 1 && NW == 8 && WITH_A && ABL == 0), "stream-K: the Euclidean k = 128 products");
    static_assert(KP == 64 || (KP == 128 && ABL == 0), "KP = 64 or 128");
    static_assert(NPROB == 1 || (NPROB == 2 && KP == 128 && WITH_OBJ && !KL), "pair mode: the k = 128 W phase with its objective");
    static_assert(WITH_A || (WITH_OBJ && !KL && NPROB == 1 && ABL == 0), "without the A-product the launch must at least compute the Euclidean objective");
    static_assert(NW == 8 || (NW == 4 && KP == 64 && NPROB == 1 && WITH_A && VMODE == 0), "four-wave blocks: the k = 64 products");
    // KL in its one-register-set form (kl128_group): KP = 128, the fused auxiliaries (VAUXF), and -- experiment NMFX_EXP_KLNW4 -- KP = 64 in four-wave blocks
    constexpr bool KLONE = KL && (KP == 128 || VMODE == 3 || NW == 4);
    constexpr int NRG = NW / 2;                        // row groups of 32 rows per block (NW = 4: 64-row blocks, two of them per CU)
    constexpr int YR = (KL && !KLONE) ? 3 : 2;       // Y ring (KL, KP = 64: the second product runs one group behind the first)
    constexpr int YT = KP * 128, YBUF = 2 * YT, VOFF = YR * YBUF, VRING = (KP == 64 && !KL && NW == 8) ? 4 : 3, VSLOT = 8192;
    constexpr int NT = KP / 32, NTP = NT / 2;          // factor tiles of 32, pairs of them (one A stage each per k-step)
    constexpr int NK = KP / 16;                        // k-steps of the product Z Y
    constexpr int YPW = 2 * (KP / 8) / NRG;            // Y pieces (8 rows x 128 B) per loader wave and group
    constexpr bool WITH_D = WITH_OBJ || KL;            // the product Z Y is formed
    constexpr int NA = WITH_A ? 2 * NTP : 0, ND = (WITH_OBJ && !KL) ? NK / 2 : 0, NS = NA + ND;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & (NRG - 1), hh = wave / NRG;  // compute role: row group, column half
    bool store_s = false;                              // VAUXF: this round is the sub-problem's last (wave-uniform)
    if constexpr (VAUXF) {
        store_s = ng != 0;
        if (!store_s) store_s = inner_round_fired(reinterpret_cast<const double*>(objpart), sk_workers, reinterpret_cast<double*>(smem));
    }
    const int n31 = lane & 31, b = lane >> 5;          // 32x32x16 coordinates
    const int x = lane & 15, g = lane >> 4;            // 16x16x32 coordinates (Gram by-product)
#ifdef NMFX_EXP_BLOCKTIME
    if (SK && threadIdx.x == 0 && blockIdx.x < 1024) nmfx_dbg_times[WITH_OBJ][0][blockIdx.x] = wall_clock64();
#endif
    if constexpr (SK) {
        if ((int)blockIdx.x == sk_workers) {           // the side job (see above); its LDS use (72 KiB at KP = 128) is inside this kernel's
            ao_prepare_body<KP, false>(reinterpret_cast<double*>(smem), side.gsrc, side.gslabs, side.k, side.Minv, side.st,
                                       side.fixed_rho, side.defer != 0);
#ifdef NMFX_EXP_BLOCKTIME
            if (threadIdx.x == 0 && blockIdx.x < 1024) nmfx_dbg_times[WITH_OBJ][1][blockIdx.x] = wall_clock64();
#endif
            return;
        }
    }
    int seg_lo = 0, seg_hi = 1;
    if constexpr (SK) {
        seg_lo = __builtin_amdgcn_readfirstlane(sk_first[blockIdx.x]);
        seg_hi = __builtin_amdgcn_readfirstlane(sk_first[blockIdx.x + 1]);
    }
    // (the body below is one segment; not re-indented)
#pragma nounroll
    for (int seg = seg_lo; seg < seg_hi; ++seg) {
    const int S = gridDim.y;
    int bx_ = blockIdx.x, sp_ = blockIdx.y;
    int g0_ = (int)((int64_t)ngroups * sp_ / S), g1_ = (int)((int64_t)ngroups * (sp_ + 1) / S);
    if constexpr (SK) {
        const int4 pl = sk_seg[seg];
        bx_ = __builtin_amdgcn_readfirstlane(pl.x); g0_ = __builtin_amdgcn_readfirstlane(pl.y);
        g1_ = __builtin_amdgcn_readfirstlane(pl.z); sp_ = __builtin_amdgcn_readfirstlane(pl.w);
    }
    // SK: the segment's groups are visited CYCLICALLY from group sk_start (slab | start << 8 in the plan): every worker is then at a
    // group index congruent to its own clock modulo the common run length, i.e. the whole grid reads the same few Y tiles at the same
    // time, as the (row block) x (split) grid does by construction -- with plain runs the workers' group indices drift apart and all
    // of Y (4 / 8 MiB at config 3) cycles through every XCD's L2: +16 / +11 us on the two products
    const int sk_start = SK ? (sp_ >> 8) : 0;
    if constexpr (SK) sp_ &= 255;
    const int bx = bx_, sp = sp_, g0 = g0_, g1 = g1_;
    const int64_t oidx = SK ? (int64_t)seg : (int64_t)blockIdx.y * gridDim.x + blockIdx.x;     // objective partial of this segment
    const int64_t r0 = (int64_t)bx * (32 * NRG) + rg * 32;

    // ---- DMA plan (roles by wave as in xyt_bf16_kernel: homogeneous vmcnt queues) ----
    //   waves 4..7: 4 of the 16 pieces (8 rows x 128 B) of the Y tiles of group grp + 1
    //   waves 0..3: the V tile [32][64] of row group lw (8 pieces of 4 rows x 256 B), VRING (- 1) groups ahead
    const bool yrole = wave >= NRG;
    const int lw = wave & (NRG - 1);
    const int ytile = (lw * YPW) / (KP / 8), p0 = (lw * YPW) % (KP / 8);
    const unsigned short* ysrc = ytile == 0 ? Yhi : Ylo;
#ifdef NMFX_EXP_REVERSE
    const bool rev = __builtin_amdgcn_readfirstlane(nmfx_rev_flag) != 0;
#else
    constexpr bool rev = false;
#endif
    const int gfirst = SK ? sk_start : rev ? g1 - 1 : g0;
    int y_left = g1 - gfirst, v_left = g1 - gfirst;     // (SK) requests until the wrap back to group g0
    const long long span_y = (long long)(g1 - g0) * 128ll, span_v = (long long)(g1 - g0) * 32768ll;
    const long long ystep = rev ? -128ll : 128ll;
    long long vstep = rev ? -32768ll : 32768ll, vstep2 = vstep;      // (XGATHER: the steps alternate; they are swapped after every group's requests)
    unsigned long long ybase = (unsigned long long)ysrc + (unsigned long long)gfirst * 128ull;
    unsigned yoffs[YPW];
#pragma unroll
    for (int i = 0; i < YPW; ++i) {
        const int row = 8 * (p0 + i) + (lane >> 3), pos = lane & 7, chunk = pos ^ yswz32(row);
        yoffs[i] = (unsigned)(((int64_t)row * ldy + 8 * chunk) * 2);
    }
    const unsigned ydst = (unsigned)(ytile * YT + p0 * 1024);
    // X is tile-major: [R/128][ldx/64] tiles of [128 rows][64 cols], 32 KiB contiguous each
    const unsigned long long tile0 = (unsigned long long)X + ((unsigned long long)(bx / (4 / NRG)) * (ldx / 64) + gfirst) * 32768ull;
    unsigned long long vbaseA = tile0 + (unsigned long long)((bx % (4 / NRG)) * NRG + lw) * 32 * 256, vbaseB = vbaseA + 16 * 256;
    unsigned voffs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int row = 4 * t + g; voffs[t] = (unsigned)((row * 64 + 4 * (x ^ row)) * 4); }   // rows 4t+g (< 16): row & 15 = row
    if (xpriv && !XGATHER) {
        // X in the auxiliaries kernel's register order (kl_dv_pos): the KiB of piece j of wave w = row group + 4 hh is what wave (row group, hh)
        // of THIS kernel wants in its va[j >> 1][j & 1], lane for lane.  So the pieces travel whole (a contiguous KiB per request instead of
        // 4 rows x 256 B) and stay in that order in the LDS slot: [hh][j][lane] float4s, read back with one conflict-free ds_read_b128 each
        vbaseA = tile0 + (unsigned long long)((bx % (4 / NRG)) * NRG + lw) * 4096; vbaseB = vbaseA + 16384;
#pragma unroll
        for (int t = 0; t < 4; ++t) voffs[t] = (unsigned)(t * 1024 + lane * 16);
    }
    if constexpr (XGATHER) {
        // X in the tiles of the OTHER orientation (rows there = columns here), in their register order: S as the auxiliaries kernel of the other
        // sub-problem left it -- the first product of a sub-problem needs no transposed copy (r5).  Row group lw of this block is column half
        // hhB = lw & 1 of the other side's column group 2 bx + (lw >> 1); this kernel's group g is the half g & 1 of the other side's row block
        // g >> 1, i.e. its row groups 2 (g & 1), 2 (g & 1) + 1.  The LDS slot is [64 columns c][32 rows] f32: request p (0..7), lane i fetches the
        // 16 bytes of (c = 8 p + (i >> 3), rows 4 (i & 7) ..) from piece w = 2 (g & 1) + (p >> 2) + 4 hhB, j = 2 s + e, lane (c & 31) + 32 bb, where
        // 4 (i & 7) = 16 s + 8 bb + 4 e -- eight runs of 128 bytes per request; read_va takes its four columns with four ds_read_b32
        const unsigned long long tilesB = (unsigned long long)(R / 64);
        vbaseA = (unsigned long long)X + ((unsigned long long)(gfirst >> 1) * tilesB + 2 * bx + (lw >> 1)) * 32768ull + 16384ull * (lw & 1) + 8192ull * (gfirst & 1);
        vbaseB = vbaseA + 4096;
        const int i7 = lane & 7, s_ = i7 >> 2, bb = (i7 >> 1) & 1, e_ = i7 & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) voffs[t] = (unsigned)((2 * s_ + e_) * 1024 + ((lane >> 3) + 32 * bb) * 16 + t * 128);
        const long long even = 8192ll, odd = (long long)tilesB * 32768ll - 8192ll;
        vstep = (gfirst & 1) ? odd : even; vstep2 = (gfirst & 1) ? even : odd;
    }
    const unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
    const unsigned vdstA = smem0 + VOFF + lw * (VRING * VSLOT), vdstB = vdstA + 4096;
    int yq = 0, vq = 0;
    auto issue_y = [&]() {
#pragma unroll
        for (int i = 0; i < YPW; i += 4)
            dma_run4(ybase, smem0 + yq * YBUF + ydst + i * 1024, yoffs[i], yoffs[i + 1], yoffs[i + 2], yoffs[i + 3]);
        ybase += ystep; yq = (yq == YR - 1) ? 0 : yq + 1;
        if (SK && --y_left == 0) ybase -= span_y;
    };
    auto dma_step = [&](int st) {                      // a quarter (V loaders) / half (Y loaders, stages 0 and 1) of a group's requests
        if (yrole) {
            if (st < YPW / 2) dma_run2(ybase, smem0 + yq * YBUF + ydst + st * 2048, yoffs[2 * (st < YPW / 2 ? st : 0)], yoffs[2 * (st < YPW / 2 ? st : 0) + 1]);
            if (st == YPW / 2 - 1) { ybase += ystep; yq = (yq == YR - 1) ? 0 : yq + 1; }
        } else {
#ifdef NMFX_EXP_VFRONT         // experiment (r4): the V requests of a group in its first two stages (four pieces each) instead of two per stage
            if (st == 0) (TEMPORAL ? dma_run4 : dma_run4_nt)(vbaseA, vdstA + vq * VSLOT, voffs[0], voffs[1], voffs[2], voffs[3]);
            if (st == 1) (TEMPORAL ? dma_run4 : dma_run4_nt)(vbaseB, vdstB + vq * VSLOT, voffs[0], voffs[1], voffs[2], voffs[3]);
#else
            (TEMPORAL ? dma_run2 : dma_run2_nt)(st < 2 ? vbaseA : vbaseB, (st < 2 ? vdstA : vdstB) + vq * VSLOT + (st & 1) * 2048, voffs[2 * (st & 1)], voffs[2 * (st & 1) + 1]);
#endif
            if (st == 3) { vbaseA += vstep; vbaseB += vstep; if constexpr (XGATHER) { const long long t_ = vstep; vstep = vstep2; vstep2 = t_; }
                           vq = (NW == 4) ? vq + 1 - 3 * (vq >> 1) : ((vq == VRING - 1) ? 0 : vq + 1); }
        }
    };
    auto issue_v = [&]() {      // rows 16..31 of the tile: same lane offsets (row & 15 repeats), base + 16 rows
        (TEMPORAL ? dma_run4 : dma_run4_nt)(vbaseA, vdstA + vq * VSLOT, voffs[0], voffs[1], voffs[2], voffs[3]);
        (TEMPORAL ? dma_run4 : dma_run4_nt)(vbaseB, vdstB + vq * VSLOT, voffs[0], voffs[1], voffs[2], voffs[3]);
        vbaseA += vstep; vbaseB += vstep; if constexpr (XGATHER) { const long long t_ = vstep; vstep = vstep2; vstep2 = t_; }
        vq = (NW == 4) ? vq + 1 - 3 * (vq >> 1) : ((vq == VRING - 1) ? 0 : vq + 1);
        if (SK && --v_left == 0) { vbaseA -= span_v; vbaseB -= span_v; }
    };

    // ---- loop-invariant LDS read offsets ----
    int vaoff[2][2], yrow[2], tro[2], ylane[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int e = 0; e < 2; ++e) vaoff[s][e] = XGATHER ? 4 * (8 * hh + 4 * s + 2 * b + e) * 128 + n31 * 4
                                                : xpriv ? ((4 * hh + 2 * s + e) * 64 + lane) * 16 : n31 * 256 + 16 * ((8 * hh + 4 * s + 2 * b + e) ^ (n31 & 15));
        yrow[s] = n31 * 128 + 16 * ((4 * hh + 2 * s + b) ^ yswz32(n31));          // + 4096 * (factor tile) + YT * (lo image)
        ylane[s] = x * 128 + 16 * ((4 * s + g) ^ yswz32(x));                       // + 2048 * (16-factor tile)
    }
    {   // transposed reads: lane (G = lane >> 4, q, p) addresses factor row 16 s + 8 (G >> 1) + 4 u + q, and its piece of 4
        // columns goes to lanes 4 p .. 4 p + 3 of the group, i.e. to rows m = 16 (G & 1) + 4 p + c of the product tile.  The
        // pieces are PERMUTED: row m = 8 a + 4 beta + c of the tile is column 16 (a >> 1) + 8 beta + 4 (a & 1) + c of V (piece p
        // from columns 8 (p & 1) + 4 (p >> 1) of the 16), so that the accumulator register 4 a + c of lane half b holds column
        // 16 (a >> 1) + 8 b + 4 (a & 1) + c -- exactly the element of the operand-layout V registers va[a >> 1][a & 1].  One
        // LDS read of the V tile serves both products (tools/lab/swizzle_search.py: still conflict free).
        const int q = (lane >> 2) & 3, pp = lane & 3, G = lane >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = 8 * (G >> 1) + 4 * u + q;
            tro[u] = 128 * row + 16 * ((4 * hh + 2 * (G & 1) + (pp & 1)) ^ yswz32(row)) + 8 * (pp >> 1);   // + 2048 * s
        }
    }
    const unsigned char* vring = smem + VOFF + rg * (VRING * VSLOT);

    f32x16 accA[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[t][r] = 0.f;
    const bool do_gram = KP == 64 && !KL && WITH_A && bx < ng;  // Gram by-product: as in xyt_bf16_kernel (16x16x32 tiles)
    constexpr int NGT = 16 / NW;                       // Gram tiles (16 x 16) per wave: tile row git, tile columns gj0 .. gj0 + NGT - 1
    const int git = wave / (4 / NGT), gj0 = NGT * (wave % (4 / NGT));
    f32x4 gacc[NGT];
#pragma unroll
    for (int c = 0; c < NGT; ++c) gacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double osum = 0.0, olog = 0.0;                     // (olog: KL, the sum of x log2(q))
    double osum2 = 0.0;                                // (NPROB = 2: the residual sum of the second problem)
    // Two loop structures.
    //  !WITH_OBJ (H phase and the other objective-free products; bound by the bytes in flight): a V slot is refilled as
    //    soon as every wave holds its tile in registers, VRING groups ahead, at the price of a second barrier per group.
    //  WITH_OBJ (W phase; twice the MFMAs): in-kernel stamps (tools/lab/stamps.py) showed a third of every group spent in
    //    its HEAD -- the V tile's LDS reads and the bf16 split, which all 8 waves run together right after the barrier
    //    with the matrix pipe idle (~1000 of 3300 cycles).  So the loop is software-pipelined ACROSS groups: the barrier
    //    of group g also guarantees V(g + 1), whose reads and split run between the MFMAs of group g into the other
    //    register set (the group loop is unrolled by two, the sets alternate).  The slot of V(g) is free at that barrier
    //    (read during group g - 1), so V(g + 4) goes out there: 3 groups in flight as before.
    constexpr bool PIPE = WITH_OBJ && !KL && KP == 64 && WITH_A;
    constexpr bool EARLY = !PIPE;
    constexpr int VAHEAD = (KL && !KLONE) ? 2 : VRING;   // groups requested before the loop
    Frag8 zh[WITH_D ? NK : 1], zl[WITH_D ? NK : 1];    // Z^T fragments: row r0 + n31, factors 16 s + 8 b .. + 7
    if (WITH_D) {                                      // ahead of the DMAs: vmcnt retires in order, so waiting for these does not drain the stream
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            zh[s].u = *reinterpret_cast<const uint4*>(Zhi + (r0 + n31) * KP + 16 * s + 8 * b);
            zl[s].u = *reinterpret_cast<const uint4*>(Zlo + (r0 + n31) * KP + 16 * s + 8 * b);
        }
    }
    if (yrole) { if (g0 < g1) issue_y(); }
    else {
#pragma unroll
        for (int a = 0; a < VAHEAD; ++a) if (g0 + a < g1) issue_v();
    }
    if (WITH_D) {
        if (yrole) dma_wait_le<YPW>(); else dma_wait_le<8 * VRING>();      // (an upper bound of the DMAs issued above: the Z loads are older)
#pragma unroll
        for (int s = 0; s < NK; ++s) { pinu(zh[s].u); pinu(zl[s].u); }
    }
#ifdef NMFX_EXP_FP8D
    constexpr bool FP8D = WITH_D && !KL && KP == 128 && NPROB == 1 && VMODE == 0 && !(WITH_OBJ && !KL && KP == 64 && WITH_A);
    fp8_i32x8 z8h[FP8D ? NK / 4 : 1], z8l[FP8D ? NK / 4 : 1], y8h, y8l;
    if constexpr (FP8D) {
#pragma unroll
        for (int q8 = 0; q8 < NK / 4; ++q8)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int a0, a1;
                frag_to_fp8(zh[4 * q8 + j], 1.f, a0, a1); z8h[q8][2 * j] = a0; z8h[q8][2 * j + 1] = a1;
                frag_to_fp8(zl[4 * q8 + j], 0.001953125f, a0, a1); z8l[q8][2 * j] = a0; z8l[q8][2 * j + 1] = a1;
            }
    }
#endif
    int ycur = 0, vcur = 0;
#ifdef NMFX_EXP_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, acc_wait = 0, acc_head = 0, acc_early = 0, acc_mfma = 0;
    const unsigned long long tbeg = __builtin_amdgcn_s_memtime(), rbeg = __builtin_amdgcn_s_memrealtime();
#endif
#define NMFX_FENCE() __builtin_amdgcn_sched_barrier(0)
    // a wave's V tile of one group: the f32 values (row n31, columns 16 s + 8 b + 4 e ..), their split A operands, and the
    // product tile Z Y of the same group (register 4 a + c <-> va[a >> 1][a & 1], component c)
    struct VRegs { Frag8 vh[2], vl[2]; float4 va[2][2]; f32x16 d; unsigned slow; };
    // residual of a finished group: d[4 a + c] = (Z Y)[row n31][column 32 hh + 8 a + 4 b + c].  PIPE: evaluated one group
    // late, right behind the next group's barrier, where it covers the latency of that group's first fragment reads (at
    // the end of its own group it was a serial tail of ~250 cycles -- MFMA result, 16 dependent adds, an f64 add -- with
    // the matrix pipe idle in front of the barrier)
    auto residual = [&](const VRegs& v, double& osum) {
        if (ABL & 4) { osum += (double)(v.d[0] + v.d[5] + v.va[0][0].x + v.va[1][1].y); return; }
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;  // four chains instead of one
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 x4 = v.va[a >> 1][a & 1];
            const float rx = x4.x - v.d[4 * a], ry = x4.y - v.d[4 * a + 1];
            const float rz = x4.z - v.d[4 * a + 2], rw = x4.w - v.d[4 * a + 3];
            p0 += rx * rx; p1 += ry * ry; p2 += rz * rz; p3 += rw * rw;
        }
        osum += (double)((p0 + p1) + (p2 + p3));
    };
    auto read_va = [&](const unsigned char* vt, float4 (&va)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if constexpr (XGATHER) {               // (slot [column][row]: the lane's four columns are 128 bytes apart; conflict free, 32 consecutive words per half wave)
                    const float* q_ = reinterpret_cast<const float*>(vt + vaoff[s][e]);
                    va[s][e] = make_float4(q_[0], q_[32], q_[64], q_[96]);
                } else va[s][e] = *reinterpret_cast<const float4*>(vt + vaoff[s][e]);
            }
    };
    // VAUX: the lane's 16 elements of a group -- row n31 of its row group, 16-byte chunks 8 hh + 4 s + 2 b + e -- of dual_v (in) and
    // of the new S / dual_v (out), at the positions of the tile-major X.  The loads are inline asm: hipcc does not count the LDS-DMA
    // requests in its own vmcnt bookkeeping and would put its waits where it believes the loads are the only ones in flight.
    f32x4 vx_cur[2][2], vx_next[2][2];
    float4 vx_s[2][2], vx_d[2][2];
    auto vaux_tile = [&](int grp) { return ((int64_t)bx * (ldx / 64) + grp) * 8192; };           // floats (NW = 8: one tile per block and group)
    // r5: dual_v is only ever touched here and by the transposes between the sub-problems, so inside its 128 x 64 tiles it lives in THIS
    // kernel's register order -- piece j = 2 s + e of wave w, lane l at float4 index (4 w + j) 64 + l (kl_dv_pos below): every load /
    // store instruction of a wave is one contiguous KiB instead of 64 pieces of 16 bytes at a 256-byte stride.  S lies the same way: the
    // product kernels that stream it by LDS-DMA give every lane its own source address anyway (`xpriv`).
    const unsigned vx_priv = (unsigned)(((wave * 4) * 64 + lane) * 16);                         // bytes; + 1024 j
    auto vaux_load = [&](int grp, f32x4 (&dst)[2][2]) {
        const unsigned long long src = (unsigned long long)(vaux_dv + vaux_tile(grp));
        asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                     : "=&v"(dst[0][0]), "=&v"(dst[0][1]), "=&v"(dst[1][0]), "=&v"(dst[1][1]) : "v"(vx_priv), "s"(src) : "memory");
    };
    auto vaux_store = [&](int grp) {
        const int64_t offp = vaux_tile(grp) + vx_priv / 4;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                *reinterpret_cast<float4*>(vaux_s + offp + 256 * (2 * s2 + e)) = vx_s[s2][e];
                *reinterpret_cast<float4*>(vaux_dv + offp + 256 * (2 * s2 + e)) = vx_d[s2][e];
            }
    };
    auto vaux_update = [&](const VRegs& v) {           // d[4 a + c] <-> va[a >> 1][a & 1], component c
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(vx_cur[s2][e]));      // (behind the counted wait: the uses stay below it)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 x4 = v.va[a >> 1][a & 1];
            const f32x4 u4 = vx_cur[a >> 1][a & 1];
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
            float so[4], uo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float p = v.d[4 * a + c];
                const float t = (p - u4[c]) - 1.f;
                const float va_ = 0.5f * (t + __builtin_sqrtf(t * t + 4.f * xs[c]));
                uo[c] = u4[c] + (va_ - p);
                so[c] = va_ + uo[c];
            }
            vx_s[a >> 1][a & 1] = make_float4(so[0], so[1], so[2], so[3]);
            vx_d[a >> 1][a & 1] = make_float4(uo[0], uo[1], uo[2], uo[3]);
        }
    };
    Frag8 fh[2][2], fl[2][2];                          // Y fragments: [register set][fragment]
    // PIPE: the LAST stage of a group (residual k-steps 2, 3; its fragments are in register set 1 by then) is carried
    // over the next group's barrier and runs while that group's first fragment reads are in flight -- with nothing but
    // those reads behind the barrier the matrix pipe sat idle for their latency, every group.
    auto carried_stage = [&](f32x16& d, bool dma_on) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            d = MFMA32X(fh[1][ss], zh[WITH_D ? 2 + ss : 0], d);
#ifndef NMFX_EXP_R2
            d = MFMA32X(fl[1][ss], zh[WITH_D ? 2 + ss : 0], d);
#endif
            if (ss == 0) { NMFX_FENCE(); if (dma_on) dma_step(0); NMFX_FENCE(); }
            d = MFMA32X(fh[1][ss], zl[WITH_D ? 2 + ss : 0], d);
        }
    };
    // one group: `cur` holds V(grp) (PIPE: filled during the previous group), `nxt` receives V(grp + 1) (PIPE)
    auto group = [&](int grp, VRegs& cur, VRegs& nxt) {
        NMFX_STAMP(ts0);
        if (ABL & 1) { }
        else if (yrole) dma_wait_le<0>();
        else if (VAUX) {                               // steady state: 12 (VRING - 1) loads are younger than V(grp) (see the kernel's VAUX note)
            if (grp >= g0 + VRING && grp + VRING - 1 < g1) dma_wait_le<12 * (VRING - 1)>(); else dma_wait_le<0>();
        }
        else if (PIPE) {                               // V(grp + 1) must have landed; V(grp + 2) (, V(grp + 3): VRING = 4) may stay in flight
            const int ahead = g1 - 2 - grp;
            if (VRING >= 4 && ahead >= 2) dma_wait_le<16>(); else if (ahead >= 1) dma_wait_le<8>(); else dma_wait_le<0>();
        } else {
            const int ahead = min(VAHEAD - 1, g1 - 1 - grp);
            if (ahead >= 3) dma_wait_le<24>(); else if (ahead == 2) dma_wait_le<16>(); else if (ahead == 1) dma_wait_le<8>(); else dma_wait_le<0>();
        }
        __syncthreads();
        NMFX_STAMP(ts1);
        // PIPE: the next requests -- Y(grp + 1) into the other Y buffer, V(grp + VRING) into the slot of V(grp), which every
        // wave read before this barrier -- go out in pairs BETWEEN the MFMAs of the stages below (dma_step): issued in one
        // burst at the top of the group they cost 500-800 cycles per loader wave with the matrix pipe idle (stamps)
        const bool dma_on = PIPE && !(ABL & 1) && (yrole ? grp + 1 < g1 : grp + VRING < g1);
        if (!PIPE && yrole) { if (grp + 1 < g1) issue_y(); }
        if constexpr (VAUX) {
            vaux_load(grp + 1 < g1 ? grp + 1 : grp, vx_next);       // (behind the last group: a harmless reload, so that every group issues the same)
            if (grp > g0) vaux_store(grp - 1);
        }
        const unsigned char* ybuf = smem + ycur * YBUF;
        const unsigned char* vt = vring + vcur * VSLOT;
        const unsigned char* vtn = vring + (vcur == VRING - 1 ? 0 : vcur + 1) * VSLOT;
        auto issue = [&](int st, int set) {
            if ((ABL & 8) && grp > g0 + 1) { pinu(fh[set][0].u); pinu(fh[set][1].u); pinu(fl[set][0].u); pinu(fl[set][1].u); return; }
            if (st < NA) {                             // Y rows (factors) 32 (2 tp + t) + n31, columns of k-step st / NTP
                const unsigned char* ys = ybuf + yrow[st / NTP] + (st % NTP) * 8192;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fh[set][t].u = *reinterpret_cast<const uint4*>(ys + t * 4096);
                    fl[set][t].u = *reinterpret_cast<const uint4*>(ys + t * 4096 + YT);
                }
            } else {                                   // factors 16 s .. + 15 of the wave's 32 columns, transposed
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int s = 2 * (st - NA) + ss;
                    const unsigned char* t0 = ybuf + tro[0] + s * 2048;
                    const unsigned char* t1 = ybuf + tro[1] + s * 2048;
                    const uint2 h0 = lds_read_tr(t0), h1 = lds_read_tr(t1);
                    const uint2 l0 = lds_read_tr(t0 + YT), l1 = lds_read_tr(t1 + YT);
                    fh[set][ss].u = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    fl[set][ss].u = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            }
        };
        if (!PIPE) {
            read_va(vt, cur.va);
            issue(0, 0);
            NMFX_FENCE();
            if (WITH_A) {
#pragma unroll
                for (int s = 0; s < 2; ++s) split8(cur.va[s][0], cur.va[s][1], cur.vh[s], cur.vl[s]);
            }
        } else {
            issue(0, 0);
            NMFX_FENCE();
            carried_stage(nxt.d, dma_on);              // of the previous group (zero fragments in front of the first one)
            NMFX_FENCE();
        }
        NMFX_STAMP(ts2);
        if (EARLY) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (!yrole && grp + VRING < g1) issue_v();
        }
        NMFX_STAMP(ts3);
        f32x16& d = cur.d;
        if (WITH_OBJ) {
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int set = st & 1;
            if (st + 1 < NS) issue(st + 1, set ^ 1);
            NMFX_FENCE();
            if (st < NA) {
                const int ks = st / NTP, t0 = 2 * (st % NTP);
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vh[ks], fh[set][t], accA[t0 + t]);
                if (PIPE) { NMFX_FENCE(); if (dma_on) dma_step(st + 1); NMFX_FENCE(); }
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vl[ks], fh[set][t], accA[t0 + t]);
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vh[ks], fl[set][t], accA[t0 + t]);
                if (TERMS >= 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vl[ks], fl[set][t], accA[t0 + t]);
                }
            } else if (!(PIPE && st == NS - 1)) {      // (PIPE: the last stage is carried over the barrier)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int s = WITH_D ? 2 * (st - NA) + ss : 0;
#ifdef NMFX_EXP_FP8D
                    if constexpr (FP8D) {              // hi hi in bf16; the fragments of four k-steps make one K = 64 operand of each image
                        d = MFMA32X(fh[set][ss], zh[s], d);
                        int a0, a1;
                        frag_to_fp8(fh[set][ss], 1.f, a0, a1); y8h[2 * (s & 3)] = a0; y8h[2 * (s & 3) + 1] = a1;
                        frag_to_fp8(fl[set][ss], 0.001953125f, a0, a1); y8l[2 * (s & 3)] = a0; y8l[2 * (s & 3) + 1] = a1;
                        if ((s & 3) == 3) {            // (lo images are stored x 2^9: scale byte 127 - 9)
                            d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(y8l, z8h[s >> 2], d, 0, 0, 0, 118, 0, 127);
                            d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(y8h, z8l[s >> 2], d, 0, 0, 0, 127, 0, 118);
                        }
                        continue;
                    }
#endif
                    d = MFMA32X(fh[set][ss], zh[s], d);
#ifndef NMFX_EXP_R2            // experiment (r5): the residual product with TWO terms (Y_lo Z_hi dropped): W phase -11 % (k = 128), -8 % (k = 64);
                               // recorded objective +1.2e-5 .. +4.3e-4 off nmfx_objective_f64 (three terms: 1.6e-8), jitter 1.5e-6 (7e-9): not shipped, LAB_NOTES R5
                    d = MFMA32X(fl[set][ss], zh[s], d);
#endif
                    if (PIPE && ss == 0) { NMFX_FENCE(); if (dma_on) dma_step(st + 1); NMFX_FENCE(); }
                    d = MFMA32X(fh[set][ss], zl[s], d);
                }
                if (NPROB == 2 && st == NA + ND / 2 - 1) {     // factors 0 .. 63 are done: the first problem's residual, then start over
                    residual(cur, osum);
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[r] = 0.f;
                }
            }
            if (PIPE && st == 0) {                     // residual of the previous group, between the MFMAs of stage 0
#pragma unroll
                for (int a = 0; a < 4; ++a) pin4(nxt.va[a >> 1][a & 1]);   // (opaque: keeps the arithmetic on this side of the barrier ...
                residual(nxt, osum);
                asm volatile("" : "+v"(osum));         //  ... and of the later stages: hipcc otherwise sinks it to the end of the group)
                // V(grp + 1) from its LDS slot into the registers the residual has just released (unconditional: behind the last
                // group this fetches a stale slot that nothing uses; a branch here costs a full LDS drain)
                read_va(vtn, nxt.va);
            }
            if (PIPE) {                                // the bf16 split of V(grp + 1) rides between the MFMAs of stages 1 and 2
                if (ABL & 2) {
                    if (st == 1) { nxt.vh[0].u = make_uint4(__float_as_uint(nxt.va[0][0].x), __float_as_uint(nxt.va[0][0].y), __float_as_uint(nxt.va[0][0].z), __float_as_uint(nxt.va[0][0].w));
                                   nxt.vl[0].u = make_uint4(__float_as_uint(nxt.va[0][1].x), __float_as_uint(nxt.va[0][1].y), __float_as_uint(nxt.va[0][1].z), __float_as_uint(nxt.va[0][1].w)); }
                    if (st == 2) { nxt.vh[1].u = make_uint4(__float_as_uint(nxt.va[1][0].x), __float_as_uint(nxt.va[1][0].y), __float_as_uint(nxt.va[1][0].z), __float_as_uint(nxt.va[1][0].w));
                                   nxt.vl[1].u = make_uint4(__float_as_uint(nxt.va[1][1].x), __float_as_uint(nxt.va[1][1].y), __float_as_uint(nxt.va[1][1].z), __float_as_uint(nxt.va[1][1].w)); }
                } else {
                if (st == 1) split8(nxt.va[0][0], nxt.va[0][1], nxt.vh[0], nxt.vl[0]);
                if (st == 2) split8(nxt.va[1][0], nxt.va[1][1], nxt.vh[1], nxt.vl[1]);
                }
            }
            NMFX_FENCE();
            if (st == NA - 1 && do_gram && ((grp - g0) % ng) == bx) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned char* ys = ybuf + ylane[s];
                    Frag8 ah, al;
                    ah.u = *reinterpret_cast<const uint4*>(ys + git * 2048);
                    al.u = *reinterpret_cast<const uint4*>(ys + git * 2048 + YT);
#pragma unroll
                    for (int c = 0; c < NGT; ++c) {
                        Frag8 bh, bl;
                        bh.u = *reinterpret_cast<const uint4*>(ys + (gj0 + c) * 2048);
                        bl.u = *reinterpret_cast<const uint4*>(ys + (gj0 + c) * 2048 + YT);
                        gacc[c] = MFMA_BF16(ah, bh, gacc[c]);
                        gacc[c] = MFMA_BF16(al, bh, gacc[c]);
                        gacc[c] = MFMA_BF16(ah, bl, gacc[c]);
                        if (TERMS >= 4) gacc[c] = MFMA_BF16(al, bl, gacc[c]);
                    }
                }
            }
        }
        if constexpr (VAUX) {
            // dual_v of this group: requested a group ago.  V loaders: 20 younger loads in the steady state (V(grp + VRING - 1), dual_v and V of
            // this group); Y loaders: their wait at the top of the group has drained it
            if (!yrole) { if (grp >= g0 + 1 && grp + VRING < g1) dma_wait_le<20>(); else dma_wait_le<0>(); }
            vaux_update(cur);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 2; ++e) vx_cur[s2][e] = vx_next[s2][e];
        }
        else if constexpr (KLOBJ) {
            float k0 = 0.f, k1 = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 x4 = cur.va[a >> 1][a & 1];
                const float vv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // x log(x / p) - x + p = p ((1 + u) log(1 + u) - u), u = (x - p) / p.  Near the optimum the objective is 1e-5 of
                    // sum x, so the three-term form cancels five digits and the hardware log / rcp (1 ulp, slightly biased)
                    // showed as 8e-5 of the objective at 16384 x 8192; for |u| <= 1/8 the series sum_{k >= 2} (-u)^k / (k (k - 1))
                    // has no cancellation at all (next term 1e-8 relative).  Elsewhere the reference's expression, with its
                    // inf / nan -> 0 (x = 0 or p = 0: u is -1, inf or nan, never small)
                    const float pv = cur.d[4 * a + c], rp = __builtin_amdgcn_rcpf(pv);
                    const float u = (vv[c] - pv) * rp;
                    const float poly = 0.5f + u * (-1.f / 6.f + u * (1.f / 12.f + u * (-1.f / 20.f + u * (1.f / 30.f + u * (-1.f / 42.f + u * (1.f / 56.f))))));
                    float t = vv[c] * (__builtin_amdgcn_logf(vv[c] * rp) * 0.69314718055994531f);
                    t = (t != t || t == __builtin_inff()) ? 0.f : t;
                    const float term = (__builtin_fabsf(u) <= 0.125f) ? pv * (u * u) * poly : (t - vv[c]) + pv;
                    if (c & 1) k1 += term; else k0 += term;
                }
            }
            osum += (double)(k0 + k1);
        }
        else if (WITH_OBJ && !PIPE) residual(cur, NPROB == 2 ? osum2 : osum);          // (PIPE: one group late, see above)
        NMFX_STAMP(ts4);
#ifdef NMFX_EXP_STAMPS
        acc_wait += ts1 - ts0; acc_head += ts2 - ts1; acc_early += ts3 - ts2; acc_mfma += ts4 - ts3;
#endif
        ycur ^= 1;
        vcur = (NW == 4) ? vcur + 1 - 3 * (vcur >> 1) : ((vcur == VRING - 1) ? 0 : vcur + 1);
    };
    // ---- KL: product, quotient, product -- software-pipelined across groups ----
    // In its first form (one register set: product Z Y, quotient + split with the matrix pipe idle, second product with the
    // objective terms between its MFMAs, two barriers per group) a group took ~4000 cycles against 1536 of MFMA: the eight
    // waves run in step behind the barriers, so the serial quotient section of one wave never met the MFMAs of another.
    // Now iteration g runs  [product Z Y of group g  ||  objective terms of group g - 1]  and then
    // [second product of group g - 1  ||  quotient + split of group g]  on two register sets (P / Q, loop unrolled by two):
    // every VALU section has independent MFMAs around it.  Y ring of three (the second product of g - 1 reads Y(g - 1) while
    // Y(g + 1) lands), V ring of three (V(g + 2) goes into the slot of V(g - 1) behind the barrier of group g), one barrier
    // per group, the DMA requests in pairs between the MFMAs as in the Euclidean pipeline.  (A second barrier per group that
    // frees the slot of V(g) as soon as every wave holds the tile, with V(g + 3) requested behind it -- three groups in
    // flight instead of two -- measured no faster: W phase 537 vs 524 us, H phase 448 vs 450, tools/lab/ab_phase.py.)
#ifdef NMFX_EXP_Q1             // experiment (r5): the quotient enters the second product as ONE bf16 image (q_hi Y_hi + q_hi Y_lo):
                               // config 4 W phase 550 -> 495 us, H phase 466 -> 441 -- and WH against the oracle 5e-7 -> 3.4e-5 .. 6e-5 at contractions of
                               // 4096 .. 16384, 1.3e-4 at 384 (tools/lab/kl_q1_check.py): half a digit inside north_star's 1e-4.  Not the default, not shipped.
    constexpr bool Q1 = KL && KP == 64;
#else
    constexpr bool Q1 = false;
#endif
    auto kl_iter = [&](int grp, VRegs& cur, VRegs& prv, auto do_d_t, auto do_a_t) {
        constexpr bool DO_D = decltype(do_d_t)::value, DO_A = decltype(do_a_t)::value;
        const unsigned char* ybuf = smem + ycur * YBUF;                              // Y(grp)
        const unsigned char* ybp = smem + (ycur == 0 ? YR - 1 : ycur - 1) * YBUF;    // Y(grp - 1)
        bool dma_on = false;
        auto issue_d = [&](int half, int set) {        // factors 16 s .. + 15 (s = 2 half, 2 half + 1) of the wave's 32 columns, transposed
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int s = 2 * half + ss;
                const unsigned char* t0 = ybuf + tro[0] + s * 2048;
                const unsigned char* t1 = ybuf + tro[1] + s * 2048;
                const uint2 h0 = lds_read_tr(t0), h1 = lds_read_tr(t1);
                const uint2 l0 = lds_read_tr(t0 + YT), l1 = lds_read_tr(t1 + YT);
                fh[set][ss].u = make_uint4(h0.x, h0.y, h1.x, h1.y);
                fl[set][ss].u = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        };
        auto issue_a = [&](int ks, int set) {          // Y(grp - 1) rows (factors) 32 t + n31, columns of k-step ks
            const unsigned char* ys = ybp + yrow[ks];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fh[set][t].u = *reinterpret_cast<const uint4*>(ys + t * 4096);
                fl[set][t].u = *reinterpret_cast<const uint4*>(ys + t * 4096 + YT);
            }
        };
        // Objective terms x log(x / zy) [inf, nan -> 0] - x + zy (utils.py:23-26).  The kernel is bound by VALU issue (two
        // waves per SIMD: ~270 VALU instructions with 48 transcendentals per wave and group next to 24 MFMAs), so the terms
        // are formed with as few instructions as the arithmetic allows:
        //   * zy >= 2^-5 in all four registers of a chunk (signed-integer minimum of the bit patterns): there zy + 1e-9f == zy
        //     in f32, so the quotient q of the update IS x / zy -- its logarithm is taken one group later (kl_chunk, between
        //     the MFMAs of the next first product; q stays in the accumulator registers), multiplied with v_mul_legacy_f32
        //     (0 * -inf = 0: the x = 0 entries that utils.py:24 zeroes), and zy - x is summed at once in packed f32;
        //   * otherwise (small, zero or negative zy; wave-uniform branch): the exact expression with its own reciprocal,
        //     inf / nan zeroed as utils.py:24 does, here and now, and the chunk is marked (`slow`) so that kl_chunk skips it.
        // The sums of x log2(q) and of zy - x go to f64 separately (they cancel to second order).
        float klog = 0.f;
        auto kl_chunk = [&](int a) {
            if (prv.slow & (1u << a)) return;          // (wave-uniform)
            const float4 x4 = prv.va[a >> 1][a & 1];
            float t0, t1, t2, t3;
            xlog2_legacy4(x4, prv.d[4 * a], prv.d[4 * a + 1], prv.d[4 * a + 2], prv.d[4 * a + 3], t0, t1, t2, t3);
            klog += (t0 + t1) + (t2 + t3);
        };
        // quotient x / (zy + 1e-9) where the accumulator stands (v_rcp_f32, 1 ulp, instead of an IEEE division)
        float4 qa[2][2];
        f32x2 klin2 = {0.f, 0.f};
        auto quot = [&](int a) {
            const float4 x4 = cur.va[a >> 1][a & 1];
            const f32x2 x01 = {x4.x, x4.y}, x23 = {x4.z, x4.w};
            const f32x2 p01 = {cur.d[4 * a], cur.d[4 * a + 1]}, p23 = {cur.d[4 * a + 2], cur.d[4 * a + 3]};
            const f32x2 e01 = p01 + 1e-9f, e23 = p23 + 1e-9f;
            const f32x2 r01 = {__builtin_amdgcn_rcpf(e01.x), __builtin_amdgcn_rcpf(e01.y)};
            const f32x2 r23 = {__builtin_amdgcn_rcpf(e23.x), __builtin_amdgcn_rcpf(e23.y)};
            const f32x2 q01 = x01 * r01, q23 = x23 * r23;
            qa[a >> 1][a & 1] = make_float4(q01.x, q01.y, q23.x, q23.y);
            if (WITH_OBJ) {
                const int mn = min(min(__float_as_int(p01.x), __float_as_int(p01.y)), min(__float_as_int(p23.x), __float_as_int(p23.y)));
                if (__builtin_amdgcn_ballot_w64(mn < 0x3D000000) != 0ull) {
                    const float vv[4] = {x4.x, x4.y, x4.z, x4.w};
                    float k0 = 0.f, k1 = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float pv = cur.d[4 * a + c];
                        float t = vv[c] * (__builtin_amdgcn_logf(vv[c] * __builtin_amdgcn_rcpf(pv)) * 0.69314718055994531f);
                        t = (t != t || t == __builtin_inff()) ? 0.f : t;
                        if (c & 1) k1 += (t - vv[c]) + pv; else k0 += (t - vv[c]) + pv;
                    }
                    klin2.x += k0; klin2.y += k1;
                    cur.slow |= 1u << a;
                } else {
                    klin2 += (p01 - x01) + (p23 - x23);
                }
                cur.d[4 * a] = q01.x; cur.d[4 * a + 1] = q01.y; cur.d[4 * a + 2] = q23.x; cur.d[4 * a + 3] = q23.y;
            }
        };
        NMFX_STAMP(ts0);
        if (DO_D) {
            if (yrole) dma_wait_le<0>();               // Y(grp)
            else if (grp + 1 < g1) dma_wait_le<8>();   // V(grp); V(grp + 1) may stay in flight
            else dma_wait_le<0>();
            __syncthreads();
            NMFX_STAMP(ts1);
            dma_on = !(ABL & 1) && (yrole ? grp + 1 < g1 : grp + 2 < g1);
            f32x16& d = cur.d;
            cur.slow = 0u;
            read_va(vring + vcur * VSLOT, cur.va);
            issue_d(0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            issue_d(1, 1);
            NMFX_FENCE();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int s = 2 * half + ss;
                    d = MFMA32X(fh[half][ss], zh[WITH_D ? s : 0], d);
                    d = MFMA32X(fl[half][ss], zh[WITH_D ? s : 0], d);
                    NMFX_FENCE();
                    if (dma_on) dma_step(s);
                    if (DO_A && WITH_OBJ) kl_chunk(s);
                    NMFX_FENCE();
                    d = MFMA32X(fh[half][ss], zl[WITH_D ? s : 0], d);
                    if (TERMS >= 4) d = MFMA32X(fl[half][ss], zl[WITH_D ? s : 0], d);
                }
                NMFX_FENCE();
                if (DO_A) issue_a(half, half);         // (behind the MFMAs that read this register set)
                NMFX_FENCE();
            }
        } else {
            NMFX_STAMP(ts1);
            issue_a(0, 0); issue_a(1, 1);
            NMFX_FENCE();
        }
        NMFX_STAMP(ts2);
        if (DO_A) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t] = MFMA32X(prv.vh[ks], fh[ks][t], accA[t]);
                NMFX_FENCE();
                if (DO_D) quot(2 * ks); else if (WITH_OBJ) kl_chunk(2 * ks);
                NMFX_FENCE();
                if (!Q1) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) accA[t] = MFMA32X(prv.vl[ks], fh[ks][t], accA[t]);
                }
                NMFX_FENCE();
                if (DO_D) quot(2 * ks + 1); else if (WITH_OBJ) kl_chunk(2 * ks + 1);
                NMFX_FENCE();
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t] = MFMA32X(prv.vh[ks], fl[ks][t], accA[t]);
                if (TERMS >= 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) accA[t] = MFMA32X(prv.vl[ks], fl[ks][t], accA[t]);
                }
                NMFX_FENCE();
                if (DO_D) { if (Q1) hi8(qa[ks][0], qa[ks][1], cur.vh[ks]); else split8(qa[ks][0], qa[ks][1], cur.vh[ks], cur.vl[ks]); }
                NMFX_FENCE();
            }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) quot(a);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { if (Q1) hi8(qa[s2][0], qa[s2][1], cur.vh[s2]); else split8(qa[s2][0], qa[s2][1], cur.vh[s2], cur.vl[s2]); }
        }
        if (WITH_OBJ) {
            if (DO_A) olog += (double)klog;
            if (DO_D) osum += (double)(klin2.x + klin2.y);
            asm volatile("" : "+v"(osum), "+v"(olog));
        }
#ifdef NMFX_EXP_STAMPS                                 // (KL: wait + barrier | first product with the objective terms of the group before | second product with quotient + split)
        NMFX_STAMP(ts3);
        acc_wait += ts1 - ts0; acc_head += ts2 - ts1; acc_mfma += ts3 - ts2;
#endif
        if (DO_D) {
            ycur = (ycur == YR - 1) ? 0 : ycur + 1;
            vcur = (NW == 4) ? vcur + 1 - 3 * (vcur >> 1) : ((vcur == VRING - 1) ? 0 : vcur + 1);
        }
    };
    // ---- KL with KP = 128 (r3; MUR-KL at k = 128 ran the 16-row kernel before): one register set, no cross-group pipeline -- as for the
    // Euclidean KP = 128 form a second V / product set does not fit next to 64 accumulator and 64 Z registers.  Per group: product
    // Z Y (8 k-steps), quotient + objective terms where the accumulator stands, bf16 split, second product (2 k-steps x 4 factor
    // tiles) on the SAME Y(grp) buffer; the two-barrier loop of the objective-free products (V slot refilled as soon as the tile is
    // in registers).  Fragment reads run one stage ahead on two register sets, D stages first, then the A stages.
    auto kl128_group = [&](int grp, VRegs& cur) {
        if (yrole) {                                   // (VAUXF: the last group's stores and this group's dual_v loads are younger than Y(grp))
            if constexpr (VAUXF) dma_wait_le_rt(4 + (grp > g0 ? (store_s ? 8 : 4) : 0)); else dma_wait_le<0>();
        }
        else {
            const int ahead = min(VAHEAD - 1, g1 - 1 - grp);
            if constexpr (VAUXF) {
                // vmcnt retires in order and counts the dual_v loads (4 per group, issued in the middle of the group before) and stores (4, or 8
                // with S, in front of them) as well: younger than V(grp) are, besides V(grp + 1 ..), the loads of this and the last two groups and
                // the stores of the last three, as far as this block has run that many
                const int d = grp - g0;
                dma_wait_le_rt(8 * ahead + 4 * min(d + 1, 3) + (store_s ? 8 : 4) * min(d, 3));
            }
            else if (ahead >= 2) dma_wait_le<16>(); else if (ahead == 1) dma_wait_le<8>(); else dma_wait_le<0>();
        }
        __syncthreads();
        if (yrole) { if (grp + 1 < g1) issue_y(); }
        const unsigned char* ybuf = smem + ycur * YBUF;
        const unsigned char* vt = vring + vcur * VSLOT;
        constexpr int NDK = NK / 2, NAK = 2 * NTP, NSK = NDK + NAK;       // stages: D (two k-steps each), then A (k-step, pair of tiles)
        auto issue_k = [&](int u, int set) {
            if (u < NDK) {
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int s = 2 * u + ss;
                    const unsigned char* t0 = ybuf + tro[0] + s * 2048;
                    const unsigned char* t1 = ybuf + tro[1] + s * 2048;
                    const uint2 h0 = lds_read_tr(t0), h1 = lds_read_tr(t1);
                    const uint2 l0 = lds_read_tr(t0 + YT), l1 = lds_read_tr(t1 + YT);
                    fh[set][ss].u = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    fl[set][ss].u = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            } else {
                const int st = u - NDK;
                const unsigned char* ys = ybuf + yrow[st / NTP] + (st % NTP) * 8192;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fh[set][t].u = *reinterpret_cast<const uint4*>(ys + t * 4096);
                    fl[set][t].u = *reinterpret_cast<const uint4*>(ys + t * 4096 + YT);
                }
            }
        };
        read_va(vt, cur.va);
        issue_k(0, 0);
        NMFX_FENCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (!yrole && grp + VRING < g1) issue_v();
        f32x16& d = cur.d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
        float klog = 0.f;
        f32x2 klin2 = {0.f, 0.f};
        float4 qa[2][2];
#pragma unroll
        for (int u = 0; u < NSK; ++u) {
            const int set = u & 1;
            if (u + 1 < NSK) issue_k(u + 1, set ^ 1);
            NMFX_FENCE();
            if (u < NDK) {
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int s = 2 * u + ss;
                    d = MFMA32X(fh[set][ss], zh[WITH_D ? s : 0], d);
                    d = MFMA32X(fl[set][ss], zh[WITH_D ? s : 0], d);
                    d = MFMA32X(fh[set][ss], zl[WITH_D ? s : 0], d);
                    if (TERMS >= 4 && !VAUXF) d = MFMA32X(fl[set][ss], zl[WITH_D ? s : 0], d);   // (VAUXF: P = Z Y with three terms, as the separate auxiliaries launch forms it)
                }
                if (VAUXF && u == NDK - 1) {           // v_aux, dual_v, S where the accumulator stands; S split as the second product's A operand
                    NMFX_FENCE();
                    // dual_v(grp): younger in the queue are only this group's V (V loaders: 8) / Y (Y loaders: YPW) requests
                    if (yrole) { if (grp + 1 < g1) dma_wait_le<YPW>(); else dma_wait_le<0>(); }
                    else { if (grp + VRING < g1) dma_wait_le<8>(); else dma_wait_le<0>(); }
                    vaux_update(cur);
                    {
                        const int64_t offp = vaux_tile(grp) + vx_priv / 4;
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int e = 0; e < 2; ++e) *reinterpret_cast<float4*>(vaux_dv + offp + 256 * (2 * s2 + e)) = vx_d[s2][e];
                        if (store_s) {
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                                for (int e = 0; e < 2; ++e) *reinterpret_cast<float4*>(vaux_s + offp + 256 * (2 * s2 + e)) = vx_s[s2][e];
                        }
                    }
                    // dual_v of the NEXT group into the registers the update has just released (there is no room for a second set): in flight
                    // through the second product, the barrier and the next product Z Y
                    if (grp + 1 < g1) vaux_load(grp + 1, vx_cur);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) split8(vx_s[s2][0], vx_s[s2][1], cur.vh[s2], cur.vl[s2]);
                    NMFX_FENCE();
                } else
                if (u == NDK - 1) {                    // quotient x / (zy + 1e-9) (v_rcp_f32) and the objective terms, as in kl_iter
                    NMFX_FENCE();
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float4 x4 = cur.va[a >> 1][a & 1];
                        const f32x2 x01 = {x4.x, x4.y}, x23 = {x4.z, x4.w};
                        const f32x2 p01 = {d[4 * a], d[4 * a + 1]}, p23 = {d[4 * a + 2], d[4 * a + 3]};
                        const f32x2 e01 = p01 + 1e-9f, e23 = p23 + 1e-9f;
                        const f32x2 r01 = {__builtin_amdgcn_rcpf(e01.x), __builtin_amdgcn_rcpf(e01.y)};
                        const f32x2 r23 = {__builtin_amdgcn_rcpf(e23.x), __builtin_amdgcn_rcpf(e23.y)};
                        const f32x2 q01 = x01 * r01, q23 = x23 * r23;
                        qa[a >> 1][a & 1] = make_float4(q01.x, q01.y, q23.x, q23.y);
                        if (WITH_OBJ) {
                            const int mn = min(min(__float_as_int(p01.x), __float_as_int(p01.y)), min(__float_as_int(p23.x), __float_as_int(p23.y)));
                            if (__builtin_amdgcn_ballot_w64(mn < 0x3D000000) != 0ull) {      // small, zero or negative zy: the exact expression
                                const float vv[4] = {x4.x, x4.y, x4.z, x4.w};
                                float k0 = 0.f, k1 = 0.f;
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const float pv = d[4 * a + c];
                                    float t = vv[c] * (__builtin_amdgcn_logf(vv[c] * __builtin_amdgcn_rcpf(pv)) * 0.69314718055994531f);
                                    t = (t != t || t == __builtin_inff()) ? 0.f : t;
                                    if (c & 1) k1 += (t - vv[c]) + pv; else k0 += (t - vv[c]) + pv;
                                }
                                klin2.x += k0; klin2.y += k1;
                            } else {                   // zy + 1e-9f == zy: q IS x / zy; x log2(q) with 0 * -inf = 0, zy - x summed in packed f32
                                klin2 += (p01 - x01) + (p23 - x23);
                                float t0, t1, t2, t3;
                                xlog2_legacy4(x4, q01.x, q01.y, q23.x, q23.y, t0, t1, t2, t3);
                                klog += (t0 + t1) + (t2 + t3);
                            }
                        }
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) split8(qa[s2][0], qa[s2][1], cur.vh[s2], cur.vl[s2]);
                    NMFX_FENCE();
                }
            } else {
                const int st = u - NDK, ks = st / NTP, t0 = 2 * (st % NTP);
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vh[ks], fh[set][t], accA[t0 + t]);
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vl[ks], fh[set][t], accA[t0 + t]);
#pragma unroll
                for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vh[ks], fl[set][t], accA[t0 + t]);
                if (TERMS >= 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) accA[t0 + t] = MFMA32X(cur.vl[ks], fl[set][t], accA[t0 + t]);
                }
            }
            NMFX_FENCE();
        }
        if (WITH_OBJ) { olog += (double)klog; osum += (double)(klin2.x + klin2.y); }
        ycur ^= 1;
        vcur = (NW == 4) ? vcur + 1 - 3 * (vcur >> 1) : ((vcur == VRING - 1) ? 0 : vcur + 1);
    };
    VRegs P, Q;
    if (PIPE) {
#pragma unroll
        for (int a = 0; a < 4; ++a) Q.va[a >> 1][a & 1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 16; ++r) Q.d[r] = 0.f;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) { fh[1][ss].u = make_uint4(0u, 0u, 0u, 0u); fl[1][ss].u = make_uint4(0u, 0u, 0u, 0u); }
    }
    if (PIPE && g0 < g1) {                             // V(g0) into registers before the loop
        if (!yrole) {
            const int ahead = min(VAHEAD - 1, g1 - 1 - g0);
            if (ahead >= 3) dma_wait_le<24>(); else if (ahead == 2) dma_wait_le<16>(); else if (ahead == 1) dma_wait_le<8>(); else dma_wait_le<0>();
        }
        __syncthreads();
        read_va(vring, P.va);
#pragma unroll
        for (int s = 0; s < 2; ++s) split8(P.va[s][0], P.va[s][1], P.vh[s], P.vl[s]);
    }
    if constexpr (VAUX || VAUXF) { if (g0 < g1) vaux_load(g0, vx_cur); }       // (behind the V requests above; VAUX: the first group's waits drain it)
    // (r5) KL: the younger wave of each SIMD (4-7, the Y loaders) loses the issue arbitration and ends its sections ~300 cycles behind
    // its partner, which then waits for it at the barrier (profiles/r05_stamps_kl_cfg4.txt); a static priority for those waves: config 4
    // W phase 530.5 -> 526.5 us, H phase 451 -> 446 (same box, bit-identical results).  (The Euclidean kernels showed nothing: r2.)
    if (KL) { if (yrole) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    // (the same on the Euclidean forms, measured again in r5: config 5 W phase 2788 -> 2799 us, config 2 117.8 -> 120.7: not there)
    if (KLONE) {
        for (int grp = g0; grp < g1; ++grp) kl128_group(grp, P);
        osum += 0.69314718055994531 * olog;
    } else if (KL) {
        if (g0 < g1) {
            const std::true_type yes; const std::false_type no;
            kl_iter(g0, P, Q, yes, no);
            int grp = g0 + 1;
            for (; grp + 1 < g1; grp += 2) { kl_iter(grp, Q, P, yes, yes); kl_iter(grp + 1, P, Q, yes, yes); }
            if (grp < g1) { kl_iter(grp, Q, P, yes, yes); kl_iter(grp + 1, P, Q, no, yes); }
            else kl_iter(grp, Q, P, no, yes);
            osum += 0.69314718055994531 * olog;
        }
    } else {
        for (int grp = g0; grp < g1; grp += 2) {
            group(grp, P, Q);
            if (grp + 1 < g1) group(grp + 1, Q, P);
        }
    }
    if constexpr (VAUX) { if (g0 < g1) vaux_store(g1 - 1); }
    if (PIPE && g0 < g1) {                             // the last group's carried stage and residual
        if ((g1 - g0) & 1) { carried_stage(P.d, false); residual(P, osum); } else { carried_stage(Q.d, false); residual(Q, osum); }
    }
#undef NMFX_FENCE
#ifdef NMFX_EXP_STAMPS
    {
        const int dbg_b = blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && dbg_b < 256) {
            unsigned long long* o = nmfx_dbg_stamps[WITH_OBJ][dbg_b][wave];
            o[0] = acc_wait; o[1] = acc_head; o[2] = acc_early; o[3] = acc_mfma;
            o[4] = __builtin_amdgcn_s_memtime() - tbeg; o[5] = __builtin_amdgcn_s_memrealtime() - rbeg;
        }
    }
#endif

    // ---- the two column halves of a row group exchange partial A tiles: wave (rg, hh) finishes the factor tiles NTP hh .. ----
    __syncthreads();                                   // everybody is done with the LDS tiles
    if constexpr (WITH_A) {
        float* xch = reinterpret_cast<float*>(smem);   // slot (rg, tile): [16 registers][64 lanes]
#pragma unroll
        for (int u = 0; u < NTP; ++u) {
            float* mine = xch + (rg * NT + NTP * (1 - hh) + u) * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = hh ? accA[u][r] : accA[NTP + u][r];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NTP; ++u) {
            const float* theirs = xch + (rg * NT + NTP * hh + u) * 1024 + lane;
            float* out = Apart + ((int64_t)sp * R + r0) * KP + 32 * (NTP * hh + u) + n31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float own = hh ? accA[NTP + u][r] : accA[u][r];
                out[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * b) * KP] = own + theirs[r * 64];
            }
        }
    }
    if (do_gram) {
        float* go = gram_part + ((int64_t)bx * S + sp) * KP * KP;
#pragma unroll
        for (int c = 0; c < NGT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                go[(int64_t)(16 * git + 4 * g + r) * KP + 16 * (gj0 + c) + x] = gacc[c][r];
    }
    if (WITH_OBJ && !VAUX) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) osum += __shfl_down(osum, off, 64);
        double* red = reinterpret_cast<double*>(smem + 4 * NT * 4096);      // behind the exchange slots
        if (lane == 0) red[wave] = osum;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NW; ++w) t += red[w];
            objpart[oidx] = ((KL || KLOBJ) ? 1.0 : 0.5) * t;
        }
        if (NPROB == 2) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) osum2 += __shfl_down(osum2, off, 64);
            if (lane == 0) red[8 + wave] = osum2;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int w = 0; w < 8; ++w) t += red[8 + w];
                objpart[((int64_t)gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = 0.5 * t;
            }
        }
    }
    if constexpr (SK) __syncthreads();                  // the exchange slots and `red` are read out before the next segment's tiles land
    }   // segments
#ifdef NMFX_EXP_BLOCKTIME
    if (SK && threadIdx.x == 0 && blockIdx.x < 1024) nmfx_dbg_times[WITH_OBJ][1][blockIdx.x] = wall_clock64();
#endif
}
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------
// out = in^T, stored TILE-MAJOR: tile (c / 128, r / 64) of [128][64] floats holds
// out[c][r] = in[r][c]; tiles of one 128-row block are consecutive (rows_in / 64 of them).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_tiled_kernel(const float* __restrict__ in, int64_t ldi,
                                                              float* __restrict__ out, int64_t rows_in)
{
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = in[(r0 + r) * ldi + c0 + tx];
    __syncthreads();
    float* dst = out + ((c0 >> 7) * (rows_in >> 6) + blockIdx.y) * 8192 + (c0 & 64) * 64;
    for (int c = ty; c < 64; c += 4) dst[c * 64 + tx] = tile[tx][c];
}

// out = in^T, both TILE-MAJOR (in: R x C as tiles [128][64], out: C x R as tiles [128][64]): the m x n state of the KL-loss ADMM
// variants changes orientation between the two sub-problems (r4)
// PRIV (r5): both sides in the register order of xyt32_bf16_kernel<..., VAUX> (dual_v): element (r, c) of a [128][64] tile at kl_dv_pos(r, c)
__device__ __forceinline__ int kl_dv_pos(int r, int c) {
    const int q = c >> 2;                              // 16-byte chunk of the row: q = 8 hh + 4 s + 2 b + e
    const int w = (r >> 5) + 4 * (q >> 3), j = ((q >> 2) & 1) * 2 + (q & 1), l = (r & 31) + 32 * ((q >> 1) & 1);
    return (((w * 4 + j) * 64 + l) << 2) + (c & 3);
}
template <bool PRIV>
__global__ __launch_bounds__(256) void tile_transpose_kernel(const float* __restrict__ in, int64_t R, int64_t C, float* __restrict__ out,
                                                              const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    if (PRIV) {
        // whole pieces on both sides: wave ty moves the four KiB pieces of the private-order wave w = (its row group of the half tile) + 4 (ty >> 1)
        const float4* src = reinterpret_cast<const float4*>(in + ((r0 >> 7) * (C >> 6) + blockIdx.x) * 8192);
        const int l31 = tx & 31, lb = tx >> 5, hq = ty >> 1, r = 32 * (ty & 1) + l31;
        {
            const int w = (int)((r0 & 64) >> 5) + (ty & 1) + 4 * hq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = src[(w * 4 + j) * 64 + tx];
                const int c = 4 * (8 * hq + 4 * (j >> 1) + 2 * lb + (j & 1));
                tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
            }
        }
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(out + ((c0 >> 7) * (R >> 6) + blockIdx.y) * 8192);
        const int w = (int)((c0 & 64) >> 5) + (ty & 1) + 4 * hq;                   // (r: row of the transposed half tile = column of `tile`)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * (8 * hq + 4 * (j >> 1) + 2 * lb + (j & 1));
            dst[(w * 4 + j) * 64 + tx] = make_float4(tile[c][r], tile[c + 1][r], tile[c + 2][r], tile[c + 3][r]);
        }
        return;
    }
    const float* src = in + ((r0 >> 7) * (C >> 6) + blockIdx.x) * 8192 + (r0 & 64) * 64;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = src[r * 64 + tx];
    __syncthreads();
    float* dst = out + ((c0 >> 7) * (R >> 6) + blockIdx.y) * 8192 + (c0 & 64) * 64;
    for (int c = ty; c < 64; c += 4) dst[c * 64 + tx] = tile[tx][c];
}

// out = in, stored TILE-MAJOR: tile (r / 128, c / 64) of [128][64] floats, the tiles of one
// 128-row block consecutive (ld / 64 of them).  One block per tile, 16-byte accesses.
__global__ __launch_bounds__(256) void retile_kernel(const float* __restrict__ in, int64_t ld, float* __restrict__ out)
{
    const int64_t rb = blockIdx.y, cb = blockIdx.x;
    const float* src = in + rb * 128 * ld + cb * 64;
    float4* dst = reinterpret_cast<float4*>(out + (rb * (ld >> 6) + cb) * 8192);
    for (int i = threadIdx.x; i < 128 * 16; i += 256)
        dst[i] = *reinterpret_cast<const float4*>(src + (int64_t)(i >> 4) * ld + 4 * (i & 15));
}

// the inverse of retile_kernel: row-major out from the tile-major copy
__global__ __launch_bounds__(256) void untile_kernel(const float* __restrict__ in, int64_t ld, float* __restrict__ out)
{
    const int64_t rb = blockIdx.y, cb = blockIdx.x;
    float* dst = out + rb * 128 * ld + cb * 64;
    const float4* src = reinterpret_cast<const float4*>(in + (rb * (ld >> 6) + cb) * 8192);
    for (int i = threadIdx.x; i < 128 * 16; i += 256)
        *reinterpret_cast<float4*>(dst + (int64_t)(i >> 4) * ld + 4 * (i & 15)) = src[i];
}

int nmfx_need_v(nmfx_engine* E) {
    if (E->V) return NMFX_OK;
    if (!E->Vtile) { E->err = "no copy of V on the device"; return NMFX_E_STATE; }
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->V), (size_t)E->mp * E->np * sizeof(float)));
    hipLaunchKernelGGL(untile_kernel, dim3((unsigned)(E->np / 64), (unsigned)(E->mp / 128)), dim3(256), 0, E->stream,
                       E->Vtile, E->np, E->V);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// bf16 hi/lo images of M [rows][cols] (row-major, ld) and of its transpose [cols][rows]
__global__ __launch_bounds__(256) void split_images_kernel(
    const float* __restrict__ M, int64_t rows, int64_t cols, int64_t ld,
    unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
    unsigned short* __restrict__ thi, unsigned short* __restrict__ tlo)
{
    __shared__ unsigned short sh[64][66], sl[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    for (int r = ty; r < 64; r += 4) {
        const float v = M[(r0 + r) * ld + c0 + tx];
        unsigned h, l;
        split2(v, 0.f, h, l);
        hi[(r0 + r) * ld + c0 + tx] = (unsigned short)h;
        lo[(r0 + r) * ld + c0 + tx] = (unsigned short)l;
        sh[r][tx] = (unsigned short)h; sl[r][tx] = (unsigned short)l;
    }
    __syncthreads();
    if (!thi) return;                                  // (block-uniform) no transposed image wanted
    for (int c = ty; c < 64; c += 4) {
        thi[(c0 + c) * rows + r0 + tx] = sh[tx][c];
        tlo[(c0 + c) * rows + r0 + tx] = sl[tx][c];
    }
}

// MUR-KL epilogues with the images and the sums the next products need (end of r2: the update, the image pass and the
// first stage of the column / row sums were three launches per factor -- 55 us of small kernels in a 1.1 ms iteration of
// config 4).  One 64 x 64 tile per block, as in split_images_kernel; the arithmetic of the update is that of
// kl_w_update_kernel / kl_h_update_kernel (kernels_kl.hip), expression for expression.
//   W (mur.py:26-27):  a = W * sum of the A slabs, b = rowsum_H[factor];  W_new = 2a / (b + sqrt(b^2 + 4 lam a));
//                      part[tile row][factor] = column sums of the tile (d = W^T 1 of the H update, finished by col_sums_final)
__global__ __launch_bounds__(256) void kl_w_epilogue_kernel(
    const float* __restrict__ Apart, int wsplit, int64_t count, int kp, int k, const float* __restrict__ Wold,
    const float* __restrict__ rowsum, float lam, float* __restrict__ Wnew, int64_t mp,
    unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, unsigned short* __restrict__ thi,
    unsigned short* __restrict__ tlo, float* __restrict__ part, const int* __restrict__ flag)
{
    if (*flag) return;
    __shared__ unsigned short sh[64][66], sl[64][66];
    __shared__ float cs[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int f = (int)c0 + tx;
    const float b = f < k ? rowsum[f] : 0.f;
    float colacc = 0.f;
    float sv[16], wv[16];                              // every load of the tile in flight before the first store
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int64_t i = (r0 + ty + 4 * u) * kp + f;
        float s = 0.f;
        if (f < k) {
            s = Apart[i];
            for (int p = 1; p < wsplit; ++p) s += Apart[(int64_t)p * count + i];
        }
        sv[u] = s; wv[u] = f < k ? Wold[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = ty + 4 * u;
        const int64_t i = (r0 + r) * kp + f;
        float v = 0.f;                                 // padded factors: 0/0 must stay out
        if (f < k) {
            const float a = wv[u] * sv[u];
            v = 2.f * a / (b + sqrtf(b * b + 4.f * lam * a));
        }
        Wnew[i] = v;
        colacc += v;
        unsigned h, l;
        split2(v, 0.f, h, l);
        hi[i] = (unsigned short)h; lo[i] = (unsigned short)l;
        sh[r][tx] = (unsigned short)h; sl[r][tx] = (unsigned short)l;
    }
    cs[ty][tx] = colacc;
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        thi[(c0 + c) * mp + r0 + tx] = sh[tx][c];
        tlo[(c0 + c) * mp + r0 + tx] = sl[tx][c];
    }
    if (ty == 0) part[(int64_t)blockIdx.y * kp + f] = (cs[0][tx] + cs[1][tx]) + (cs[2][tx] + cs[3][tx]);
}

//   H (mur.py:41-43):  c = H * B, d = colsum_W[factor];  H_new = 2c / (d + sqrt(d^2 + 4 lam c)), after the objective
//                      bookkeeping / convergence test (same protocol as MUR-eu); xf32 = [B (kp x np) | d];
//                      part[tile column][factor] = row sums of the tile (b = 1 H^T of the next W update)
__global__ __launch_bounds__(256) void kl_h_epilogue_kernel(
    const float* __restrict__ xf32, const double* __restrict__ xf64, float* __restrict__ H, int64_t np, int kp, int k,
    float lam, long long j, long long min_iter, double tol1, double tol2, DevState* __restrict__ st,
    double* __restrict__ obj_hist, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
    unsigned short* __restrict__ thi, unsigned short* __restrict__ tlo, float* __restrict__ part)
{
    if (st->flag) return;
    const int rule = nmfx_record_objective(st, obj_hist, xf64[0], j, min_iter, tol1, tol2,
                                           blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
    if (rule) return;
    __shared__ unsigned short sh[64][66], sl[64][66];
    __shared__ float rsum[64][4];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;      // factor rows, columns
    const int64_t count = (int64_t)kp * np;
    float hv[16], bv[16], dv[16];                      // every load of the tile in flight before the first store to H
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = ty + 4 * u;
        const int64_t i = (r0 + r) * np + c0 + tx;
        const bool live = r0 + r < k;
        hv[u] = live ? H[i] : 0.f; bv[u] = live ? xf32[i] : 0.f; dv[u] = live ? xf32[count + r0 + r] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = ty + 4 * u;
        const int64_t i = (r0 + r) * np + c0 + tx;
        float v = 0.f;                                 // padded factor rows stay zero
        if (r0 + r < k) {
            const float d = dv[u];
            const float c = hv[u] * bv[u];
            v = 2.f * c / (d + sqrtf(d * d + 4.f * lam * c));
            H[i] = v;
        }
        unsigned h, l;
        split2(v, 0.f, h, l);
        hi[i] = (unsigned short)h; lo[i] = (unsigned short)l;
        sh[r][tx] = (unsigned short)h; sl[r][tx] = (unsigned short)l;
        float t = v;                                   // row sum over the tile's 64 columns: one wave holds the row
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (tx == 0) rsum[r][0] = t;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        thi[(c0 + c) * kp + r0 + tx] = sh[tx][c];
        tlo[(c0 + c) * kp + r0 + tx] = sl[tx][c];
    }
    if (ty == 0) part[(int64_t)blockIdx.x * kp + r0 + tx] = rsum[tx][0];
}

// ---------------------------------------------------------------------------
// W epilogue (nmf/mur.py:29): W_new = W * A / (W (H H^T) + lam W + 1e-9), plus the bf16 images
// of the new W: row-major (next iteration's residual) and transposed (this iteration's H
// phase).  Block = 64 rows; the 64 x 64 x 64 product W (H H^T) runs on the f32 MFMA.
// ---------------------------------------------------------------------------
#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef __attribute__((address_space(3))) float lds_f32;      // (a volatile access through a generic pointer would be a flat_ instruction)
__device__ __forceinline__ volatile lds_f32* lds_volatile(void* p) {
    return (volatile lds_f32*)(__attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void add4(float* acc, const float* src) {   // acc[0..3] += one 16-byte load
    const float4 t = *reinterpret_cast<const float4*>(src);
    acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
}
__device__ __forceinline__ uint4 pack8(const unsigned short* p) {       // 8 consecutive bf16 from LDS
    return make_uint4(p[0] | ((unsigned)p[1] << 16), p[2] | ((unsigned)p[3] << 16),
                      p[4] | ((unsigned)p[5] << 16), p[6] | ((unsigned)p[7] << 16));
}

// PAIR (KP = 128): two independent problems in the factor halves [0, 64) and [64, 128) -- H H^T is taken block diagonal, lambda is
// per half (lam, lam1), and a problem whose stop rule has fired keeps the iterate the reference returns: that one lives in the
// W buffer (stop_i + 1) & 1, so launches that would write THAT buffer (`nxt`) leave the problem's half alone.
template <int KP, bool PAIR = false>
__global__ __launch_bounds__(512) void mur_w_update_bf16_kernel(
    const float* __restrict__ Apart, int wsplit, int64_t mp, const float* __restrict__ Wold,
    const float* __restrict__ HHtpart, int hslabs, float lam, float* __restrict__ Wnew,
    unsigned short* __restrict__ Whi, unsigned short* __restrict__ Wlo,
    unsigned short* __restrict__ WThi, unsigned short* __restrict__ WTlo, const int* __restrict__ flag,
    float lam1 = 0.f, const DevState* __restrict__ st = nullptr, int nxt = 0)
{
    static_assert(!PAIR || KP == 128, "pair mode stacks two k <= 64 problems into the k = 128 layouts");
    constexpr int RB = 64, LDW = KP + 4, LDH = KP + 16, NT = 512;   // padded LDS rows: conflict-free dword reads
    constexpr int WV = KP / 32;                        // 16-byte pieces of the 64 x KP W tile per thread
    constexpr int HV = KP * KP / 2048;                 // ... of one KP x KP Gram slab per thread
    constexpr int HCH = 32 / HV;                       // Gram slabs requested together (32 pieces per thread)
    constexpr int EP = KP / 8;                         // outputs per thread: row tid / 8, factors EP (tid % 8) ..
    constexpr int CT = KP / 32;                        // column tiles of the product per wave
    extern __shared__ __attribute__((aligned(16))) float wdyn[];
    float* hs = wdyn;                                  // H H^T [KP][LDH], later the product tile D [row][LDW]
    float* ws = hs + KP * LDH;                         // W tile [RB][LDW]
    unsigned short* th = reinterpret_cast<unsigned short*>(ws + RB * LDW);   // [KP][RB + 2] transposed hi image
    unsigned short* tl = th + KP * (RB + 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * RB;
    // The kernel is a chain of dependent memory round trips, so it is built to have as few as possible: 512
    // threads, and everything is requested before the first wait -- the stop flag, the W tile, the A slabs
    // of this thread's outputs four slabs at a time, and the H H^T slabs HCH at a time.
    const int stop = *flag;
    bool keep[2] = {false, false};                     // (PAIR) halves whose final iterate sits in the buffer this launch writes
    if (PAIR) {
#pragma unroll
        for (int p = 0; p < 2; ++p) keep[p] = st->pflag[p] != 0 && (int)((st->pstop_i[p] + 1) & 1) == nxt;
    }
    const int erow = tid >> 3, ej0 = EP * (tid & 7);
    const int64_t eidx = (r0 + erow) * KP + ej0;
    float4 wt[WV];
#pragma unroll
    for (int u = 0; u < WV; ++u) wt[u] = reinterpret_cast<const float4*>(Wold + r0 * KP)[tid + NT * u];
    float a[EP] = {};
    float v[HV][4] = {};
    for (int p0 = 0, h0 = 0; p0 < wsplit || h0 < hslabs; p0 += 4, h0 += HCH) {
        float4 ta[4][EP / 4], tv[HCH][HV];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int v4 = 0; v4 < EP / 4; ++v4)
                ta[pp][v4] = (p0 + pp < wsplit)
                    ? *reinterpret_cast<const float4*>(Apart + (int64_t)(p0 + pp) * mp * KP + eidx + 4 * v4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pp = 0; pp < HCH; ++pp)
#pragma unroll
            for (int u = 0; u < HV; ++u)
                tv[pp][u] = (h0 + pp < hslabs)
                    ? reinterpret_cast<const float4*>(HHtpart + (int64_t)(h0 + pp) * KP * KP)[tid + NT * u]
                    : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int v4 = 0; v4 < EP / 4; ++v4) {
                a[4 * v4] += ta[pp][v4].x; a[4 * v4 + 1] += ta[pp][v4].y; a[4 * v4 + 2] += ta[pp][v4].z; a[4 * v4 + 3] += ta[pp][v4].w;
            }
#pragma unroll
        for (int pp = 0; pp < HCH; ++pp)
#pragma unroll
            for (int u = 0; u < HV; ++u) {
                v[u][0] += tv[pp][u].x; v[u][1] += tv[pp][u].y; v[u][2] += tv[pp][u].z; v[u][3] += tv[pp][u].w;
            }
    }
    if (stop) return;
#pragma unroll
    for (int u = 0; u < HV; ++u) {
        const int i = tid + NT * u;
        const bool off = PAIR && ((i / (KP / 4)) < KP / 2) != ((i % (KP / 4)) < KP / 8);      // (off-diagonal block: the other problem's factors)
        *reinterpret_cast<float4*>(hs + (i / (KP / 4)) * LDH + 4 * (i % (KP / 4))) =
            off ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
    }
#pragma unroll
    for (int u = 0; u < WV; ++u) {
        const int i = tid + NT * u;
        *reinterpret_cast<float4*>(ws + (i / (KP / 4)) * LDW + 4 * (i % (KP / 4))) = wt[u];
    }
    __syncthreads();
    // D[row][j] = sum_l W[row][l] HHt[l][j]; wave = 16 rows (wave & 3), CT column tiles (half wave >> 2)
    const int rt = wave & 3, ch = wave >> 2;
    f32x4 acc[CT];
#pragma unroll
    for (int jt = 0; jt < CT; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int u = 0; u < KP / 4; ++u) {
        const float av = ws[(16 * rt + x) * LDW + 4 * u + q];
#pragma unroll
        for (int jt = 0; jt < CT; ++jt) acc[jt] = MFMA_F32(av, hs[(4 * u + q) * LDH + 16 * (CT * ch + jt) + x], acc[jt]);
    }
    __syncthreads();                                   // everybody is done with H H^T
    // volatile: hipcc 7.2's DS store merging (ds_write2_b32) mis-encodes offset0 for these
    // strided stores (17/34/51 dwords instead of 68/136/204); volatile keeps them single.
    volatile lds_f32* dt = lds_volatile(hs);           // D tile [row][LDW]
#pragma unroll
    for (int jt = 0; jt < CT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dt[(16 * rt + 4 * q + r) * LDW + 16 * (CT * ch + jt) + x] = acc[jt][r];
    __syncthreads();
    {   // epilogue, vectorised along the factor index: thread = (row, EP consecutive j)
        float wn[EP];
        unsigned ph[EP / 2], pl[EP / 2];
        const float lam_t = (PAIR && ej0 >= KP / 2) ? lam1 : lam;
        const bool skip = PAIR && keep[ej0 >= KP / 2 ? 1 : 0];
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const float w = ws[erow * LDW + ej0 + e];
            wn[e] = w * a[e] / (dt[erow * LDW + ej0 + e] + lam_t * w + 1e-9f);
        }
#pragma unroll
        for (int e = 0; e < EP / 2; ++e) {
            split2(wn[2 * e], wn[2 * e + 1], ph[e], pl[e]);
            th[(ej0 + 2 * e) * (RB + 2) + erow] = (unsigned short)(ph[e] & 0xffffu); th[(ej0 + 2 * e + 1) * (RB + 2) + erow] = (unsigned short)(ph[e] >> 16);
            tl[(ej0 + 2 * e) * (RB + 2) + erow] = (unsigned short)(pl[e] & 0xffffu); tl[(ej0 + 2 * e + 1) * (RB + 2) + erow] = (unsigned short)(pl[e] >> 16);
        }
        if (!skip) {
#pragma unroll
        for (int v4 = 0; v4 < EP / 4; ++v4)
            *reinterpret_cast<float4*>(Wnew + eidx + 4 * v4) = make_float4(wn[4 * v4], wn[4 * v4 + 1], wn[4 * v4 + 2], wn[4 * v4 + 3]);
#pragma unroll
        for (int v8 = 0; v8 < EP / 8; ++v8) {
            *reinterpret_cast<uint4*>(Whi + eidx + 8 * v8) = make_uint4(ph[4 * v8], ph[4 * v8 + 1], ph[4 * v8 + 2], ph[4 * v8 + 3]);
            *reinterpret_cast<uint4*>(Wlo + eidx + 8 * v8) = make_uint4(pl[4 * v8], pl[4 * v8 + 1], pl[4 * v8 + 2], pl[4 * v8 + 3]);
        }
        }
    }
    __syncthreads();
    {   // transposed images: thread (f = tid / 8 (+ 64), eighth = tid % 8) stores 8 rows = 16 bytes per image
        // (PAIR: a stopped problem's transposed images are no longer read for anything that is kept -- written as they come)
        const int oc = tid & 7;
#pragma unroll
        for (int f = tid >> 3; f < KP; f += 64) {
            *reinterpret_cast<uint4*>(WThi + (int64_t)f * mp + r0 + 8 * oc) = pack8(th + f * (RB + 2) + 8 * oc);
            *reinterpret_cast<uint4*>(WTlo + (int64_t)f * mp + r0 + 8 * oc) = pack8(tl + f * (RB + 2) + 8 * oc);
        }
    }
}

// H epilogue (nmf/mur.py:45): H_new = H * B / (G H + lam H + 1e-9) from B^T = V^T W (stored
// [np][KP]) + objective bookkeeping + the bf16 images of the new H ([KP][np]; the W phase takes
// its transposed fragments from the same image).  Block = 64 columns; the product G H runs on the f32
// MFMA and every global access is a 16-byte vector (tiles are turned through LDS).
// FROM_SLABS: single-GPU runs read the split slabs of the H phase, the Gram slabs and the
// objective partials directly (no `pack` launch); sharded runs read the all-reduced
// exchange buffers.
// PAIR (KP = 128, not FROM_SLABS): two problems in the factor halves -- W^T W block diagonal, lambda per half (lam, lam1), the two
// objectives summed here from the product kernel's per-block partials osrc[p][nobj], one stop test and history each
// (nmfx_record_objective_pair); the rows of a problem that has stopped are left as they are.
template <int KP, bool FROM_SLABS, bool PAIR = false>
__global__ __launch_bounds__(512) void mur_h_update_bf16_kernel(
    const float* __restrict__ bsrc, int bsplit, const float* __restrict__ gsrc, int gsplit,
    const double* __restrict__ osrc, int64_t nobj, float* __restrict__ H,
    int64_t np, float lam, long long j, long long min_iter, double tol1, double tol2,
    DevState* __restrict__ st, double* __restrict__ obj_hist,
    unsigned short* __restrict__ Hhi, unsigned short* __restrict__ Hlo,
    const float* __restrict__ xtail = nullptr, int xworld = 0,   // NMFX_XTAIL: every rank's objective partial, as 16-bit digits
    float lam1 = 0.f,
    int cb0 = 0, float* __restrict__ stage = nullptr)            // reduce-scatter / all-gather exchange (r5): the grid covers the column blocks
{                                                                // from cb0 on (this rank's slice) and ALSO leaves the new KP x 64 tile, as it stands
                                                                 // ([factor][64] f32), where its B^T tile came from: stage + c0 * KP -- the all-gather's send range
    static_assert(!PAIR || (KP == 128 && !FROM_SLABS), "pair mode: k = 128 layouts, sums from the pack launch");
    constexpr int CB = 64, LDG = KP + 4, LDC = 80, LDD = 68, NT = 512;
    constexpr int TV = KP / 32;                        // 16-byte pieces per thread of the KP x 64 H tile / the 64 x KP B^T tile
    constexpr int GV = KP * KP / 2048;                 // ... of one KP x KP Gram slab
    constexpr int SCH = 16 / GV;                       // slab indices requested together (<= 16 + 2 SCH pieces per thread)
    constexpr int RTN = KP / 16;                       // row tiles of the product; wave -> (row tile, group of column tiles)
    constexpr int CT = 4 * RTN / 8;                    // column tiles per wave
    constexpr int EP = KP / 8, TPR = CB / EP;          // outputs per thread, threads per factor row
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* gs = dyn;                                   // G [j][LDG], later the product tile D [j][LDD]
    float* hs = gs + KP * LDG;                         // H tile [j][LDC]
    float* bt = hs + KP * LDC;                         // B^T tile [c][LDG]
    __shared__ double shd[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int64_t c0 = (int64_t)(blockIdx.x + cb0) * CB;
    const int64_t kk = (int64_t)KP * KP, bn = (int64_t)KP * np;
    int dead[2] = {0, 0};                              // (PAIR) problems that stopped in an earlier iteration
    double sacc1 = 0.0;
    if (PAIR) { dead[0] = st->pflag[0]; dead[1] = st->pflag[1]; for (int64_t i = tid; i < nobj; i += NT) sacc1 += osrc[nobj + i]; }
    // The kernel is a chain of dependent memory round trips, so it is built to have as few as possible:
    // 512 threads, and everything that does not depend on a decision -- the stop flag, the objective
    // partials, the H tile, and the G / B^T slabs of SCH slab indices at a time -- is requested before the
    // first wait.  (With 256 threads and two slabs per round the 16 Gram slabs of config 2 were 8 round
    // trips: 17.5 us; now 13.)
    const int stop = st->flag;
    double sacc = 0.0;
    if (FROM_SLABS || PAIR) { for (int64_t i = tid; i < nobj; i += NT) sacc += osrc[i]; }
    else if (xtail) {                                  // the partials of all ranks, exact, summed in rank order (the same on every rank)
        for (int r = 0; r < xworld; ++r) {
            const float4 dg = *reinterpret_cast<const float4*>(xtail + 4 * r);
            const unsigned long long bits = (unsigned long long)dg.x | ((unsigned long long)dg.y << 16) |
                                            ((unsigned long long)dg.z << 32) | ((unsigned long long)dg.w << 48);
            sacc += __longlong_as_double((long long)bits);
        }
    }
    else sacc = osrc[0];
    float4 ht[TV];
#pragma unroll
    for (int u = 0; u < TV; ++u) {
        const int i = tid + NT * u;
        ht[u] = *reinterpret_cast<const float4*>(H + (int64_t)(i >> 4) * np + c0 + 4 * (i & 15));
    }
    float g[GV][4] = {}, b[TV][4] = {};
    {   // G and the B^T tile: slab sums (fixed order)
        const int ng = FROM_SLABS ? gsplit : 1, nb = FROM_SLABS ? bsplit : 1;
        const int nslab = ng > nb ? ng : nb;
        auto chunk = [&](int p0) {
            float4 tg[SCH][GV], tb[SCH][TV];
#pragma unroll
            for (int pp = 0; pp < SCH; ++pp) {
                const bool pg = p0 + pp < ng, pb = p0 + pp < nb;
#pragma unroll
                for (int u = 0; u < GV; ++u)
                    tg[pp][u] = pg ? reinterpret_cast<const float4*>(gsrc + (int64_t)(p0 + pp) * kk)[tid + NT * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < TV; ++u) {
                    const int i = tid + NT * u;
                    tb[pp][u] = pb ? *reinterpret_cast<const float4*>(bsrc + (int64_t)(p0 + pp) * bn + (c0 + i / (KP / 4)) * KP + 4 * (i % (KP / 4)))
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int pp = 0; pp < SCH; ++pp) {
#pragma unroll
                for (int u = 0; u < GV; ++u) {
                    g[u][0] += tg[pp][u].x; g[u][1] += tg[pp][u].y; g[u][2] += tg[pp][u].z; g[u][3] += tg[pp][u].w;
                }
#pragma unroll
                for (int u = 0; u < TV; ++u) {
                    b[u][0] += tb[pp][u].x; b[u][1] += tb[pp][u].y; b[u][2] += tb[pp][u].z; b[u][3] += tb[pp][u].w;
                }
            }
        };
        chunk(0);
        // the H tile goes to LDS as soon as the first round trip is over (it was requested in front of it): held in registers
        // across ALL slab chunks it did not fit next to their 32 in-flight float4 and hipcc parked it in scratch (48 B / lane)
#pragma unroll
        for (int u = 0; u < TV; ++u) {
            const int i = tid + NT * u;
            *reinterpret_cast<float4*>(hs + (i >> 4) * LDC + 4 * (i & 15)) = ht[u];
        }
        for (int p0 = SCH; p0 < nslab; p0 += SCH) chunk(p0);
    }
    if (stop) return;
    double obj;
    if (FROM_SLABS || PAIR) {                          // same fixed-order sum in every block
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sacc += __shfl_down(sacc, off, 64);
        if (lane == 0) shd[wave] = sacc;
        if (PAIR) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sacc1 += __shfl_down(sacc1, off, 64);
            if (lane == 0) shd[8 + wave] = sacc1;
        }
        __syncthreads();
        obj = ((((((shd[0] + shd[1]) + shd[2]) + shd[3]) + shd[4]) + shd[5]) + shd[6]) + shd[7];
    } else {
        obj = sacc;
    }
    if (PAIR) {
        const double obj1 = ((((((shd[8] + shd[9]) + shd[10]) + shd[11]) + shd[12]) + shd[13]) + shd[14]) + shd[15];
        const bool writer = blockIdx.x == 0 && tid == 0;
        if (!dead[0]) dead[0] = nmfx_record_objective_pair(st, obj_hist, obj, 0, j, min_iter, tol1, tol2, writer);
        if (!dead[1]) dead[1] = nmfx_record_objective_pair(st, obj_hist, obj1, 1, j, min_iter, tol1, tol2, writer);
        if (dead[0] && dead[1]) { if (writer) st->flag = 1; return; }       // both done: every later launch returns at once
    } else {
    const int rule = nmfx_record_objective(st, obj_hist, obj, j, min_iter, tol1, tol2,
                                           blockIdx.x == 0 && tid == 0);
    if (rule) return;
    }
#pragma unroll
    for (int u = 0; u < TV; ++u) {
        const int i = tid + NT * u;
        *reinterpret_cast<float4*>(bt + (i / (KP / 4)) * LDG + 4 * (i % (KP / 4))) = make_float4(b[u][0], b[u][1], b[u][2], b[u][3]);
    }
#pragma unroll
    for (int u = 0; u < GV; ++u) {
        const int i = tid + NT * u;
        const bool off = PAIR && ((i / (KP / 4)) < KP / 2) != ((i % (KP / 4)) < KP / 8);      // (the other problem's factors)
        *reinterpret_cast<float4*>(gs + (i / (KP / 4)) * LDG + 4 * (i % (KP / 4))) =
            off ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(g[u][0], g[u][1], g[u][2], g[u][3]);
    }
    __syncthreads();
    // D[jrow][c] = sum_l G[jrow][l] H[l][c]; wave = 16 factor rows (tile wave % RTN), CT column tiles
    const int rt = wave % RTN, cb = (wave / RTN) * CT;
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int u = 0; u < KP / 4; ++u) {
        const float a = gs[(16 * rt + x) * LDG + 4 * u + q];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = MFMA_F32(a, hs[(4 * u + q) * LDC + 16 * (cb + ct) + x], acc[ct]);
    }
    __syncthreads();                                   // everybody is done with G
    volatile lds_f32* dt = lds_volatile(gs);           // D tile [j][LDD]; volatile: see mur_w_update_bf16_kernel
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) dt[(16 * rt + 4 * q + r) * LDD + 16 * (cb + ct) + x] = acc[ct][r];
    __syncthreads();
    {   // epilogue, vectorised along the columns: thread = (factor row jr, EP consecutive c)
        const int jr = tid / TPR, cq = EP * (tid % TPR);
        float hn[EP];
        unsigned ph[EP / 2], pl[EP / 2];
        const float lam_t = (PAIR && jr >= KP / 2) ? lam1 : lam;
        const bool skip = PAIR && dead[jr >= KP / 2 ? 1 : 0] != 0;
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const float h = hs[jr * LDC + cq + e];
            hn[e] = h * bt[(cq + e) * LDG + jr] / (dt[jr * LDD + cq + e] + lam_t * h + 1e-9f);
        }
#pragma unroll
        for (int e = 0; e < EP / 2; ++e) split2(hn[2 * e], hn[2 * e + 1], ph[e], pl[e]);
        const int64_t idx = (int64_t)jr * np + c0 + cq;
        if (!skip) {
#pragma unroll
        for (int v4 = 0; v4 < EP / 4; ++v4)
            *reinterpret_cast<float4*>(H + idx + 4 * v4) = make_float4(hn[4 * v4], hn[4 * v4 + 1], hn[4 * v4 + 2], hn[4 * v4 + 3]);
#pragma unroll
        for (int v8 = 0; v8 < EP / 8; ++v8) {
            *reinterpret_cast<uint4*>(Hhi + idx + 8 * v8) = make_uint4(ph[4 * v8], ph[4 * v8 + 1], ph[4 * v8 + 2], ph[4 * v8 + 3]);
            *reinterpret_cast<uint4*>(Hlo + idx + 8 * v8) = make_uint4(pl[4 * v8], pl[4 * v8 + 1], pl[4 * v8 + 2], pl[4 * v8 + 3]);
        }
        if (stage) {                                   // (every thread of the block has its B^T values in LDS since two barriers ago)
#pragma unroll
            for (int v4 = 0; v4 < EP / 4; ++v4)
                *reinterpret_cast<float4*>(stage + c0 * KP + jr * CB + cq + 4 * v4) = make_float4(hn[4 * v4], hn[4 * v4 + 1], hn[4 * v4 + 2], hn[4 * v4 + 3]);
        }
        }
    }
}

// The other ranks' columns of the new H after the all-gather of the reduce-scatter / all-gather exchange: the KP x 64 tiles the
// ranks' mur_h_update_bf16_kernel launches left in the exchange buffer (stage + c0 * KP, [factor][64] f32) go to H and its bf16
// images -- the values, the split and hence the bits the owning rank wrote for itself.  The grid skips the column blocks
// [skip0, skip1) (this rank's own slice).
template <int KP>
__global__ __launch_bounds__(512) void mur_h_unpack_bf16_kernel(
    const float* __restrict__ stage, float* __restrict__ H, unsigned short* __restrict__ Hhi, unsigned short* __restrict__ Hlo,
    int64_t np, int skip0, int skip1, const int* __restrict__ flag)
{
    if (*flag) return;                                 // (the stop rule fired in the epilogue launch in front: H stays as it is)
    constexpr int CB = 64, EP = KP / 8, TPR = CB / EP;
    const int tid = threadIdx.x;
    const int cbk = (int)blockIdx.x < skip0 ? (int)blockIdx.x : (int)blockIdx.x + (skip1 - skip0);
    const int64_t c0 = (int64_t)cbk * CB;
    const int jr = tid / TPR, cq = EP * (tid % TPR);
    float hn[EP];
    unsigned ph[EP / 2], pl[EP / 2];
#pragma unroll
    for (int v4 = 0; v4 < EP / 4; ++v4) {
        const float4 t = *reinterpret_cast<const float4*>(stage + c0 * KP + jr * CB + cq + 4 * v4);
        hn[4 * v4] = t.x; hn[4 * v4 + 1] = t.y; hn[4 * v4 + 2] = t.z; hn[4 * v4 + 3] = t.w;
    }
#pragma unroll
    for (int e = 0; e < EP / 2; ++e) split2(hn[2 * e], hn[2 * e + 1], ph[e], pl[e]);
    const int64_t idx = (int64_t)jr * np + c0 + cq;
#pragma unroll
    for (int v4 = 0; v4 < EP / 4; ++v4)
        *reinterpret_cast<float4*>(H + idx + 4 * v4) = make_float4(hn[4 * v4], hn[4 * v4 + 1], hn[4 * v4 + 2], hn[4 * v4 + 3]);
#pragma unroll
    for (int v8 = 0; v8 < EP / 8; ++v8) {
        *reinterpret_cast<uint4*>(Hhi + idx + 8 * v8) = make_uint4(ph[4 * v8], ph[4 * v8 + 1], ph[4 * v8 + 2], ph[4 * v8 + 3]);
        *reinterpret_cast<uint4*>(Hlo + idx + 8 * v8) = make_uint4(pl[4 * v8], pl[4 * v8 + 1], pl[4 * v8 + 2], pl[4 * v8 + 3]);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int lazy_alloc(nmfx_engine* E, T** p, int64_t count) {
    if (*p) return NMFX_OK;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T)));
    NMFX_HIP(hipMemsetAsync(*p, 0, (size_t)count * sizeof(T), E->stream));
    return NMFX_OK;
}

bool nmfx_bf16_supported(const nmfx_engine* E) { return (E->kp == 64 || E->kp == 128) && E->mp % 128 == 0 && E->np % 128 == 0; }

template <int KP, bool OBJ, bool KL, int TERMS, bool WITH_A = true>
static int launch_xyt_t(nmfx_engine* E, const float* X, bool tiled, int64_t ldx, int64_t R, int ngroups, int splits,
                        const unsigned short* Yhi, const unsigned short* Ylo, int64_t ldy, const unsigned short* Zhi,
                        const unsigned short* Zlo, float* Apart, float* gram_part, int ng) {
    dim3 grid((unsigned)(R / 128), (unsigned)splits), block(512);
    const size_t shm = 160 * 1024;                                       // Y double buffer + V rings
    auto kern = xyt_bf16_kernel<KP, OBJ, KL, TERMS, WITH_A>;
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, grid, block, shm, E->stream, X, ldx, Yhi, Ylo, ldy, Zhi, Zlo, Apart, E->obj_part,
                       gram_part, R, ngroups, &E->state->flag, tiled ? 1 : 0, ng);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

static bool nmfx_kl_nw4() {
#ifdef NMFX_EXP_KLNW4
    static const bool on = getenv("NMFX_KL_NW4") && atoi(getenv("NMFX_KL_NW4")) == 1;
    return on;
#else
    return false;
#endif
}
template <bool OBJ, int TERMS, bool KL = false, int KP = 64, bool WITH_A = true>
static int launch_xyt32_t(nmfx_engine* E, const float* X, int64_t ldx, int64_t R, int ngroups, int splits,
                          const unsigned short* Yhi, const unsigned short* Ylo, int64_t ldy, const unsigned short* Zhi,
                          const unsigned short* Zlo, float* Apart, float* gram_part, int ng) {
    dim3 grid((unsigned)(R / 128), (unsigned)splits), block(512);
    size_t shm = 160 * 1024;
    auto kern = xyt32_bf16_kernel<OBJ, TERMS, 0, KL, KP, 1, false, WITH_A>;
    if constexpr (!KL && KP == 64 && TERMS == 3 && WITH_A) {
        if (E->xyt_nw == 4) {      // four-wave blocks of 64 rows, two resident per CU (see the kernel's NW note)
            grid = dim3((unsigned)(R / 64), (unsigned)splits); block = dim3(256); shm = 80 * 1024;
            static const int forced = getenv("NMFX_TEMPORAL") ? atoi(getenv("NMFX_TEMPORAL")) : -1;
            const bool small = 2.0 * (double)E->mp * (double)E->np * 4.0 <= 192.0 * 1024 * 1024;
            auto k4 = (forced == 1 || (forced < 0 && small)) ? xyt32_bf16_kernel<OBJ, TERMS, 0, KL, KP, 1, true, WITH_A, 4>
                                                             : xyt32_bf16_kernel<OBJ, TERMS, 0, KL, KP, 1, false, WITH_A, 4>;
            int rc4 = nmfx_allow_lds(E, reinterpret_cast<const void*>(k4), (int)shm); if (rc4) return rc4;
            hipLaunchKernelGGL(k4, grid, block, shm, E->stream, X, ldx, Yhi, Ylo, ldy, Zhi, Zlo, Apart, E->obj_part,
                               gram_part, R, ngroups, &E->state->flag, ng, (const int4*)nullptr, (const int*)nullptr, 0, XytSide(), (float*)nullptr, (float*)nullptr, E->xyt_flag2, E->xyt_xpriv);
            NMFX_HIP(hipGetLastError());
            return NMFX_OK;
        }
    }
#ifdef NMFX_EXP_KLNW4          // experiment (r5): MUR-KL at k padded to 64 in FOUR-wave blocks of 64 rows, two per CU, in the one-register-set
                               // form (80 KiB of LDS each: Y double buffer + V ring of 3) -- the two waves of a SIMD then belong to different workgroups, no
                               // barrier ties them, one's VALU sections can meet the other's MFMAs.  Correct (the KL tests pass with NMFX_KL_NW4=1) and
                               // SLOWER than the eight-wave cross-group pipeline: config 4 W phase 538 -> 588 us, H phase 414-430 -> 474-479
                               // (profiles/r05_kl_four_wave_blocks_experiment.txt).  Not built by default.
    if constexpr (KL && KP == 64 && TERMS == 3 && WITH_A) {
        if (nmfx_kl_nw4()) {
            grid = dim3((unsigned)(R / 64), (unsigned)splits); block = dim3(256); shm = 80 * 1024;
            auto k4 = xyt32_bf16_kernel<OBJ, TERMS, 0, KL, KP, 1, false, WITH_A, 4>;
            int rc4 = nmfx_allow_lds(E, reinterpret_cast<const void*>(k4), (int)shm); if (rc4) return rc4;
            hipLaunchKernelGGL(k4, grid, block, shm, E->stream, X, ldx, Yhi, Ylo, ldy, Zhi, Zlo, Apart, E->obj_part,
                               gram_part, R, ngroups, &E->state->flag, ng, (const int4*)nullptr, (const int*)nullptr, 0, XytSide(), (float*)nullptr, (float*)nullptr, E->xyt_flag2, 0);
            NMFX_HIP(hipGetLastError());
            return NMFX_OK;
        }
    }
#endif
    if constexpr (!KL && KP == 64 && TERMS == 3) {
        // V and V^T together small enough to live in the Infinity Cache (see the kernel's TEMPORAL note; NMFX_TEMPORAL=0/1 overrides)
        static const int forced = getenv("NMFX_TEMPORAL") ? atoi(getenv("NMFX_TEMPORAL")) : -1;
        const bool small = 2.0 * (double)E->mp * (double)E->np * 4.0 <= 192.0 * 1024 * 1024;
        if (forced == 1 || (forced < 0 && small)) kern = xyt32_bf16_kernel<OBJ, TERMS, 0, KL, KP, 1, true, WITH_A>;
    }
    if constexpr (OBJ && !KL && KP == 128 && TERMS == 3 && WITH_A) {
        if (E->pair) kern = xyt32_bf16_kernel<OBJ, TERMS, 0, KL, KP, 2>;     // two stacked problems: one objective each
    }
#ifdef NMFX_EXP_ABLATE
    if constexpr (TERMS == 3 && KL && KP == 64) {
        static const int abl = getenv("NMFX_ABLATE") ? atoi(getenv("NMFX_ABLATE")) : 0;
        switch (abl) {
            case 1: kern = xyt32_bf16_kernel<OBJ, TERMS, 1, KL, KP>; break;
            case 16: kern = xyt32_bf16_kernel<OBJ, TERMS, 16, KL, KP>; break;
            case 17: kern = xyt32_bf16_kernel<OBJ, TERMS, 17, KL, KP>; break;
            default: break;
        }
    }
    if (OBJ && TERMS == 3 && !KL && KP == 64) {
        static const int abl = getenv("NMFX_ABLATE") ? atoi(getenv("NMFX_ABLATE")) : 0;
        switch (abl) {
            case 1: kern = xyt32_bf16_kernel<OBJ, TERMS, 1>; break;
            case 2: kern = xyt32_bf16_kernel<OBJ, TERMS, 2>; break;
            case 4: kern = xyt32_bf16_kernel<OBJ, TERMS, 4>; break;
            case 8: kern = xyt32_bf16_kernel<OBJ, TERMS, 8>; break;
            case 16: kern = xyt32_bf16_kernel<OBJ, TERMS, 16>; break;
            case 6: kern = xyt32_bf16_kernel<OBJ, TERMS, 6>; break;
            case 17: kern = xyt32_bf16_kernel<OBJ, TERMS, 17>; break;
            case 30: kern = xyt32_bf16_kernel<OBJ, TERMS, 30>; break;
            default: break;
        }
    }
#endif
    int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, block, shm, E->stream, X, ldx, Yhi, Ylo, ldy, Zhi, Zlo, Apart, E->obj_part,
                       gram_part, R, ngroups, &E->state->flag, ng, (const int4*)nullptr, (const int*)nullptr, 0, XytSide(), (float*)nullptr, (float*)nullptr, E->xyt_flag2, E->xyt_xpriv);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

static int launch_xyt(nmfx_engine* E, bool obj, const float* X, bool tiled, int64_t ldx, int64_t R, int ngroups, int splits,
                      const unsigned short* Yhi, const unsigned short* Ylo, int64_t ldy, const unsigned short* Zhi,
                      const unsigned short* Zlo, float* Apart, float* gram_part, const char* name, bool kl = false,
                      int ng = 1, int terms = 4) {
    static const bool force4 = getenv("NMFX_BF16_TERMS") && atoi(getenv("NMFX_BF16_TERMS")) == 4;
    if (force4) terms = 4;
    ProfScope ps(E, name);
    const bool nw4 = E->xyt_nw == 4 && E->kp == 64 && !kl && tiled && gram_part && Apart && terms == 3 &&
                     !(getenv("NMFX_XYT16") && atoi(getenv("NMFX_XYT16")) == 1);
    if (!nw4) E->xyt_nw = 8;
    const bool klnw4 = kl && tiled && E->kp == 64 && terms == 3 && nmfx_kl_nw4() && !(getenv("NMFX_XYT16") && atoi(getenv("NMFX_XYT16")) == 1);
    if (obj) E->obj_count = (R / ((nw4 || klnw4) ? 64 : 128)) * splits;
#define NMFX_XYT2(KP_, OBJ_, KL_, T_) \
    launch_xyt_t<KP_, OBJ_, KL_, T_>(E, X, tiled, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, Apart, gram_part, ng)
#define NMFX_XYT(KP_, OBJ_, KL_) (terms == 3 ? NMFX_XYT2(KP_, OBJ_, KL_, 3) : NMFX_XYT2(KP_, OBJ_, KL_, 4))
    if (!Apart) {                                      // objective only (Euclidean; three terms: nothing is fed back from it)
        if (!obj || kl) { E->err = "xyt: a launch without the A-product must compute the Euclidean objective"; return NMFX_E_ARG; }
        static const bool rows16o = getenv("NMFX_XYT16") && atoi(getenv("NMFX_XYT16")) == 1;
        if (tiled && !rows16o) {                       // the 32-row kernel without its A stages (r3)
            if (E->kp == 64) return launch_xyt32_t<true, 3, false, 64, false>(E, X, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, nullptr, nullptr, ng);
            return launch_xyt32_t<true, 3, false, 128, false>(E, X, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, nullptr, nullptr, ng);
        }
        if (E->kp == 64)
            return launch_xyt_t<64, true, false, 3, false>(E, X, tiled, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, nullptr, nullptr, ng);
        return launch_xyt_t<128, true, false, 3, false>(E, X, tiled, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, nullptr, nullptr, ng);
    }
    if (E->kp == 64) {
        // k padded to 64: the 32-row kernel (NMFX_XYT16=1 keeps the 16-row form, for A/B runs)
        static const bool rows16_env = getenv("NMFX_XYT16") && atoi(getenv("NMFX_XYT16")) == 1;
        const bool rows16 = rows16_env && !E->xyt_xpriv;      // (an X in the auxiliaries' register order is only read by the 32-row kernel)
        if (kl && tiled && !rows16) {
#define NMFX_X32K(OBJ_, T_) launch_xyt32_t<OBJ_, T_, true>(E, X, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, Apart, nullptr, ng)
            if (terms == 3) return obj ? NMFX_X32K(true, 3) : NMFX_X32K(false, 3);
            return obj ? NMFX_X32K(true, 4) : NMFX_X32K(false, 4);
#undef NMFX_X32K
        }
        if (kl) return obj ? NMFX_XYT(64, true, true) : NMFX_XYT(64, false, true);
        if (tiled && gram_part && !rows16) {
#define NMFX_X32(OBJ_, T_) launch_xyt32_t<OBJ_, T_>(E, X, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, Apart, gram_part, ng)
            if (terms == 3) return obj ? NMFX_X32(true, 3) : NMFX_X32(false, 3);
            return obj ? NMFX_X32(true, 4) : NMFX_X32(false, 4);
#undef NMFX_X32
        }
        return obj ? NMFX_XYT(64, true, false) : NMFX_XYT(64, false, false);
    }
    if (kl) {      // MUR-KL at k padded to 128: the 32-row kernel as well (r3; NMFX_XYT16=1: the 16-row form)
        static const bool rows16k = getenv("NMFX_XYT16") && atoi(getenv("NMFX_XYT16")) == 1;
        if (tiled && !rows16k) {
#define NMFX_X32KB(OBJ_, T_) launch_xyt32_t<OBJ_, T_, true, 128>(E, X, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, Apart, nullptr, ng)
            if (terms == 3) return obj ? NMFX_X32KB(true, 3) : NMFX_X32KB(false, 3);
            return obj ? NMFX_X32KB(true, 4) : NMFX_X32KB(false, 4);
#undef NMFX_X32KB
        }
        return obj ? NMFX_XYT(128, true, true) : NMFX_XYT(128, false, true);
    }
    {   // Euclidean products with k padded to 128: the 32-row kernel as well (NMFX_XYT16=1: the 16-row form)
        static const bool rows16_env = getenv("NMFX_XYT16") && atoi(getenv("NMFX_XYT16")) == 1;
        const bool rows16 = rows16_env && !E->xyt_xpriv;
        if (tiled && !rows16) {
#define NMFX_X32B(OBJ_, T_) launch_xyt32_t<OBJ_, T_, false, 128>(E, X, ldx, R, ngroups, splits, Yhi, Ylo, ldy, Zhi, Zlo, Apart, nullptr, ng)
            if (terms == 3) return obj ? NMFX_X32B(true, 3) : NMFX_X32B(false, 3);
            return obj ? NMFX_X32B(true, 4) : NMFX_X32B(false, 4);
#undef NMFX_X32B
        }
    }
    return obj ? NMFX_XYT(128, true, false) : NMFX_XYT(128, false, false);
#undef NMFX_XYT2
#undef NMFX_XYT
}

static int kl_part_alloc(nmfx_engine* E) {
    return lazy_alloc(E, &E->kl_part, (int64_t)(E->np / 64 + E->mp / 64) * E->kp);
}

// W_new, its images in both layouts (buffer `nxt`), and the column-sum partials kl_part[np/64 ..][kp] (mp / 64 of them)
int nmfx_bf16_kl_w_epilogue(nmfx_engine* E, const float* Wold, float* Wnew, int nxt, float lam, const float* rowsum) {
    ProfScope ps(E, "w_update");
    int rc;
    if ((rc = kl_part_alloc(E))) return rc;
    hipLaunchKernelGGL(kl_w_epilogue_kernel, dim3((unsigned)(E->kp / 64), (unsigned)(E->mp / 64)), dim3(256), 0, E->stream,
                       E->A_part, E->bf_wsplit, E->mp * E->kp, E->kp, E->k, Wold, rowsum, lam, Wnew, E->mp, E->Whi[nxt],
                       E->Wlo[nxt], E->WThi, E->WTlo, E->kl_part + (int64_t)(E->np / 64) * E->kp, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// H_new from the exchange buffers, its images in both layouts, and the row-sum partials kl_part[0 .. np/64)[kp]
int nmfx_bf16_kl_h_epilogue(nmfx_engine* E, float lam, int64_t j, int64_t min_iter, double tol1, double tol2) {
    ProfScope ps(E, "h_update");
    int rc;
    if ((rc = kl_part_alloc(E))) return rc;
    if ((rc = lazy_alloc(E, &E->HThi, (int64_t)E->kp * E->np))) return rc;
    if ((rc = lazy_alloc(E, &E->HTlo, (int64_t)E->kp * E->np))) return rc;
    hipLaunchKernelGGL(kl_h_epilogue_kernel, dim3((unsigned)(E->np / 64), (unsigned)(E->kp / 64)), dim3(256), 0, E->stream,
                       E->xf32, E->xf64, E->H, E->np, E->kp, E->k, lam, (long long)j, (long long)min_iter, tol1, tol2,
                       E->state, E->obj_hist, E->Hhi, E->Hlo, E->HThi, E->HTlo, E->kl_part);
    NMFX_HIP(hipGetLastError());
    E->himg_both = true;
    E->kl_h_iter = j;
    return NMFX_OK;
}

// Allocate the bf16 state and build V^T and the images of the initial factors.
int nmfx_bf16_prepare(nmfx_engine* E) {
    if (E->bf_ready) return NMFX_OK;
    int rc;
    const int64_t mp = E->mp, np = E->np, kp = E->kp;
    if ((rc = nmfx_need_v(E))) return rc;              // (a second prepare after new factors / new rows of V)
    if ((rc = lazy_alloc(E, &E->Vt, mp * np))) return rc;
    if ((rc = lazy_alloc(E, &E->Vtile, mp * np))) return rc;
    for (int b = 0; b < 2; ++b) {
        if ((rc = lazy_alloc(E, &E->Whi[b], mp * kp))) return rc;
        if ((rc = lazy_alloc(E, &E->Wlo[b], mp * kp))) return rc;
    }
    if ((rc = lazy_alloc(E, &E->WThi, mp * kp))) return rc;
    if ((rc = lazy_alloc(E, &E->WTlo, mp * kp))) return rc;
    if ((rc = lazy_alloc(E, &E->Hhi, kp * np))) return rc;
    if ((rc = lazy_alloc(E, &E->Hlo, kp * np))) return rc;
    // splits of the H phase: rows of V^T are columns of V
    // one 512-thread block per CU: grid = (rows / 128) x splits ~ number of CUs
    const int64_t rbt = np / 128, cbt = mp / 64;
    int64_t hs2 = std::max<int64_t>(1, ((int64_t)E->ncu + rbt / 2) / rbt);
    hs2 = std::min<int64_t>(hs2, std::max<int64_t>(1, cbt / 4));
    E->bt_split = (int)hs2;
    int64_t ws2 = std::max<int64_t>(1, ((int64_t)E->ncu + (mp / 128) / 2) / (mp / 128));
    ws2 = std::min<int64_t>(ws2, std::max<int64_t>(1, (np / 64) / 4));
    ws2 = std::min<int64_t>(ws2, E->wsplit);           // A_part was sized for wsplit slabs
    E->bf_wsplit = (int)ws2;
    // row blocks that share the Gram by-product: every consumer block sums all ng x splits slabs
    // (a chain of dependent L2 reads), so keep that product at 16 -- short shards already have many splits
    E->gram_ng_w = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, 16 / ws2), mp / 128));
    E->gram_ng_h = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(4, 16 / hs2), np / 128));
    if ((rc = lazy_alloc(E, &E->Bt_part, hs2 * np * kp))) return rc;
    hipLaunchKernelGGL(transpose_tiled_kernel, dim3((unsigned)(np / 64), (unsigned)(mp / 64)), dim3(256), 0, E->stream,
                       E->V, np, E->Vt, mp);
    hipLaunchKernelGGL(retile_kernel, dim3((unsigned)(np / 64), (unsigned)(mp / 128)), dim3(256), 0, E->stream,
                       E->V, np, E->Vtile);
    if (E->drop_v) {               // two copies of V instead of three from here on (nmfx_need_v rebuilds the row-major one)
        NMFX_HIP(hipStreamSynchronize(E->stream));
        NMFX_HIP(hipFree(E->V));
        E->V = nullptr;
    }
    E->bf_ready = true;
    if ((rc = nmfx_bf16_images_w(E, E->W[E->wsel], E->wsel))) return rc;
    return nmfx_bf16_images_h(E, false);
}

// ---- building blocks shared by the solvers (kp = 64 or 128); results land where the exact-f32
// kernels put theirs, so everything downstream (epilogues, inner rounds) is unchanged ----
int nmfx_bf16_images_w(nmfx_engine* E, const float* W, int buf) {      // Whi/Wlo[buf] ([mp][kp]) and WThi/WTlo ([kp][mp])
    ProfScope ps(E, "images");
    hipLaunchKernelGGL(split_images_kernel, dim3((unsigned)(E->kp / 64), (unsigned)(E->mp / 64)), dim3(256), 0, E->stream,
                       W, E->mp, (int64_t)E->kp, (int64_t)E->kp, E->Whi[buf], E->Wlo[buf], E->WThi, E->WTlo);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_bf16_images_h(nmfx_engine* E, bool transposed, const float* src) {   // Hhi/Hlo ([kp][np]) (+ HThi/HTlo ([np][kp])) of src (default H)
    ProfScope ps(E, "images");
    int rc;
    E->himg_both = transposed && !src;
    if (transposed) {
        if ((rc = lazy_alloc(E, &E->HThi, (int64_t)E->kp * E->np))) return rc;
        if ((rc = lazy_alloc(E, &E->HTlo, (int64_t)E->kp * E->np))) return rc;
    }
    hipLaunchKernelGGL(split_images_kernel, dim3((unsigned)(E->np / 64), (unsigned)(E->kp / 64)), dim3(256), 0, E->stream,
                       src ? src : E->H, (int64_t)E->kp, E->np, E->np, E->Hhi, E->Hlo, transposed ? E->HThi : nullptr,
                       transposed ? E->HTlo : nullptr);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// A_part[bf_wsplit][mp][kp] = V H^T (+ obj_part[(mp/128) * bf_wsplit] = residual objective of (W images `zbuf`, H))
// kl: A_part = (V / (W H + 1e-9)) H^T and the KL objective (the W images `zbuf` are then always read)
int nmfx_bf16_vht(nmfx_engine* E, bool obj, int zbuf, const char* name, bool kl, int terms) {
    const bool z = obj || kl;
    return launch_xyt(E, obj, E->Vtile, true, E->np, E->mp, (int)(E->np / 64), E->bf_wsplit, E->Hhi, E->Hlo, E->np,
                      z ? E->Whi[zbuf] : nullptr, z ? E->Wlo[zbuf] : nullptr, E->A_part,
                      E->kp == 64 ? E->HHt_part : nullptr, name, kl, E->gram_ng_w, terms);
}

// Bt_part[bt_split][np][kp] = V^T W (+ obj_part[(np/128) * bt_split] = residual objective, Z = H^T images)
// kl: Bt_part = (V / (W H + 1e-9))^T W
// obj_part[(mp/128) * bf_wsplit] = residual objective of (W images `zbuf`, H images): one pass over V, no product output
int nmfx_bf16_objective(nmfx_engine* E, int zbuf, const char* name) {
    return launch_xyt(E, true, E->Vtile, true, E->np, E->mp, (int)(E->np / 64), E->bf_wsplit, E->Hhi, E->Hlo, E->np,
                      E->Whi[zbuf], E->Wlo[zbuf], nullptr, nullptr, name);
}

int nmfx_bf16_vtw(nmfx_engine* E, bool obj, const char* name, bool kl, int terms) {
    const bool z = obj || kl;
    return launch_xyt(E, obj, E->Vt, true, E->mp, E->np, (int)(E->mp / 64), E->bt_split, E->WThi, E->WTlo, E->mp,
                      z ? E->HThi : nullptr, z ? E->HTlo : nullptr, E->Bt_part,
                      E->kp == 64 ? E->G_part : nullptr, name, kl, E->gram_ng_h, terms);
}

// ---- KL-loss ADMM variants on the split-bf16 kernels (r4) ----
// State: S = v_aux + dual_v and dual_v, tile-major, in the orientation of the sub-problem at hand -- side 0 (H sub-problem): rows n
// like Vt (E->kl_S[0], kl_DV[0]); side 1 (W sub-problem): rows m like Vtile (kl_S[1], kl_DV[1]).
int nmfx_bf16_kl_state(nmfx_engine* E, bool reset) {
    int rc;
    const int64_t cnt = E->mp * E->np;
    for (int i = 0; i < 2; ++i) {
        if ((rc = lazy_alloc(E, &E->kl_S[i], cnt))) return rc;
        if ((rc = lazy_alloc(E, &E->kl_DV[i], cnt))) return rc;
    }
    if (reset) {           // v_aux = dual_v = 0 (ao_admm.py:28-30, admm.py:31-35: both names are bound to ONE array of zeros)
        for (int i = 0; i < 2; ++i) {
            NMFX_HIP(hipMemsetAsync(E->kl_S[i], 0, (size_t)cnt * sizeof(float), E->stream));
            NMFX_HIP(hipMemsetAsync(E->kl_DV[i], 0, (size_t)cnt * sizeof(float), E->stream));
        }
        E->kl_side = 0; E->kl_s_side = 0;
    }
    return NMFX_OK;
}

// bring the state into the orientation of `side` (a no-op for what is there): S (with_s) from the side that holds it (kl_s_side), dual_v
// (with_dv) from kl_side.  r5: a sub-problem's first product can read S from the other orientation's tiles (nmfx_bf16_kl_product, gather),
// so the AO-ADMM loop only moves dual_v and the ADMM loop nothing
int nmfx_bf16_kl_orient(nmfx_engine* E, int side, bool with_dv, bool with_s) {
    const bool do_s = with_s && E->kl_s_side != side, do_dv = with_dv && E->kl_side != side;
    if (!do_s && !do_dv) return NMFX_OK;
    ProfScope ps(E, "transpose");
    const int from = 1 - side;
    const int64_t R = from == 0 ? E->np : E->mp, C = from == 0 ? E->mp : E->np;
    const dim3 grid((unsigned)(C / 64), (unsigned)(R / 64));
    if (do_s)
        hipLaunchKernelGGL(tile_transpose_kernel<true>, grid, dim3(256), 0, E->stream, (const float*)E->kl_S[from], R, C, E->kl_S[side], &E->state->flag);
    if (do_dv)
        hipLaunchKernelGGL(tile_transpose_kernel<true>, grid, dim3(256), 0, E->stream, (const float*)E->kl_DV[from], R, C, E->kl_DV[side], &E->state->flag);
    NMFX_HIP(hipGetLastError());
    if (do_s) E->kl_s_side = side;
    if (do_dv) E->kl_side = side;
    return NMFX_OK;
}

// the m x n auxiliaries (see xyt32_bf16_kernel<..., VAUX>): side 0: X = V^T, Y = W^T images, Z = (h_aux)^T images (HThi / HTlo);
// side 1: X = V, Y = H-like images (Hhi / Hlo), Z = W-like images (Whi / Wlo[0]); flag2: optional second skip flag (the inner stop)
int nmfx_bf16_vaux(nmfx_engine* E, int side, const int* flag2) {
    ProfScope ps(E, "kl_vaux");
    const float* X = side == 0 ? E->Vt : E->Vtile;
    const int64_t ldx = side == 0 ? E->mp : E->np, R = side == 0 ? E->np : E->mp;
    const int splits = side == 0 ? E->bt_split : E->bf_wsplit;
    const unsigned short* Yhi = side == 0 ? E->WThi : E->Hhi;
    const unsigned short* Ylo = side == 0 ? E->WTlo : E->Hlo;
    const unsigned short* Zhi = side == 0 ? E->HThi : E->Whi[0];
    const unsigned short* Zlo = side == 0 ? E->HTlo : E->Wlo[0];
    if (!E->kl_S[side] || !Zhi) { E->err = "vaux: state or images missing"; return NMFX_E_STATE; }
    const dim3 grid((unsigned)(R / 128), (unsigned)splits), block(512);
    const size_t shm = 160 * 1024;
    auto k64 = xyt32_bf16_kernel<true, 3, 0, false, 64, 1, false, false, 8, false, 1>;
    auto k128 = xyt32_bf16_kernel<true, 3, 0, false, 128, 1, false, false, 8, false, 1>;
    auto kern = E->kp == 64 ? k64 : k128;
    int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, block, shm, E->stream, X, ldx, Yhi, Ylo, ldx, Zhi, Zlo, (float*)nullptr, E->obj_part, (float*)nullptr, R,
                       (int)(ldx / 64), &E->state->flag, 1, (const int4*)nullptr, (const int*)nullptr, 0, XytSide(), E->kl_DV[side], E->kl_S[side], flag2, 0);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// r5: the auxiliaries of round r AND the right-hand-side product of round r + 1 in one launch (xyt32_bf16_kernel<..., VAUXF>):
// slabs where nmfx_bf16_kl_product(E, side, 4) leaves them, bit for bit (same grid, same group order, S split in registers as the product
// splits it after its LDS read).  nrm / nblk: the `terminate` partials round r's factor kernel has just written; last: r is the final round.
int nmfx_bf16_vaux_fused(nmfx_engine* E, int side, const int* flag2, const double* nrm, int nblk, bool last) {
    ProfScope ps(E, "kl_vaux_fused");
    const float* X = side == 0 ? E->Vt : E->Vtile;
    const int64_t ldx = side == 0 ? E->mp : E->np, R = side == 0 ? E->np : E->mp;
    const int splits = side == 0 ? E->bt_split : E->bf_wsplit;
    const unsigned short* Yhi = side == 0 ? E->WThi : E->Hhi;
    const unsigned short* Ylo = side == 0 ? E->WTlo : E->Hlo;
    const unsigned short* Zhi = side == 0 ? E->HThi : E->Whi[0];
    const unsigned short* Zlo = side == 0 ? E->HTlo : E->Wlo[0];
    float* Apart = side == 0 ? E->Bt_part : E->A_part;
    if (!E->kl_S[side] || !Zhi || !Apart) { E->err = "vaux_fused: state, images or slabs missing"; return NMFX_E_STATE; }
    const dim3 grid((unsigned)(R / 128), (unsigned)splits), block(512);
    const size_t shm = 160 * 1024;
    auto k64 = xyt32_bf16_kernel<false, 4, 0, true, 64, 1, false, true, 8, false, 3>;
    auto k128 = xyt32_bf16_kernel<false, 4, 0, true, 128, 1, false, true, 8, false, 3>;
    auto kern = E->kp == 64 ? k64 : k128;
    int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, block, shm, E->stream, X, ldx, Yhi, Ylo, ldx, Zhi, Zlo, Apart, const_cast<double*>(nrm), (float*)nullptr, R,
                       (int)(ldx / 64), &E->state->flag, last ? 1 : 0, (const int4*)nullptr, (const int*)nullptr, nblk, XytSide(), E->kl_DV[side], E->kl_S[side], flag2, 0);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// obj_part[(mp / 128) * bf_wsplit] = KL(V, W H) partials from the images Whi / Wlo[0] and Hhi / Hlo (one pass over the tile-major V)
int nmfx_bf16_kl_objective(nmfx_engine* E) {
    ProfScope ps(E, "objective");
    const dim3 grid((unsigned)(E->mp / 128), (unsigned)E->bf_wsplit), block(512);
    const size_t shm = 160 * 1024;
    auto k64 = xyt32_bf16_kernel<true, 3, 0, false, 64, 1, false, false, 8, false, 2>;
    auto k128 = xyt32_bf16_kernel<true, 3, 0, false, 128, 1, false, false, 8, false, 2>;
    auto kern = E->kp == 64 ? k64 : k128;
    int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc) return rc;
    E->obj_count = (E->mp / 128) * E->bf_wsplit;
    hipLaunchKernelGGL(kern, grid, block, shm, E->stream, (const float*)E->Vtile, E->np, (const unsigned short*)E->Hhi, (const unsigned short*)E->Hlo, E->np,
                       (const unsigned short*)E->Whi[0], (const unsigned short*)E->Wlo[0], (float*)nullptr, E->obj_part, (float*)nullptr, E->mp,
                       (int)(E->np / 64), &E->state->flag, 1, (const int4*)nullptr, (const int*)nullptr, 0, XytSide(), (float*)nullptr, (float*)nullptr,
                       (const int*)nullptr, 0);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// the two right-hand-side products with S in the place of V (slabs as nmfx_bf16_vtw / nmfx_bf16_vht leave them; no objective).
// gather (r5): S is read from the OTHER orientation's buffer through the transposing request pattern of xyt32_bf16_kernel (VMODE 4, XGATHER) --
// the same operand values in the same order as from a transposed copy, hence the same slabs bit for bit, without the V-sized transpose
int nmfx_bf16_kl_product(nmfx_engine* E, int side, int terms, const int* flag2, bool gather) {
    (void)flag2;       // (a product behind the inner stop is wasted work, not a wrong result: its consumers are no-ops)
    const float* S = E->kl_S[gather ? 1 - side : side];
    const int64_t ldx = side == 0 ? E->mp : E->np, R = side == 0 ? E->np : E->mp;
    const int splits = side == 0 ? E->bt_split : E->bf_wsplit;
    const unsigned short* Yhi = side == 0 ? E->WThi : E->Hhi;
    const unsigned short* Ylo = side == 0 ? E->WTlo : E->Hlo;
    float* Apart = side == 0 ? E->Bt_part : E->A_part;
    float* gram = E->kp == 64 ? (side == 0 ? E->G_part : E->HHt_part) : nullptr;
    const int ng = side == 0 ? E->gram_ng_h : E->gram_ng_w;
    const char* name = side == 0 ? "hphase" : "wphase_noobj";
    if (gather && terms == 4) {                        // the launch launch_xyt would make for these arguments, in the XGATHER form of the kernel
        ProfScope ps(E, name);
        E->xyt_nw = 8;
        const dim3 grid((unsigned)(R / 128), (unsigned)splits), block(512);
        const size_t shm = 160 * 1024;
        auto k64 = xyt32_bf16_kernel<false, 4, 0, false, 64, 1, false, true, 8, false, 4>;
        auto k128 = xyt32_bf16_kernel<false, 4, 0, false, 128, 1, false, true, 8, false, 4>;
        auto kern = E->kp == 64 ? k64 : k128;
        int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm); if (rc) return rc;
        hipLaunchKernelGGL(kern, grid, block, shm, E->stream, S, ldx, Yhi, Ylo, ldx, (const unsigned short*)nullptr, (const unsigned short*)nullptr, Apart, E->obj_part,
                           gram, R, (int)(ldx / 64), &E->state->flag, ng, (const int4*)nullptr, (const int*)nullptr, 0, XytSide(), (float*)nullptr, (float*)nullptr,
                           E->xyt_flag2, 1);
        NMFX_HIP(hipGetLastError());
        return NMFX_OK;
    }
    if (gather) { E->err = "kl_product: the gathered form is built for four terms"; return NMFX_E_ARG; }
    E->xyt_xpriv = 1;                                  // S lies in the auxiliaries kernel's register order (r5)
    const int rc = launch_xyt(E, false, S, true, ldx, R, (int)(ldx / 64), splits, Yhi, Ylo, ldx, nullptr, nullptr, Apart, gram, name, false, ng, terms);
    E->xyt_xpriv = 0;
    return rc;
}

// xf32 = [ (sum of the B^T slabs)^T  (kp x np) | sum of the G slabs ], xf64[0] = sum of obj_part
// r4 (behind a stream-K product): bcnt = slabs per 128-column block of B^T (bsplit = the slab stride's count is then unused),
// Gpart = nullptr (the side job has summed the Gram slabs itself), st != nullptr: the objective is recorded as obj[j] here
template <int KP>
__global__ __launch_bounds__(256) void pack_t_kernel(
    const float* __restrict__ Btpart, int bsplit, int64_t np, const float* __restrict__ Gpart, int gsplit,
    const double* __restrict__ objpart, int64_t nobj, float* __restrict__ xf32, double* __restrict__ xf64,
    const int* __restrict__ flag, const int* __restrict__ bcnt = nullptr, DevState* __restrict__ st = nullptr,
    double* __restrict__ obj_hist = nullptr, long long j = 0, long long min_iter = 0, double tol1 = 0.0, double tol2 = 0.0)
{
    if (*flag) return;
    __shared__ float tile[64][KP + 1];
    __shared__ double sh[4];
    const int tid = threadIdx.x, nbb = (int)(np / 64), b = blockIdx.x;
    const int64_t bcount = (int64_t)KP * np;
    if (b < nbb) {
        const int64_t c0 = (int64_t)b * 64;
        constexpr int NE = 64 * (KP / 4) / 256;        // 16-byte pieces of the tile per thread: all their slab loads in flight together
        float v[NE][4];
#pragma unroll
        for (int u = 0; u < NE; ++u) { v[u][0] = 0.f; v[u][1] = 0.f; v[u][2] = 0.f; v[u][3] = 0.f; }
        if (bcnt) bsplit = bcnt[c0 >> 7];
        for (int p = 0; p < bsplit; ++p) {             // (slab order per element, as before)
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int e = tid + 256 * u, c = e / (KP / 4), j4 = e % (KP / 4);
                add4(v[u], Btpart + (int64_t)p * bcount + (c0 + c) * KP + 4 * j4);
            }
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + 256 * u, c = e / (KP / 4), j4 = e % (KP / 4);
            tile[c][4 * j4] = v[u][0]; tile[c][4 * j4 + 1] = v[u][1]; tile[c][4 * j4 + 2] = v[u][2]; tile[c][4 * j4 + 3] = v[u][3];
        }
        __syncthreads();
        for (int e = tid; e < KP * 64; e += 256) xf32[(int64_t)(e >> 6) * np + c0 + (e & 63)] = tile[e & 63][e >> 6];
    } else if (b < nbb + KP * KP / 256) {
        if (!Gpart) return;
        const int64_t i = (int64_t)(b - nbb) * 256 + tid;
        float s2 = 0.f;
        int p = 0;
        for (; p + 8 <= gsplit; p += 8) {               // eight slabs in flight together, added in slab order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = Gpart[(int64_t)(p + u) * KP * KP + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s2 += t[u];
        }
        for (; p < gsplit; ++p) s2 += Gpart[(int64_t)p * KP * KP + i];
        xf32[bcount + i] = s2;
    } else {
        if (nobj < 0) return;                          // (no objective behind this product: xf64[0] stays what it is)
        double t = 0.0;
        for (int64_t i = tid; i < nobj; i += 256) t += objpart[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if ((tid & 63) == 0) sh[tid >> 6] = t;
        __syncthreads();
        if (tid == 0) {
            const double obj = ((sh[0] + sh[1]) + sh[2]) + sh[3];
            xf64[0] = obj;
            if (st) {
                const int rule = nmfx_record_objective(st, obj_hist, obj, j, min_iter, tol1, tol2, true);
                if (!rule && st->notpd_pending) { st->notpd = 1; st->flag = 3; }      // (notpd_pending: written by every side job, never here)
            }
        }
    }
}

// ---- stream-K plan (see xyt32_bf16_kernel<..., SK>) ----
bool nmfx_sk_enabled(const nmfx_engine* E) {
    static const bool on = !(getenv("NMFX_SK") && atoi(getenv("NMFX_SK")) == 0);
    return on && E->kp == 128 && E->precision == 1 && nmfx_bf16_supported(E);
}

static int sk_plan(nmfx_engine* E, int side) {
    nmfx_engine::SkPlan& P = E->sk[side];
    if (P.seg) return NMFX_OK;
    const int64_t R = side == 0 ? E->np : E->mp;       // rows of X
    const int ngroups = (int)((side == 0 ? E->mp : E->np) / 64);
    const int64_t rb = R / 128, units = rb * ngroups;
    static const int forced = getenv("NMFX_SK_WORKERS") ? atoi(getenv("NMFX_SK_WORKERS")) : 0;
    // one CU stays free for the side job; runs of at least four groups (the depth of the kernel's rings) where the problem allows
    int64_t workers = std::min<int64_t>(E->ncu - 1, std::max<int64_t>(1, units / 4));
    if (forced > 0) workers = std::min<int64_t>(forced, units);
    // Runs of equal COST: a worker's cost is its groups plus `cross` group-equivalents per extra segment (the epilogue of a segment --
    // exchange of the partial tiles, 64 KiB of slab stores -- and the refill of the rings: measured 7-10 us on config 3, where a group
    // takes 1.5 us on the W side and 2.4 us with the objective).  The smallest bound T that `workers` greedy runs can meet, by bisection.
    static const int cross = getenv("NMFX_SK_CROSS") ? atoi(getenv("NMFX_SK_CROSS")) : 5;
    auto deal = [&](int64_t T, std::vector<int64_t>* ends) {        // greedy runs of cost <= T; returns the number of runs
        int64_t u = 0, nw = 0;
        while (u < units) {
            int64_t cost = 0, v = u;
            bool first_seg = true;
            while (v < units) {
                const int64_t b = v / ngroups, room = (b + 1) * ngroups - v;
                const int64_t extra = first_seg ? 0 : cross;
                if (cost + extra + 1 > T) break;
                const int64_t take = std::min<int64_t>(room, T - cost - extra);
                cost += extra + take; v += take; first_seg = false;
                if (take < room) break;
            }
            if (v == u) v = u + 1;                     // (T < 1 cannot happen; guard against an endless loop)
            u = v; ++nw;
            if (ends) ends->push_back(u);
        }
        return nw;
    };
    int64_t lo = 1, hi = units + cross * rb;
    while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (deal(mid, nullptr) <= workers) hi = mid; else lo = mid + 1; }
    std::vector<int64_t> ends;
    workers = deal(lo, &ends);
    std::vector<int4> seg;
    std::vector<int> first(workers + 1), cnt(rb, 0);
    static const bool cyclic = !(getenv("NMFX_SK_CYCLIC") && atoi(getenv("NMFX_SK_CYCLIC")) == 0);
    for (int64_t w = 0; w < workers; ++w) {
        first[w] = (int)seg.size();
        int64_t u = w ? ends[w - 1] : 0, tau = 0;      // tau: groups this worker has done before the segment (its clock)
        const int64_t u1 = ends[w];
        while (u < u1) {
            const int64_t b = u / ngroups, e = std::min<int64_t>(u1, (b + 1) * ngroups);
            const int64_t a = u - b * ngroups, z = e - b * ngroups;
            // start at the group congruent to the clock modulo the common run length lo (if the segment has one): see the kernel
            int64_t st = a + (((tau - a) % lo) + lo) % lo;
            if (st >= z || !cyclic) st = a;
            if (cnt[b] > 255 || st > 0x7fffff) { E->err = "stream-K plan: slab / start out of range"; return NMFX_E_ARG; }
            seg.push_back(make_int4((int)b, (int)a, (int)z, cnt[b]++ | (int)(st << 8)));
            tau += z - a;
            u = e;
        }
    }
    first[workers] = (int)seg.size();
    int maxslab = 1;
    for (int64_t b = 0; b < rb; ++b) maxslab = std::max(maxslab, cnt[b]);
    // validate first, build in locals, publish last: a plan that failed half way must not look finished to the next call (ADVICE r4:
    // `if (P.seg) return NMFX_OK` would then launch with an oversized nseg or a null slab buffer)
    if ((int64_t)seg.size() > E->obj_part_cap) { E->err = "stream-K plan: more segments than objective partials"; return NMFX_E_ARG; }
    int4* d_seg = nullptr; int* d_first = nullptr; int* d_cnt = nullptr; float* d_slabs = nullptr;
    const bool ok =
        hipMalloc(reinterpret_cast<void**>(&d_seg), seg.size() * sizeof(int4)) == hipSuccess &&
        hipMalloc(reinterpret_cast<void**>(&d_first), first.size() * sizeof(int)) == hipSuccess &&
        hipMalloc(reinterpret_cast<void**>(&d_cnt), cnt.size() * sizeof(int)) == hipSuccess &&
        hipMalloc(reinterpret_cast<void**>(&d_slabs), (size_t)maxslab * R * E->kp * sizeof(float)) == hipSuccess &&
        hipMemcpy(d_seg, seg.data(), seg.size() * sizeof(int4), hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(d_cnt, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (d_seg) hipFree(d_seg);
        if (d_first) hipFree(d_first);
        if (d_cnt) hipFree(d_cnt);
        if (d_slabs) hipFree(d_slabs);
        E->err = "stream-K plan: device allocation or copy failed";
        return NMFX_E_HIP;
    }
    P.first = d_first; P.cnt = d_cnt; P.slabs = d_slabs;
    P.workers = (int)workers; P.nseg = (int)seg.size(); P.maxslab = maxslab;
    P.seg = d_seg;                                     // (last: what `if (P.seg) return NMFX_OK` looks at)
    return NMFX_OK;
}

int nmfx_bf16_sk_product(nmfx_engine* E, int side, bool obj, const float* gsrc, int gslabs, double fixed_rho, const char* name) {
    int rc;
    if ((rc = sk_plan(E, side))) return rc;
    ProfScope ps(E, name);
    const nmfx_engine::SkPlan& P = E->sk[side];
    const float* X = side == 0 ? E->Vt : E->Vtile;
    const int64_t ldx = side == 0 ? E->mp : E->np, R = side == 0 ? E->np : E->mp;
    const int ngroups = (int)(ldx / 64);
    const unsigned short* Yhi = side == 0 ? E->WThi : E->Hhi;
    const unsigned short* Ylo = side == 0 ? E->WTlo : E->Hlo;
    const unsigned short* Zhi = !obj ? nullptr : side == 0 ? E->HThi : E->Whi[0];
    const unsigned short* Zlo = !obj ? nullptr : side == 0 ? E->HTlo : E->Wlo[0];
    XytSide job; job.gsrc = gsrc; job.gslabs = gslabs; job.k = E->k; job.Minv = E->Minv; job.st = E->state; job.fixed_rho = fixed_rho;
    job.defer = side == 0 ? 1 : 0;
    const dim3 grid((unsigned)(P.workers + (gsrc ? 1 : 0))), block(512);
    const size_t shm = 160 * 1024;
    auto kern = obj ? xyt32_bf16_kernel<true, 3, 0, false, 128, 1, false, true, 8, true>
                    : xyt32_bf16_kernel<false, 3, 0, false, 128, 1, false, true, 8, true>;
    if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm))) return rc;
    if (obj) E->obj_count = P.nseg;
    hipLaunchKernelGGL(kern, grid, block, shm, E->stream, X, ldx, Yhi, Ylo, ldx, Zhi, Zlo, P.slabs, E->obj_part,
                       (float*)nullptr, R, ngroups, &E->state->flag, 1, (const int4*)P.seg, (const int*)P.first, P.workers, job, (float*)nullptr, (float*)nullptr, (const int*)nullptr, 0);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_bf16_pack_sk(nmfx_engine* E, int64_t j, int64_t min_iter, double tol1, double tol2) {
    ProfScope ps(E, "pack");
    const nmfx_engine::SkPlan& P = E->sk[0];
    const unsigned grid = (unsigned)(E->np / 64 + E->kp * E->kp / 256 + 1);
    hipLaunchKernelGGL((pack_t_kernel<128>), dim3(grid), dim3(256), 0, E->stream, P.slabs, P.maxslab, E->np, (const float*)nullptr, 0,
                       E->obj_part, (int64_t)P.nseg, E->xf32, E->xf64, &E->state->flag, (const int*)P.cnt, E->state, E->obj_hist,
                       (long long)j, (long long)min_iter, tol1, tol2);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_bf16_pack_t(nmfx_engine* E, const float* Gpart, int gsplit, int64_t nobj) {
    ProfScope ps(E, "pack");
    const unsigned grid = (unsigned)(E->np / 64 + E->kp * E->kp / 256 + 1);
    if (E->kp == 64)
        hipLaunchKernelGGL((pack_t_kernel<64>), dim3(grid), dim3(256), 0, E->stream, E->Bt_part, E->bt_split, E->np, Gpart,
                           gsplit, E->obj_part, nobj, E->xf32, E->xf64, &E->state->flag);
    else
        hipLaunchKernelGGL((pack_t_kernel<128>), dim3(grid), dim3(256), 0, E->stream, E->Bt_part, E->bt_split, E->np, Gpart,
                           gsplit, E->obj_part, nobj, E->xf32, E->xf64, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int KP>
static int launch_w_update_bf16(nmfx_engine* E, const float* Wold, float* Wnew, int nxt, const float* hht, int hslabs, float lam) {
    ProfScope ps(E, "w_update");
    constexpr size_t shm = (size_t)(KP * (KP + 16) + 64 * (KP + 4)) * sizeof(float) + (size_t)2 * KP * 66 * sizeof(unsigned short);
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(mur_w_update_bf16_kernel<KP>), (int)shm); if (rc_) return rc_; }
    hipLaunchKernelGGL(mur_w_update_bf16_kernel<KP>, dim3((unsigned)(E->mp / 64)), dim3(512), shm, E->stream, E->A_part,
                       E->bf_wsplit, E->mp, Wold, hht, hslabs, lam, Wnew, E->Whi[nxt], E->Wlo[nxt], E->WThi, E->WTlo,
                       &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// from_slabs: B^T, G and the objective partials straight from the H phase's slabs (single GPU, kp = 64);
// otherwise from the (all-reduced) exchange buffers: xf32 = [B^T sums [np][kp] | G], xf64[0] = objective
template <int KP>
static int launch_h_update_bf16(nmfx_engine* E, bool from_slabs, float lam, int64_t j, int64_t min_iter, double tol1, double tol2,
                               int cb0 = 0, int nblk = -1) {      // (cb0, nblk: this rank's column blocks of the sliced update, with the staged copy)
    ProfScope ps(E, "h_update");
    const bool sliced = nblk >= 0;
    const dim3 grid((unsigned)(sliced ? nblk : E->np / 64)), block(512);
    if (sliced && nblk == 0) return NMFX_OK;
    constexpr size_t shm = (size_t)(KP * (KP + 4) + KP * 80 + 64 * (KP + 4)) * sizeof(float);
    { int rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(mur_h_update_bf16_kernel<KP, true>), (int)shm); if (rc_) return rc_;
      rc_ = nmfx_allow_lds(E, reinterpret_cast<const void*>(mur_h_update_bf16_kernel<KP, false>), (int)shm); if (rc_) return rc_; }
    if (from_slabs)
        hipLaunchKernelGGL((mur_h_update_bf16_kernel<KP, true>), grid, block, shm, E->stream, E->Bt_part, E->bt_split,
                           E->G_part, nmfx_bf16_g_slabs(E), E->obj_part, E->obj_count, E->H, E->np,
                           lam, (long long)j, (long long)min_iter, tol1, tol2, E->state, E->obj_hist, E->Hhi, E->Hlo);
    else
        hipLaunchKernelGGL((mur_h_update_bf16_kernel<KP, false>), grid, block, shm, E->stream, E->xf32, 1,
                           E->xf32 + (int64_t)E->kp * E->np, 1, E->xf64, (int64_t)1, E->H, E->np,
                           lam, (long long)j, (long long)min_iter, tol1, tol2, E->state, E->obj_hist, E->Hhi, E->Hlo,
                           E->xworld > 0 ? E->xf32 + (int64_t)E->kp * E->np + (int64_t)E->kp * E->kp + E->kp : (const float*)nullptr, E->xworld,
                           0.f, cb0, sliced ? E->xf32 : (float*)nullptr);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int KP>
static int launch_h_unpack_bf16(nmfx_engine* E, int skip0, int skip1) {
    ProfScope ps(E, "h_unpack");
    const int nblk = (int)(E->np / 64) - (skip1 - skip0);
    if (nblk <= 0) return NMFX_OK;
    hipLaunchKernelGGL((mur_h_unpack_bf16_kernel<KP>), dim3((unsigned)nblk), dim3(512), 0, E->stream, E->xf32, E->H, E->Hhi, E->Hlo,
                       E->np, skip0, skip1, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// MUR-Euclidean, k = 128: split-bf16 products, the same epilogue kernels as k = 64 (they also write the
// images of the new factors), Gram matrices from the Gram kernels (no by-product at this width)
// between them, and one image pass per factor update.
// ---------------------------------------------------------------------------------------------------------------------
// W^T W from the transposed bf16 images of W (WThi / WTlo [kp][mp], written by the W epilogue) for k padded to 128, where
// the product kernel has no Gram by-product.  The exact-f32 gram_tn_kernel reads W eight times through L2 (one block per
// tile ROW of G and row split) with 4-byte loads: 124 us at 131072 rows.  Here a block takes a range of rows and ALL 64
// tiles: wave w = tile row w; per step of 64 rows every wave loads ITS OWN tile row (the lane's 16 bytes are its A
// fragment as they stand), publishes it in LDS (chunk c of factor row f at c ^ (f >> 1 & 7): conflict free for the
// fragment reads), and takes the eight B fragments from there; four-term split products; double-buffered stage, one
// barrier per step.  Up to 256 row splits, folded to E->gsplit slabs by a second small launch.
// ---------------------------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(KP * 4) void gram_tn_bf16_kernel(
    const unsigned short* __restrict__ Thi, const unsigned short* __restrict__ Tlo, int64_t ld, int64_t nsteps,
    float* __restrict__ out, const int* __restrict__ flag)
{
    if (*flag) return;
    constexpr int NB = KP / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char gst[];     // [2 buffers][2 images][KP][128 B]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, x = lane & 15, g = lane >> 4;
    const int S = gridDim.x, s = blockIdx.x;
    const int64_t t0 = nsteps * s / S, t1 = nsteps * (s + 1) / S;
    f32x4 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int swz = (x >> 1) & 7;
    const unsigned short* ph = Thi + (int64_t)(16 * w + x) * ld + 8 * g;
    const unsigned short* pl = Tlo + (int64_t)(16 * w + x) * ld + 8 * g;
    Frag8 nh[2], nl[2];
    if (t0 < t1) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            nh[ks].u = *reinterpret_cast<const uint4*>(ph + 64 * t0 + 32 * ks);
            nl[ks].u = *reinterpret_cast<const uint4*>(pl + 64 * t0 + 32 * ks);
        }
    }
    for (int64_t t = t0; t < t1; ++t) {
        unsigned char* bufh = gst + (size_t)((t - t0) & 1) * (2 * KP * 128);
        unsigned char* bufl = bufh + KP * 128;
        Frag8 ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ah[ks] = nh[ks]; al[ks] = nl[ks];
            const int off = (16 * w + x) * 128 + 16 * ((4 * ks + g) ^ swz);
            *reinterpret_cast<uint4*>(bufh + off) = ah[ks].u;
            *reinterpret_cast<uint4*>(bufl + off) = al[ks].u;
        }
        __syncthreads();
        if (t + 1 < t1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                nh[ks].u = *reinterpret_cast<const uint4*>(ph + 64 * (t + 1) + 32 * ks);
                nl[ks].u = *reinterpret_cast<const uint4*>(pl + 64 * (t + 1) + 32 * ks);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int off = (16 * j + x) * 128 + 16 * ((4 * ks + g) ^ swz);
                Frag8 bh, bl;
                bh.u = *reinterpret_cast<const uint4*>(bufh + off);
                bl.u = *reinterpret_cast<const uint4*>(bufl + off);
                acc[j] = MFMA_BF16(ah[ks], bh, acc[j]);
                acc[j] = MFMA_BF16(al[ks], bh, acc[j]);
                acc[j] = MFMA_BF16(ah[ks], bl, acc[j]);
                acc[j] = MFMA_BF16(al[ks], bl, acc[j]);
            }
        }
    }
    float* o = out + (int64_t)s * KP * KP;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(int64_t)(16 * w + 4 * g + r) * KP + 16 * j + x] = acc[j][r];
}

// out[f] = sum over p = f, f + F, f + 2 F, ... of part[p]   (slabs of `count` floats; fixed order)
__global__ __launch_bounds__(256) void fold_slabs_kernel(const float* __restrict__ part, int slabs, int64_t count, int F,
                                                         float* __restrict__ out, const int* __restrict__ flag)
{
    if (*flag) return;
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (i4 * 4 >= count) return;
    float4 a = *reinterpret_cast<const float4*>(part + (int64_t)f * count + i4 * 4);
    for (int p = f + F; p < slabs; p += F) {
        const float4 t = *reinterpret_cast<const float4*>(part + (int64_t)p * count + i4 * 4);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    *reinterpret_cast<float4*>(out + (int64_t)f * count + i4 * 4) = a;
}

// (hi, lo: bf16 images [kp][ld] of a factor with `ld` rows / columns; out: the slab buffer of its Gram matrix)
static int bf16_gram_images(nmfx_engine* E, const unsigned short* hi, const unsigned short* lo, int64_t ld, float* out, int* slabs);

int nmfx_bf16_gram_tn(nmfx_engine* E, int* slabs) {
    ProfScope ps(E, "gram_tn");
    return bf16_gram_images(E, E->WThi, E->WTlo, E->mp, E->G_part, slabs);
}

// H H^T from the images of H (k padded to 128), into HHt_part: the same kernel -- Hhi / Hlo are laid out like the transposed
// images of W
int nmfx_bf16_gram_h(nmfx_engine* E, int* slabs) {
    ProfScope ps(E, "gram_nt");
    return bf16_gram_images(E, E->Hhi, E->Hlo, E->np, E->HHt_part, slabs);
}

static int bf16_gram_images(nmfx_engine* E, const unsigned short* hi, const unsigned short* lo, int64_t ld, float* out, int* slabs) {
    if (E->kp != 128) { E->err = "bf16_gram: k padded to 128 only"; return NMFX_E_ARG; }
    constexpr int KP = 128;
    int rc;
    const int64_t nsteps = ld / 64, kk = (int64_t)KP * KP;
    int S = (int)std::min<int64_t>(256, std::max<int64_t>(1, nsteps / 4));
    const size_t shm = (size_t)2 * 2 * KP * 128;
    auto kern = gram_tn_bf16_kernel<KP>;
    if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm))) return rc;
    const int keep = std::max(*slabs, 1);              // in: the most slabs the consumer wants to sum itself; out: what it gets
    const bool fold = S > keep;
    float* dst = out;
    if (fold) {
        if ((rc = lazy_alloc(E, &E->G_big, (int64_t)256 * kk))) return rc;
        dst = E->G_big;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)S), dim3(KP * 4), shm, E->stream, hi, lo, ld, nsteps, dst, &E->state->flag);
    NMFX_HIP(hipGetLastError());
    if (fold) {
        const int F = std::min(keep, E->gsplit);
        hipLaunchKernelGGL(fold_slabs_kernel, dim3((unsigned)((kk / 4 + 255) / 256), (unsigned)F), dim3(256), 0, E->stream,
                           E->G_big, S, kk, F, out, &E->state->flag);
        NMFX_HIP(hipGetLastError());
        S = F;
    }
    *slabs = S;
    return NMFX_OK;
}

static int mur_eu_phase_a_bf16_k128(nmfx_engine* E, double lambda_w, int64_t j) {
    int rc;
    const int cur = (int)(j & 1), nxt = cur ^ 1;
    const int64_t kk = (int64_t)E->kp * E->kp;
    { ProfScope ps(E, "sum_hht");
      if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc; }
    if ((rc = nmfx_bf16_vht(E, true, cur, "wphase", false, 3))) return rc;
    if ((rc = launch_w_update_bf16<128>(E, E->W[cur], E->W[nxt], nxt, E->HHt, 1, (float)lambda_w))) return rc;
    int gslabs = E->gsplit;                            // W^T W from the images the epilogue above has just written
    if ((rc = nmfx_bf16_gram_tn(E, &gslabs))) return rc;
    if ((rc = nmfx_bf16_vtw(E, false, "hphase", false, 3))) return rc;
    return nmfx_launch_pack_from(E, E->Bt_part, E->bt_split, E->G_part, gslabs, (int64_t)(E->mp / 128) * E->bf_wsplit);
}

static int mur_eu_phase_b_bf16_k128(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    if ((rc = launch_h_update_bf16<128>(E, false, (float)lambda_h, j, min_iter, tol1, tol2))) return rc;
    int hslabs = E->gsplit;                            // H H^T from the images the epilogue above has just written (sum_hht adds gsplit slabs)
    if ((rc = nmfx_bf16_gram_h(E, &hslabs))) return rc;
    if (hslabs < E->gsplit)                            // (short H: fewer slabs than sum_hht reads -- the rest must be zero)
        NMFX_HIP(hipMemsetAsync(E->HHt_part + (int64_t)hslabs * E->kp * E->kp, 0, (size_t)(E->gsplit - hslabs) * E->kp * E->kp * sizeof(float), E->stream));
    return NMFX_OK;
}

int nmfx_mur_eu_phase_a_bf16(nmfx_engine* E, double lambda_w, int64_t j) {
    int rc;
    if (!E->bf_ready) E->wsel = (int)(j & 1);   // the images of the current W are built from W[j & 1]
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if (E->kp != 64) return mur_eu_phase_a_bf16_k128(E, lambda_w, j);
    const int cur = (int)(j & 1), nxt = cur ^ 1;
    const float* Wold = E->W[cur];
    float* Wnew = E->W[nxt];
    // W phase: A = V H^T, residual objective of (W_j, H_j), and H H^T as a by-product
    static const int nw4 = getenv("NMFX_NW4") ? atoi(getenv("NMFX_NW4")) : 0;      // bit 0: W phase, bit 1: H phase on four-wave blocks
    E->xyt_nw = (nw4 & 1) ? 4 : 8;
    rc = launch_xyt(E, true, E->Vtile, true, E->np, E->mp, (int)(E->np / 64), E->bf_wsplit, E->Hhi, E->Hlo, E->np,
                    E->Whi[cur], E->Wlo[cur], E->A_part, E->HHt_part, "wphase", false, E->gram_ng_w, 3);
    E->xyt_nw = 8;
    if (rc) return rc;
    if ((rc = launch_w_update_bf16<64>(E, Wold, Wnew, nxt, E->HHt_part, nmfx_bf16_hht_slabs(E), (float)lambda_w))) return rc;
    // H phase: B^T = V^T W_new, and W_new^T W_new as a by-product
    E->xyt_nw = (nw4 & 2) ? 4 : 8;
    rc = launch_xyt(E, false, E->Vt, true, E->mp, E->np, (int)(E->mp / 64), E->bt_split, E->WThi, E->WTlo, E->mp,
                    nullptr, nullptr, E->Bt_part, E->G_part, "hphase", false, E->gram_ng_h, 3);
    E->xyt_nw = 8;
    if (rc) return rc;
    if (E->fused_pack) return NMFX_OK;          // single GPU: h_update reads the slabs itself
    return nmfx_launch_pack_from(E, E->Bt_part, E->bt_split, E->G_part, nmfx_bf16_g_slabs(E), E->obj_count);
}

int nmfx_mur_eu_phase_b_bf16(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2,
                             int64_t j) {
    if (E->kp != 64) return mur_eu_phase_b_bf16_k128(E, lambda_h, min_iter, tol1, tol2, j);
    return launch_h_update_bf16<64>(E, E->fused_pack, (float)lambda_h, j, min_iter, tol1, tol2);
}

// Phase B in two parts for the reduce-scatter / all-gather exchange (r5): `slice` updates the column blocks
// [cb0, cb0 + nblk) of H from this rank's reduce-scattered range of the exchange buffer (the whole Gram matrix and the objective
// digits behind it are all-reduced as before) and leaves the new tiles in that range; after the all-gather `rest` brings the
// other ranks' tiles into H and its images and does what follows the update (k = 128: H H^T from the images).
int nmfx_mur_eu_phase_b_slice_bf16(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j, int cb0, int nblk) {
    if (E->kp != 64) return launch_h_update_bf16<128>(E, false, (float)lambda_h, j, min_iter, tol1, tol2, cb0, nblk);
    return launch_h_update_bf16<64>(E, false, (float)lambda_h, j, min_iter, tol1, tol2, cb0, nblk);
}

int nmfx_mur_eu_phase_b_rest_bf16(nmfx_engine* E, int cb0, int nblk) {
    int rc;
    if (E->kp == 64) return launch_h_unpack_bf16<64>(E, cb0, cb0 + nblk);
    if ((rc = launch_h_unpack_bf16<128>(E, cb0, cb0 + nblk))) return rc;
    int hslabs = E->gsplit;                            // (as mur_eu_phase_b_bf16_k128)
    if ((rc = nmfx_bf16_gram_h(E, &hslabs))) return rc;
    if (hslabs < E->gsplit)
        NMFX_HIP(hipMemsetAsync(E->HHt_part + (int64_t)hslabs * E->kp * E->kp, 0, (size_t)(E->gsplit - hslabs) * E->kp * E->kp * sizeof(float), E->stream));
    return NMFX_OK;
}

// ---- phase A in pieces: an exchange that overlaps with the H-side product ("column chunks") ----------------
// The exchange buffer of the split-bf16 path is [column][factor], so the columns [c0, c1) of V are the contiguous range
// xf32[c0 kp, c1 kp): the caller can hand that range to the collective while the product of the next range runs.
//   head:  everything of phase A in front of the H-side product (W phase, W epilogue; k = 128: W^T W from the images)
//   cols:  B^T = V[:, c0:c1]^T W -- a row range of the V^T copy, i.e. a launch of the same product kernel with fewer row
//          blocks and MORE reduction splits (so that the grid still fills the CUs), its slabs in a buffer of their own --
//          and the pack of that range; the range that ends at the padded n also packs W^T W, the objective and the tail.
// Results differ from the one-piece phase A by the summation order of the splits only.
__global__ __launch_bounds__(256) void mur_pack_cols_kernel(
    const float* __restrict__ Bpart, int bsplit, int64_t bcount, float* __restrict__ bout,
    const float* __restrict__ Gpart, int gsplit, int64_t gcount, float* __restrict__ gout,
    const double* __restrict__ objpart, int64_t nobj, double* __restrict__ xf64, int nb, int with_g,
    const int* __restrict__ flag, float* __restrict__ xtail, int xrank, int xworld)
{
    if (*flag) return;
    __shared__ double sh[4];
    const int b = blockIdx.x;
    if (b < nb) {
        for (int64_t i4 = (int64_t)b * 256 + threadIdx.x; i4 * 4 < bcount; i4 += (int64_t)nb * 256) {
            float4 s = *reinterpret_cast<const float4*>(Bpart + i4 * 4);
            for (int p = 1; p < bsplit; ++p) {
                const float4 t = *reinterpret_cast<const float4*>(Bpart + (int64_t)p * bcount + i4 * 4);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            *reinterpret_cast<float4*>(bout + i4 * 4) = s;
        }
    } else if (!with_g) {
        return;
    } else if (b < nb + (int)((gcount + 255) / 256)) {
        const int64_t i = (int64_t)(b - nb) * 256 + threadIdx.x;
        if (i < gcount) {
            float s = Gpart[i];
            for (int p = 1; p < gsplit; ++p) s += Gpart[(int64_t)p * gcount + i];
            gout[i] = s;
        }
    } else {
        double t = 0.0;
        for (int64_t i = threadIdx.x; i < nobj; i += 256) t += objpart[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
        __syncthreads();
        t = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        if (threadIdx.x == 0) xf64[0] = t;
        if (xtail) {                                   // NMFX_XTAIL: own slot = the four 16-bit digits of t, zeros elsewhere
            const unsigned long long tbits = (unsigned long long)__double_as_longlong(t);
            for (int i = threadIdx.x; i < 4 * xworld; i += blockDim.x)
                xtail[i] = (i >> 2) == xrank ? (float)((tbits >> (16 * (i & 3))) & 0xffffull) : 0.f;
        }
    }
}

int nmfx_mur_eu_phase_a_head_bf16(nmfx_engine* E, double lambda_w, int64_t j) {
    int rc;
    if (!E->bf_ready) E->wsel = (int)(j & 1);
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    const int cur = (int)(j & 1), nxt = cur ^ 1;
    if (E->kp != 64) {
        { ProfScope ps(E, "sum_hht");
          if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, (int64_t)E->kp * E->kp, E->HHt))) return rc; }
        if ((rc = nmfx_bf16_vht(E, true, cur, "wphase", false, 3))) return rc;
        if ((rc = launch_w_update_bf16<128>(E, E->W[cur], E->W[nxt], nxt, E->HHt, 1, (float)lambda_w))) return rc;
        E->chunk_gslabs = E->gsplit;
        return nmfx_bf16_gram_tn(E, &E->chunk_gslabs);
    }
    if ((rc = launch_xyt(E, true, E->Vtile, true, E->np, E->mp, (int)(E->np / 64), E->bf_wsplit, E->Hhi, E->Hlo, E->np,
                         E->Whi[cur], E->Wlo[cur], E->A_part, E->HHt_part, "wphase", false, E->gram_ng_w, 3))) return rc;
    return launch_w_update_bf16<64>(E, E->W[cur], E->W[nxt], nxt, E->HHt_part, nmfx_bf16_hht_slabs(E), (float)lambda_w);
}

int nmfx_mur_eu_phase_a_cols_bf16(nmfx_engine* E, int64_t c0, int64_t c1) {
    int rc;
    if (!E->bf_ready) { E->err = "phase_a_cols: call nmfx_mur_phase_a_head first"; return NMFX_E_STATE; }
    if (c0 < 0 || c1 <= c0 || c1 > E->np || c0 % 128 || c1 % 128 || (c0 > 0 && c1 - c0 < 512 && c1 - c0 < E->np)) {
        E->err = "phase_a_cols: column range must be a multiple of 128 inside the padded n (at least 512 columns)"; return NMFX_E_ARG; }
    const int64_t R = c1 - c0, rblocks = R / 128, mgroups = E->mp / 64;
    // reduction splits of this launch: enough blocks for the CUs, at least 4 groups of 64 rows per block
    int64_t S = std::max<int64_t>(E->bt_split, ((int64_t)E->ncu + rblocks / 2) / rblocks);
    S = std::max<int64_t>(1, std::min<int64_t>(S, std::max<int64_t>(1, mgroups / 4)));
    const int64_t need = S * R * E->kp;
    if (need > E->Bt_chunk_cap) {
        if (E->Bt_chunk) { NMFX_HIP(hipStreamSynchronize(E->stream)); NMFX_HIP(hipFree(E->Bt_chunk)); E->Bt_chunk = nullptr; }
        NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->Bt_chunk), (size_t)need * sizeof(float)));
        E->Bt_chunk_cap = need;
    }
    const float* X = E->Vt + (c0 / 128) * mgroups * 8192;             // tile-major V^T: row block c0 / 128
    const int ng = (E->kp == 64 && c0 == 0) ? (int)std::min<int64_t>(E->gram_ng_h, rblocks) : 0;
    if ((rc = launch_xyt(E, false, X, true, E->mp, R, (int)mgroups, (int)S, E->WThi, E->WTlo, E->mp, nullptr, nullptr,
                         E->Bt_chunk, E->kp == 64 ? E->G_part : nullptr, "hphase", false, ng, 3))) return rc;
    if (E->kp == 64 && c0 == 0) E->chunk_gslabs = ng * (int)S;
    ProfScope ps(E, "pack");
    const bool last = c1 == E->np;
    const int nb = (int)std::min<int64_t>(256, (R * E->kp / 4 + 255) / 256);
    const int64_t kk = (int64_t)E->kp * E->kp;
    float* tail = E->xworld > 0 ? E->xf32 + (int64_t)E->kp * E->np + kk + E->kp : nullptr;
    hipLaunchKernelGGL(mur_pack_cols_kernel, dim3((unsigned)(nb + (last ? (int)((kk + 255) / 256) + 1 : 0))), dim3(256), 0, E->stream,
                       E->Bt_chunk, (int)S, R * E->kp, E->xf32 + c0 * E->kp, E->G_part, E->chunk_gslabs, kk,
                       E->xf32 + (int64_t)E->kp * E->np, E->obj_part, (int64_t)(E->mp / 128) * E->bf_wsplit, E->xf64, nb,
                       last ? 1 : 0, &E->state->flag, tail, E->xrank, E->xworld);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- pair mode: two MUR-Euclidean problems in one pass over V (the reference author's parameter grids,
// nmf/nmf_old.py:52-66 / nmf/nmf.py:38-45, run one factorization per (lambda_w, lambda_h, seed) on the SAME data) ----------------
// A handle created with k = 128 holds problem 0 in the factor columns [0, 64) and problem 1 in [64, 128) (each with k_p <= 64,
// zero padded: a zero column of W with a zero row of H is a fixed point of the multiplicative updates).  A = V [H_0; H_1]^T and
// B = [W_0 W_1]^T V are the k = 128 products as they stand -- V and V^T are streamed ONCE for both problems -- while everything
// that couples factors is taken per problem: the residual objective (two sums, xyt32_bf16_kernel<..., NPROB = 2>), the Gram
// matrices (block diagonal), lambda, the stop test and the objective history (slots 2 j + p).  A problem whose stop rule has
// fired keeps the iterate the reference returns while the other one goes on (see mur_w_update_bf16_kernel).
static int pair_ready(nmfx_engine* E, int64_t first, int64_t count) {
    if (!E) return NMFX_E_ARG;
    E->anls_a_ready = false; E->himg_both = false; E->kl_h_iter = -2;
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    if (first < 0 || count < 0) { E->err = "negative iteration range"; return NMFX_E_ARG; }
    if (E->kp != 128 || !(E->precision == 1 && nmfx_bf16_supported(E))) {
        E->err = "pair mode needs a handle created with k = 128 (two problems of k <= 64) on the split-bf16 path"; return NMFX_E_STATE; }
    if (E->xworld > 0 || E->comm) { E->err = "pair mode is a single-GPU form"; return NMFX_E_STATE; }
    int rc;
    if ((rc = nmfx_enter_family(E, 1))) return rc;
    if (!E->pair && E->family_started) { E->err = "pair mode cannot continue a run that was started as one k = 128 problem (nmfx_set_factors first)"; return NMFX_E_STATE; }
    E->pair = true; E->family_started = true;
    NMFX_HIP(hipSetDevice(E->device));
    return nmfx_ensure_obj_capacity(E, 2 * (first + count) + 8);
}

__global__ __launch_bounds__(256) void pair_finalize_kernel(const double* __restrict__ osrc, int64_t nobj, long long j, long long min_iter,
                                                            double tol1, double tol2, DevState* __restrict__ st, double* __restrict__ obj_hist)
{
    if (st->flag) return;
    __shared__ double sh[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a = 0.0, b = 0.0;
    for (int64_t i = tid; i < nobj; i += 256) { a += osrc[i]; b += osrc[nobj + i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; }
    __syncthreads();
    if (tid == 0) {
        for (int p = 0; p < 2; ++p)
            if (!st->pflag[p]) nmfx_record_objective_pair(st, obj_hist, ((sh[p][0] + sh[p][1]) + sh[p][2]) + sh[p][3], p, j, min_iter, tol1, tol2, true);
        if (st->pflag[0] && st->pflag[1]) st->flag = 1;
    }
}

static int pair_iteration(nmfx_engine* E, const double* lw, const double* lh, int64_t min_iter, double tol1, double tol2, int64_t j) {
    int rc;
    if (!E->bf_ready) E->wsel = (int)(j & 1);
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    const int cur = (int)(j & 1), nxt = cur ^ 1;
    const int64_t kk = (int64_t)E->kp * E->kp;
    const int64_t nobj = (int64_t)(E->mp / 128) * E->bf_wsplit;
    { ProfScope ps(E, "sum_hht");
      if ((rc = nmfx_launch_sum_partials(E, E->HHt_part, E->gsplit, kk, E->HHt))) return rc; }
    if ((rc = nmfx_bf16_vht(E, true, cur, "wphase", false, 3))) return rc;          // (E->pair: two objective partials per block)
    { ProfScope ps(E, "w_update");
      constexpr int KP = 128;
      constexpr size_t shm = (size_t)(KP * (KP + 16) + 64 * (KP + 4)) * sizeof(float) + (size_t)2 * KP * 66 * sizeof(unsigned short);
      auto kern = mur_w_update_bf16_kernel<KP, true>;
      if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm))) return rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)(E->mp / 64)), dim3(512), shm, E->stream, E->A_part, E->bf_wsplit, E->mp,
                         (const float*)E->W[cur], (const float*)E->HHt, 1, (float)lw[0], E->W[nxt], E->Whi[nxt], E->Wlo[nxt], E->WThi, E->WTlo,
                         (const int*)&E->state->flag, (float)lw[1], (const DevState*)E->state, nxt);
      NMFX_HIP(hipGetLastError()); }
    int gslabs = E->gsplit;
    if ((rc = nmfx_bf16_gram_tn(E, &gslabs))) return rc;
    if ((rc = nmfx_bf16_vtw(E, false, "hphase", false, 3))) return rc;
    if ((rc = nmfx_launch_pack_from(E, E->Bt_part, E->bt_split, E->G_part, gslabs, nobj))) return rc;
    E->wsel = (int)((j + 1) & 1);
    E->w_in_place = false;
    { ProfScope ps(E, "h_update");
      constexpr int KP = 128;
      constexpr size_t shm = (size_t)(KP * (KP + 4) + KP * 80 + 64 * (KP + 4)) * sizeof(float);
      auto kern = mur_h_update_bf16_kernel<KP, false, true>;
      if ((rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(kern), (int)shm))) return rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)(E->np / 64)), dim3(512), shm, E->stream, (const float*)E->xf32, 1,
                         (const float*)(E->xf32 + (int64_t)E->kp * E->np), 1, (const double*)E->obj_part, nobj, E->H, E->np, (float)lh[0],
                         (long long)j, (long long)min_iter, tol1, tol2, E->state, E->obj_hist, E->Hhi, E->Hlo,
                         (const float*)nullptr, 0, (float)lh[1], 0, (float*)nullptr);
      NMFX_HIP(hipGetLastError()); }
    int hslabs = E->gsplit;
    if ((rc = nmfx_bf16_gram_h(E, &hslabs))) return rc;
    if (hslabs < E->gsplit)
        NMFX_HIP(hipMemsetAsync(E->HHt_part + (int64_t)hslabs * kk, 0, (size_t)(E->gsplit - hslabs) * kk * sizeof(float), E->stream));
    return NMFX_OK;
}

extern "C" int nmfx_mur_pair_run(nmfx_handle_t E, const double* lambda_w, const double* lambda_h, int64_t min_iter, double tol1,
                                 double tol2, int64_t first, int64_t count) {
    if (!E || !lambda_w || !lambda_h) { if (E) E->err = "mur_pair_run: lambda_w[2], lambda_h[2]"; return NMFX_E_ARG; }
    int rc = pair_ready(E, first, count); if (rc) return rc;
    for (int64_t j = first; j < first + count; ++j)
        if ((rc = pair_iteration(E, lambda_w, lambda_h, min_iter, tol1, tol2, j))) return rc;
    return NMFX_OK;
}

// objective of the pair(s) the last iteration left, and the last evaluation of the stop rule (nmf/mur.py:127-131 for i = max_iter - 1)
extern "C" int nmfx_mur_pair_finish(nmfx_handle_t E, int64_t min_iter, double tol1, double tol2, int64_t iters_done) {
    int rc = pair_ready(E, iters_done, 1); if (rc) return rc;
    if (!E->bf_ready) E->wsel = (int)(iters_done & 1);
    if ((rc = nmfx_bf16_prepare(E))) return rc;
    if ((rc = nmfx_bf16_vht(E, true, (int)(iters_done & 1), "wphase", false, 3))) return rc;      // (its A output is scratch here)
    hipLaunchKernelGGL(pair_finalize_kernel, dim3(1), dim3(256), 0, E->stream, (const double*)E->obj_part,
                       (int64_t)(E->mp / 128) * E->bf_wsplit, (long long)iters_done, (long long)min_iter, tol1, tol2, E->state, E->obj_hist);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

static int pair_state(nmfx_engine* E, DevState* hs) {
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_HIP(hipMemcpyAsync(hs, E->state, sizeof(DevState), hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

extern "C" int nmfx_pair_get_state(nmfx_handle_t E, int p, int* stop_rule, int64_t* stop_i, int64_t* n_obj) {
    if (!E || (p != 0 && p != 1)) { if (E) E->err = "pair: problem index 0 or 1"; return NMFX_E_ARG; }
    DevState hs; int rc;
    if ((rc = pair_state(E, &hs))) return rc;
    if (stop_rule) *stop_rule = hs.pflag[p];
    if (stop_i) *stop_i = hs.pstop_i[p];
    if (n_obj) *n_obj = hs.pn_obj[p];
    return NMFX_OK;
}

extern "C" int nmfx_pair_get_objectives(nmfx_handle_t E, int p, int64_t first, int64_t count, double* out) {
    if (!E || !out || (p != 0 && p != 1) || first < 0 || count < 0 || 2 * (first + count) > E->obj_cap) {
        if (E) E->err = "pair_get_objectives: range"; return NMFX_E_ARG; }
    if (count == 0) return NMFX_OK;
    NMFX_HIP(hipSetDevice(E->device));
    std::vector<double> tmp((size_t)(2 * count));
    NMFX_HIP(hipMemcpyAsync(tmp.data(), E->obj_hist + 2 * first, tmp.size() * 8, hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    for (int64_t i = 0; i < count; ++i) out[i] = tmp[(size_t)(2 * i + p)];
    return NMFX_OK;
}

// factors of problem p (k_p <= 64 columns of W / rows of H from offset 64 p): the iterate the reference returns -- W_{stop_i + 1}
// once the problem's stop rule has fired, else the latest one
extern "C" int nmfx_pair_get_factors(nmfx_handle_t E, int p, int k_p, double* w, double* hmat) {
    if (!E || (p != 0 && p != 1) || k_p < 1 || k_p > 64) { if (E) E->err = "pair_get_factors: p in {0, 1}, 1 <= k_p <= 64"; return NMFX_E_ARG; }
    if (E->kp != 128) { E->err = "pair_get_factors: handle was not created with k = 128"; return NMFX_E_STATE; }
    DevState hs; int rc;
    if ((rc = pair_state(E, &hs))) return rc;
    const int buf = hs.pflag[p] ? (int)((hs.pstop_i[p] + 1) & 1) : E->wsel;
    if (w) {
        std::vector<float> tmp((size_t)E->m * k_p);
        NMFX_HIP(hipMemcpy2DAsync(tmp.data(), (size_t)k_p * 4, E->W[buf] + 64 * p, (size_t)E->kp * 4, (size_t)k_p * 4, (size_t)E->m,
                                  hipMemcpyDeviceToHost, E->stream));
        NMFX_HIP(hipStreamSynchronize(E->stream));
        for (size_t i = 0; i < tmp.size(); ++i) w[i] = (double)tmp[i];
    }
    if (hmat) {
        std::vector<float> tmp((size_t)k_p * E->n);
        NMFX_HIP(hipMemcpy2DAsync(tmp.data(), (size_t)E->n * 4, E->H + (int64_t)64 * p * E->np, (size_t)E->np * 4, (size_t)E->n * 4, (size_t)k_p,
                                  hipMemcpyDeviceToHost, E->stream));
        NMFX_HIP(hipStreamSynchronize(E->stream));
        for (size_t i = 0; i < tmp.size(); ++i) hmat[i] = (double)tmp[i];
    }
    return NMFX_OK;
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_bf16() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(transpose_tiled_kernel)) == hipSuccess ? 0 : -1; }
