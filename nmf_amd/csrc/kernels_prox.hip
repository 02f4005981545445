// prox 'l1inf' and 'l1inf_transpose' of ADMM, exactly as written in the reference
// (nmf/admm.py:158-183 and :185-210), on the device.
//
// Both act on mat_aux (k x cols: h_aux, or w_aux^T) and the dual of the same shape:
//   pos = max(aux + dual - lambda/rho, 0)
//   for every vector (a ROW of mat_aux for 'l1inf', a COLUMN for 'l1inf_transpose'):
//     sum(pos) <= upper_bound            ->  out = pos
//     otherwise  val = sort_desc(aux - dual)          [transpose: aux[:, i] - dual[:, 1], admm.py:196]
//                first j >= 1 with  rho val[j-1] + lambda - rho/j (sum(val[:j]) + lambda/rho - upper_bound) < 0
//                count = j - 1   (no such j: count = len + 1)
//                theta = rho / count (sum(val[:count + 1]) + lambda/rho - upper_bound)   [transpose: max(theta, 0)]
//                out = max(aux + dual - lambda/rho - theta/rho, 0)
// The quirks are the reference's (the shifted vector uses aux + dual, the sorted one aux - dual; the count is one
// less than the first failing index; column 1 of the dual for every column) and are kept: the operator is pinned at
// function level by the reference's own outputs (tests/golden/functions.npz) and over the first iterations of
// admm (tests/golden/admm_eu_l1inf*.npz).  As written it makes the ADMM iteration diverge -- that is
// the reference's behaviour too.
//
// 'l1inf': k vectors of n (or m) entries -> one 1024-thread workgroup per vector, the keys sorted in LDS (bitonic,
// up to 32768 entries = 128 KiB), prefix sums and the sign test in f64.  'l1inf_transpose': n (or m) vectors of
// k <= 128 entries -> one wavefront per vector.  Both finish with the dual update U += X - AUX of the ADMM loop
// (admm.py:321-322) when asked to.
#include "nmfx_internal.h"
#include <climits>

namespace {

constexpr int ROWS_NT = 1024;

__device__ __forceinline__ double block_sum_1024(double v, double* red /* [16] */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < ROWS_NT / 64; ++w) t += red[w];      // fixed order, the same in every thread
    return t;
}

// vector `blockIdx.x`: element e at base + e * es (aux, X and U share the layout).
// GLOBALK (r3, vectors beyond 32768 entries -- W-side vectors of a tall matrix): the keys of vector b are sorted in the global
// work area gkeys[b][Lpad] instead of LDS; the workgroup barrier between the passes orders its own global accesses (one CU, one
// vector cache), the data stays in L2.  Same passes, same result, about a millisecond per vector of 131072 entries.
template <bool GLOBALK>
__global__ __launch_bounds__(ROWS_NT) void prox_l1inf_rows_kernel(
    const float* __restrict__ AUX, float* __restrict__ X, float* __restrict__ U, int64_t vec_stride, int64_t es,
    int L, int Lpad, double rho, double lam, double ub, int update_dual, const int* __restrict__ flag, float* __restrict__ gkeys,
    const DevState* __restrict__ st,      // st != nullptr (AO-ADMM, r4): rho = trace(G) / k of the sub-problem, no-op once its inner stop fired
    int transpose = 0)                    // 1 (r4, 'l1inf_transpose' beyond 128 components: one workgroup per column instead of one wavefront):
{                                         // the sorted vector takes the dual of vector 1 (admm.py:196), theta is clamped at zero (admm.py:206)
    if (*flag) return;
    if (st) { if (st->inner_stop) return; rho = st->rho; }
    extern __shared__ __attribute__((aligned(16))) float lkeys[];      // [Lpad] (LDS form)
    float* keys = GLOBALK ? gkeys + (int64_t)blockIdx.x * Lpad : lkeys;
    __shared__ double red[ROWS_NT / 64];
    __shared__ double scan[ROWS_NT / 64];
    __shared__ int first_bad;
    __shared__ double s_total;
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * vec_stride;
    const double shift = lam / rho;
    if (tid == 0) { first_bad = INT_MAX; s_total = 0.0; }
    double psum = 0.0;
    for (int e = tid; e < Lpad; e += ROWS_NT) {
        if (e < L) {
            const float a = AUX[base + e * es], d = U[base + e * es];
            const double p = (double)a + (double)d - shift;
            psum += p < 0.0 ? 0.0 : p;
            keys[e] = a - (transpose ? U[vec_stride + e * es] : d);
        } else keys[e] = -__builtin_inff();
    }
    const double total = block_sum_1024(psum, red);
    double theta_over_rho = 0.0;
    if (total > ub) {                                   // (block-uniform)
        for (int k2 = 2; k2 <= Lpad; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                __syncthreads();
                for (int i = tid; i < Lpad; i += ROWS_NT) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const float a = keys[i], b = keys[ixj];
                        const bool desc = (i & k2) == 0;
                        if ((a < b) == desc) { keys[i] = b; keys[ixj] = a; }
                    }
                }
            }
        __syncthreads();
        // prefix sums: thread t owns the contiguous chunk [t C, (t + 1) C)
        const int C = (L + ROWS_NT - 1) / ROWS_NT;
        const int c0 = min(tid * C, L), c1 = min(c0 + C, L);
        double local = 0.0;
        for (int e = c0; e < c1; ++e) local += (double)keys[e];
        // exclusive scan over the 1024 chunk sums: within the wave by shuffles, across waves through LDS
        double incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double t = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += t;
        }
        if ((tid & 63) == 63) scan[tid >> 6] = incl;
        __syncthreads();
        double before = 0.0;
        for (int w = 0; w < (tid >> 6); ++w) before += scan[w];
        double run = before + incl - local;             // sum of the entries in front of this chunk
        const double run0 = run;
        for (int e = c0; e < c1; ++e) {
            const double val = (double)keys[e];
            run += val;
            const double test = rho * val + lam - rho / (double)(e + 1) * (run + shift - ub);
            if (test < 0.0) { atomicMin(&first_bad, e); break; }
        }
        __syncthreads();
        const int count = first_bad == INT_MAX ? L + 1 : first_bad;     // first failing 1-based index minus one
        const int tgt = min(count, L - 1);                             // sum(val[:count + 1]) = prefix through entry `count`
        if (tgt >= c0 && tgt < c1) {
            double r2 = run0;
            for (int e = c0; e <= tgt; ++e) r2 += (double)keys[e];
            s_total = r2;
        }
        __syncthreads();
        const double theta = rho / (double)count * (s_total + shift - ub);
        theta_over_rho = (transpose && !(theta > 0.0)) ? 0.0 : theta / rho;
    }
    for (int e = tid; e < L; e += ROWS_NT) {
        const float a = AUX[base + e * es], d = U[base + e * es];
        const double z = (double)a + (double)d - shift - theta_over_rho;
        const float x = z < 0.0 ? 0.f : (float)z;
        X[base + e * es] = x;
        if (update_dual) U[base + e * es] = d + x - a;
    }
}

// one wavefront per vector i: element t at i * vs + t * es; the sorted vector takes the dual of vector 1 (admm.py:196)
__global__ __launch_bounds__(64) void prox_l1inf_cols_kernel(
    const float* __restrict__ AUX, float* __restrict__ X, float* __restrict__ U, int64_t vs, int64_t es, int k,
    double rho, double lam, double ub, int update_dual, const int* __restrict__ flag, const DevState* __restrict__ st)
{
    if (*flag) return;
    if (st) { if (st->inner_stop) return; rho = st->rho; }
    __shared__ float keys[128];
    __shared__ double res[2];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * vs;
    const double shift = lam / rho;
    float a[2], d[2];
    double psum = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        if (t < k) {
            a[h] = AUX[base + t * es]; d[h] = U[base + t * es];
            const double p = (double)a[h] + (double)d[h] - shift;
            psum += p < 0.0 ? 0.0 : p;
            keys[t] = a[h] - U[vs + t * es];             // dual[:, 1]
        } else { a[h] = 0.f; d[h] = 0.f; keys[t] = -__builtin_inff(); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off, 64);
    double theta_over_rho = 0.0;
    if (psum > ub) {                                     // (wave-uniform)
        for (int k2 = 2; k2 <= 128; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                __syncthreads();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = lane + 64 * h, ixj = i ^ j;
                    if (ixj > i) {
                        const float x = keys[i], y = keys[ixj];
                        const bool desc = (i & k2) == 0;
                        if ((x < y) == desc) { keys[i] = y; keys[ixj] = x; }
                    }
                }
            }
        __syncthreads();
        if (lane == 0) {                                 // k <= 128 entries: one lane walks them
            double run = 0.0;
            int count = k + 1;
            for (int e = 0; e < k; ++e) {
                const double val = (double)keys[e];
                run += val;
                const double test = rho * val + lam - rho / (double)(e + 1) * (run + shift - ub);
                if (test < 0.0) { count = e; break; }
            }
            double tot = 0.0;
            for (int e = 0; e <= min(count, k - 1); ++e) tot += (double)keys[e];
            const double theta = rho / (double)count * (tot + shift - ub);
            res[0] = theta > 0.0 ? theta / rho : 0.0;    // admm.py:206
        }
        __syncthreads();
        theta_over_rho = res[0];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        if (t < k) {
            const double z = (double)a[h] + (double)d[h] - shift - theta_over_rho;
            const float x = z < 0.0 ? 0.f : (float)z;
            X[base + t * es] = x;
            if (update_dual) U[base + t * es] = d[h] + x - a[h];
        }
    }
}

// U += X - AUX (admm.py:321-322), all entries (the padding is zero in all three)
__global__ __launch_bounds__(256) void dual_update_kernel(const float* __restrict__ AUX, const float* __restrict__ X,
                                                          float* __restrict__ U, int64_t count, const int* __restrict__ flag,
                                                          const DevState* __restrict__ st)
{
    if (*flag) return;
    if (st && st->inner_stop) return;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;
    const float4 a = *reinterpret_cast<const float4*>(AUX + i), x = *reinterpret_cast<const float4*>(X + i);
    float4 u = *reinterpret_cast<float4*>(U + i);
    u.x += x.x - a.x; u.y += x.y - a.y; u.z += x.z - a.z; u.w += x.w - a.w;
    *reinterpret_cast<float4*>(U + i) = u;
}

}  // namespace

// X = prox(AUX, U) for the W side (mat_aux = w_aux^T, k x m) or the H side (mat_aux = h_aux, k x n)
int nmfx_launch_prox_l1inf(nmfx_engine* E, bool h_side, bool transpose, double rho, double lam, double ub, bool update_dual, bool ao) {
    ProfScope ps(E, "prox_l1inf");
    const DevState* st = ao ? E->state : nullptr;       // (AO-ADMM: rho lives on the device, the rounds stop on their own)
    const float* aux = h_side ? E->auxH : E->auxW;
    float* x = h_side ? E->H : E->W[0];
    float* u = h_side ? E->dualH : E->dualW;
    if (!aux || !u) { E->err = "prox_l1inf: ADMM state not allocated"; return NMFX_E_STATE; }
    // storage: H-like [kp][np] (factor t, column c at t * np + c), W-like [mp][kp] (row r, factor t at r * kp + t)
    const int64_t cols = h_side ? E->n : E->m;
    if (!transpose) {       // a vector = all entries of one factor: k vectors of `cols` entries
        int64_t Lpad = 2; while (Lpad < cols) Lpad <<= 1;
        if (Lpad > (int64_t)1 << 30) { E->err = "prox 'l1inf': vector too long"; return NMFX_E_ARG; }
        if (Lpad <= 32768) {
            const size_t shm = (size_t)Lpad * sizeof(float);
            int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(prox_l1inf_rows_kernel<false>), (int)shm + 1024); if (rc) return rc;
            hipLaunchKernelGGL(prox_l1inf_rows_kernel<false>, dim3((unsigned)E->k), dim3(ROWS_NT), shm, E->stream, aux, x, u,
                               h_side ? E->np : (int64_t)1, h_side ? (int64_t)1 : (int64_t)E->kp, (int)cols, (int)Lpad, rho, lam, ub,
                               update_dual ? 1 : 0, &E->state->flag, (float*)nullptr, st, 0);
        } else {            // longer vectors: the same kernel with its keys in a global work area
            if (E->prox_keys_cap < (int64_t)E->k * Lpad) {
                if (E->prox_keys) { NMFX_HIP(hipStreamSynchronize(E->stream)); hipFree(E->prox_keys); E->prox_keys = nullptr; }
                NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&E->prox_keys), (size_t)E->k * Lpad * sizeof(float)));
                E->prox_keys_cap = (int64_t)E->k * Lpad;
            }
            hipLaunchKernelGGL(prox_l1inf_rows_kernel<true>, dim3((unsigned)E->k), dim3(ROWS_NT), 16, E->stream, aux, x, u,
                               h_side ? E->np : (int64_t)1, h_side ? (int64_t)1 : (int64_t)E->kp, (int)cols, (int)Lpad, rho, lam, ub,
                               update_dual ? 1 : 0, &E->state->flag, E->prox_keys, st, 0);
        }
    } else {                // a vector = the k factors of one column of mat_aux
        if (cols < 2) { E->err = "prox 'l1inf_transpose' reads column 1 of the dual: needs at least 2 columns"; return NMFX_E_ARG; }
        // (every vector reads the dual of vector 1: the dual update is a launch of its own behind this one)
        if (E->k > 128) {   // r4: a column of more than 128 entries takes the one-workgroup-per-vector kernel of 'l1inf' with the transposed strides
            int64_t Lpad = 2; while (Lpad < E->k) Lpad <<= 1;
            const size_t shm = (size_t)Lpad * sizeof(float);
            int rc = nmfx_allow_lds(E, reinterpret_cast<const void*>(prox_l1inf_rows_kernel<false>), (int)shm + 1024); if (rc) return rc;
            hipLaunchKernelGGL(prox_l1inf_rows_kernel<false>, dim3((unsigned)cols), dim3(ROWS_NT), shm, E->stream, aux, x, u,
                               h_side ? (int64_t)1 : (int64_t)E->kp, h_side ? E->np : (int64_t)1, E->k, (int)Lpad, rho, lam, ub,
                               0, &E->state->flag, (float*)nullptr, st, 1);
        } else
        hipLaunchKernelGGL(prox_l1inf_cols_kernel, dim3((unsigned)cols), dim3(64), 0, E->stream, aux, x, u,
                           h_side ? (int64_t)1 : (int64_t)E->kp, h_side ? E->np : (int64_t)1, E->k, rho, lam, ub,
                           0, &E->state->flag, st);
        if (update_dual) {
            const int64_t count = h_side ? (int64_t)E->kp * E->np : E->mp * (int64_t)E->kp;
            hipLaunchKernelGGL(dual_update_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, E->stream, aux, x, u,
                               count, &E->state->flag, st);
        }
    }
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_prox() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(dual_update_kernel)) == hipSuccess ? 0 : -1; }
