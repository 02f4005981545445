// Leading singular triplets of the resident V in f64 -- the device side of NNDSVD
// (nmf/utils.py:36-93 starts from numpy.linalg.svd of the whole matrix and uses the
// first `rank` triplets only, utils.py:50-57).
//
// Method: block subspace iteration with Rayleigh-Ritz on an L-column block (L > k):
//     Y = V Q;   Y^T Y = S Theta S^T;   U = Y S Theta^-1/2,  Qr = Q S,  sigma = sqrt(Theta)
//     Z = V^T U; residual_i = || Z_i - sigma_i Qr_i ||;      Z^T Z = S' Theta' S'^T;  Q = Z S' Theta'^-1/2
// until the residuals of the k leading triplets are below tol * sigma_i.  V stays in its
// f32 device layout (exact in f64), all products accumulate in f64 on v_mfma_f64_16x16x4_f64,
// the L x L eigenproblems run on the host (threshold Jacobi; they are nearly diagonal after
// the first sweep).  Per sweep V is read twice; nothing else is larger than (m + n) x L.
#include "nmfx_internal.h"

#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
// C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg  (NOT the f32 form's 4 * (lane >> 4) + reg)
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

constexpr int KC = 32;            // contraction chunk staged through LDS
constexpr int LDA = 80;           // floats per k-row of the A tile (conflict-free column reads)

// C_part[split][row][j] = sum over the split's contraction range of A[row][kk] * B[kk][j]
//   TRANS = false: A[row][kk] = V[r0 + row][kk]   (out rows = rows of V, contraction = columns)
//   TRANS = true : A[row][kk] = V[kk][c0 + row]   (out rows = columns of V, contraction = rows)
// Block = 64 output rows (4 waves x 16), all L = 16 NT columns; B is [contraction][L] f64.
template <int NT, bool TRANS>
__global__ __launch_bounds__(256) void svd_apply_kernel(
    const float* __restrict__ V, int64_t ldv, int64_t clen, const double* __restrict__ B,
    double* __restrict__ Cpart, int64_t out_rows)
{
    constexpr int L = 16 * NT;
    constexpr int LP = (L % 32 == 0) ? L + 16 : L;      // LP = 16 (mod 32): the four k-rows of a B read hit distinct banks
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* As = reinterpret_cast<float*>(lds_raw);                       // [KC][LDA]   (k-major)
    double* Bs = reinterpret_cast<double*>(lds_raw + KC * LDA * 4);      // [KC][LP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, q = lane >> 4;
    const int64_t o0 = (int64_t)blockIdx.x * 64;
    const int S = gridDim.y, sp = blockIdx.y;
    const int64_t nchunk = clen / KC;
    const int64_t k0 = nchunk * sp / S * KC, k1 = nchunk * (sp + 1) / S * KC;
    f64x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int64_t kk = k0; kk < k1; kk += KC) {
        if (TRANS) {        // V[kk + k][o0 + i]: rows are contiguous in i
            for (int e = tid; e < KC * 16; e += 256) {
                const int k = e >> 4, i4 = e & 15;
                *reinterpret_cast<float4*>(As + k * LDA + 4 * i4) =
                    *reinterpret_cast<const float4*>(V + (kk + k) * ldv + o0 + 4 * i4);
            }
        } else {            // V[o0 + i][kk + k]: rows are contiguous in k; stored transposed
            for (int e = tid; e < 64 * (KC / 4); e += 256) {
                const int i = e / (KC / 4), k4 = e % (KC / 4);
                const float4 v = *reinterpret_cast<const float4*>(V + (o0 + i) * ldv + kk + 4 * k4);
                As[(4 * k4 + 0) * LDA + i] = v.x; As[(4 * k4 + 1) * LDA + i] = v.y;
                As[(4 * k4 + 2) * LDA + i] = v.z; As[(4 * k4 + 3) * LDA + i] = v.w;
            }
        }
        for (int e = tid; e < KC * (L / 2); e += 256) {
            const int k = e / (L / 2), j2 = e % (L / 2);
            *reinterpret_cast<double2*>(Bs + k * LP + 2 * j2) =
                *reinterpret_cast<const double2*>(B + (kk + k) * L + 2 * j2);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KC / 4; ++u) {
            const double a = (double)As[(4 * u + q) * LDA + 16 * wave + x];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = MFMA_F64(a, Bs[(4 * u + q) * LP + 16 * t + x], acc[t]);
        }
        __syncthreads();
    }
    double* out = Cpart + ((int64_t)sp * out_rows + o0 + 16 * wave) * L;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(int64_t)(q + 4 * r) * L + 16 * t + x] = acc[t][r];
}

// out[e] = sum_s part[s][e]   (fixed order)
__global__ __launch_bounds__(256) void svd_sum_kernel(const double* __restrict__ part, int nsplit, int64_t count,
                                                      double* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;
    for (int p = 0; p < nsplit; ++p) s += part[(int64_t)p * count + e];
    out[e] = s;
}

// Tpart[chunk][i][j] = sum over the chunk's rows of X[r][i] X[r][j];  grid (chunks, NT): block
// (chunk, ti) owns tile row ti, wave w the tile columns w, w + 4, ...
template <int NT>
__global__ __launch_bounds__(256) void svd_gram_kernel(const double* __restrict__ X, int64_t rows_per_chunk,
                                                       double* __restrict__ Tpart)
{
    constexpr int L = 16 * NT, NW = (NT + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
    const int ti = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
    f64x4 acc[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) acc[c] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int64_t r = r0; r < r0 + rows_per_chunk; r += 4) {
        const double* row = X + (r + q) * L;
        const double a = row[16 * ti + x];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            const int tj = wave + 4 * c;
            if (tj < NT) acc[c] = MFMA_F64(a, row[16 * tj + x], acc[c]);
        }
    }
    double* out = Tpart + (int64_t)blockIdx.x * L * L;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int tj = wave + 4 * c;
        if (tj < NT)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(int64_t)(16 * ti + q + 4 * r) * L + 16 * tj + x] = acc[c][r];
    }
}

// Xout[r][j] = sum_i Xin[r][i] M[i][j]   (block = 64 rows, M read through L2)
template <int NT>
__global__ __launch_bounds__(256) void svd_rotate_kernel(const double* __restrict__ Xin, const double* __restrict__ M,
                                                         double* __restrict__ Xout)
{
    constexpr int L = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 64 + 16 * wave;
    f64x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double* arow = Xin + (r0 + x) * L;
    for (int u = 0; u < L / 4; ++u) {
        const double a = arow[4 * u + q];
        const double* mrow = M + (int64_t)(4 * u + q) * L;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MFMA_F64(a, mrow[16 * t + x], acc[t]);
    }
    double* out = Xout + r0 * L;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(int64_t)(q + 4 * r) * L + 16 * t + x] = acc[t][r];
}

// part[chunk][j] = sum over the chunk's rows of (Z[r][j] - sigma[j] Qr[r][j])^2
__global__ __launch_bounds__(256) void svd_resid_kernel(const double* __restrict__ Z, const double* __restrict__ Qr,
                                                        const double* __restrict__ sigma, int L, int64_t rows_per_chunk,
                                                        double* __restrict__ part)
{
    const int j = threadIdx.x;
    if (j >= L) return;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
    const double sg = sigma[j];
    double s = 0.0;
    for (int64_t r = r0; r < r0 + rows_per_chunk; ++r) {
        const double d = Z[r * L + j] - sg * Qr[r * L + j];
        s += d * d;
    }
    part[(int64_t)blockIdx.x * L + j] = s;
}

// Tpart[chunk][i][j] = sum over the chunk's rows of A[r][i] B[r][j]   (svd_gram_kernel with two operands)
template <int NT>
__global__ __launch_bounds__(256) void svd_cross_gram_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                             int64_t rows_per_chunk, double* __restrict__ Tpart)
{
    constexpr int L = 16 * NT, NW = (NT + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
    const int ti = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
    f64x4 acc[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) acc[c] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int64_t r = r0; r < r0 + rows_per_chunk; r += 4) {
        const double a = A[(r + q) * L + 16 * ti + x];
        const double* brow = B + (r + q) * L;
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            const int tj = wave + 4 * c;
            if (tj < NT) acc[c] = MFMA_F64(a, brow[16 * tj + x], acc[c]);
        }
    }
    double* out = Tpart + (int64_t)blockIdx.x * L * L;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int tj = wave + 4 * c;
        if (tj < NT)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(int64_t)(16 * ti + q + 4 * r) * L + 16 * tj + x] = acc[c][r];
    }
}

// out[r][j] = j < nlock ? a[r][j] : b[r][j]
__global__ __launch_bounds__(256) void svd_select_cols_kernel(const double* __restrict__ a, const double* __restrict__ b, int nlock,
                                                              int L, int64_t count, double* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    out[e] = (int)(e % L) < nlock ? a[e] : b[e];
}

// out = s1 * (ax . colscale) + s2 * x1 + s3 * x0     (column j of ax scaled by colscale[j] when given)
__global__ __launch_bounds__(256) void svd_combine_kernel(const double* __restrict__ ax, const double* __restrict__ colscale,
                                                          const double* __restrict__ x1, const double* __restrict__ x0,
                                                          double s1, double s2, double s3, int L, int64_t count,
                                                          double* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    double v = s1 * ax[e] * (colscale ? colscale[e % L] : 1.0) + s2 * x1[e];
    if (x0) v += s3 * x0[e];
    out[e] = v;
}

// ---- host: symmetric eigenproblem of an L x L matrix (cyclic threshold Jacobi) ----
// a: in = the matrix (row-major, both triangles), out = destroyed; evec columns = eigenvectors,
// eval descending.
void jacobi_eigh(int L, std::vector<double>& a, std::vector<double>& evec, std::vector<double>& eval)
{
    evec.assign((size_t)L * L, 0.0);
    for (int i = 0; i < L; ++i) evec[(size_t)i * L + i] = 1.0;
    double scale = 0.0;
    for (int i = 0; i < L; ++i) scale = std::max(scale, std::fabs(a[(size_t)i * L + i]));
    if (scale == 0.0) scale = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < L; ++p)
            for (int qq = p + 1; qq < L; ++qq) off = std::max(off, std::fabs(a[(size_t)p * L + qq]));
        if (off <= 1e-300 || off <= 4e-17 * scale) break;
        for (int p = 0; p < L - 1; ++p)
            for (int qq = p + 1; qq < L; ++qq) {
                const double apq = a[(size_t)p * L + qq];
                if (std::fabs(apq) <= 1e-18 * scale) continue;
                const double app = a[(size_t)p * L + p], aqq = a[(size_t)qq * L + qq];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int i = 0; i < L; ++i) {           // columns p, q
                    const double aip = a[(size_t)i * L + p], aiq = a[(size_t)i * L + qq];
                    a[(size_t)i * L + p] = c * aip - s * aiq;
                    a[(size_t)i * L + qq] = s * aip + c * aiq;
                }
                for (int i = 0; i < L; ++i) {           // rows p, q
                    const double api = a[(size_t)p * L + i], aqi = a[(size_t)qq * L + i];
                    a[(size_t)p * L + i] = c * api - s * aqi;
                    a[(size_t)qq * L + i] = s * api + c * aqi;
                }
                for (int i = 0; i < L; ++i) {
                    const double vip = evec[(size_t)i * L + p], viq = evec[(size_t)i * L + qq];
                    evec[(size_t)i * L + p] = c * vip - s * viq;
                    evec[(size_t)i * L + qq] = s * vip + c * viq;
                }
            }
    }
    std::vector<int> order(L);
    for (int i = 0; i < L; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int i, int j) { return a[(size_t)i * L + i] > a[(size_t)j * L + j]; });
    std::vector<double> v2((size_t)L * L);
    eval.resize(L);
    for (int c = 0; c < L; ++c) {
        eval[c] = a[(size_t)order[c] * L + order[c]];
        for (int i = 0; i < L; ++i) v2[(size_t)i * L + c] = evec[(size_t)i * L + order[c]];
    }
    evec.swap(v2);
}

struct SvdWork {
    nmfx_engine* E; int L, NT;
    double *Q = nullptr, *Qr = nullptr, *Z = nullptr, *Y = nullptr, *U = nullptr, *part = nullptr, *T = nullptr,
           *M = nullptr, *sig = nullptr, *C1 = nullptr, *C2 = nullptr, *C3 = nullptr;
    int64_t part_count = 0;
    ~SvdWork() { for (double* p : {Q, Qr, Z, Y, U, part, T, M, sig, C1, C2, C3}) if (p) hipFree(p); }
};

template <int NT>
int launch_apply(SvdWork& w, bool trans, const double* B, double* C)
{
    nmfx_engine* E = w.E;
    constexpr int L = 16 * NT, LP = (L % 32 == 0) ? L + 16 : L;
    const int64_t out_rows = trans ? E->np : E->mp, clen = trans ? E->mp : E->np;
    const int64_t blocks = out_rows / 64;
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>((E->ncu + blocks - 1) / blocks, clen / (4 * KC)));
    while ((int64_t)splits * out_rows * L > w.part_count && splits > 1) --splits;
    const size_t shm = (size_t)KC * LDA * 4 + (size_t)KC * LP * 8;
    dim3 grid((unsigned)blocks, (unsigned)splits);
    { int rc_ = nmfx_need_v(E); if (rc_) return rc_; }
    if (trans) hipLaunchKernelGGL((svd_apply_kernel<NT, true>), grid, dim3(256), shm, E->stream, E->V, E->np, clen, B, w.part, out_rows);
    else hipLaunchKernelGGL((svd_apply_kernel<NT, false>), grid, dim3(256), shm, E->stream, E->V, E->np, clen, B, w.part, out_rows);
    NMFX_HIP(hipGetLastError());
    const int64_t count = out_rows * L;
    hipLaunchKernelGGL(svd_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, E->stream, w.part, splits, count, C);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int NT>
int gram_to_host(SvdWork& w, const double* X, int64_t rows, std::vector<double>& T)
{
    nmfx_engine* E = w.E;
    constexpr int L = 16 * NT;
    int64_t rpc = 512;
    while (rows % rpc) rpc >>= 1;                      // rows is a multiple of 128
    const int64_t chunks = rows / rpc;
    hipLaunchKernelGGL((svd_gram_kernel<NT>), dim3((unsigned)chunks, NT), dim3(256), 0, E->stream, X, rpc, w.part);
    NMFX_HIP(hipGetLastError());
    hipLaunchKernelGGL(svd_sum_kernel, dim3((unsigned)((L * L + 255) / 256)), dim3(256), 0, E->stream, w.part, (int)chunks,
                       (int64_t)L * L, w.T);
    NMFX_HIP(hipGetLastError());
    T.resize((size_t)L * L);
    NMFX_HIP(hipMemcpyAsync(T.data(), w.T, T.size() * 8, hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    for (int i = 0; i < L; ++i)                        // exact symmetry for the host solver
        for (int j = i + 1; j < L; ++j) T[(size_t)j * L + i] = T[(size_t)i * L + j] = 0.5 * (T[(size_t)i * L + j] + T[(size_t)j * L + i]);
    return NMFX_OK;
}

template <int NT>
int cross_gram_to_host(SvdWork& w, const double* A, const double* B, int64_t rows, std::vector<double>& T)
{
    nmfx_engine* E = w.E;
    constexpr int L = 16 * NT;
    int64_t rpc = 512;
    while (rows % rpc) rpc >>= 1;
    const int64_t chunks = rows / rpc;
    hipLaunchKernelGGL((svd_cross_gram_kernel<NT>), dim3((unsigned)chunks, NT), dim3(256), 0, E->stream, A, B, rpc, w.part);
    NMFX_HIP(hipGetLastError());
    hipLaunchKernelGGL(svd_sum_kernel, dim3((unsigned)((L * L + 255) / 256)), dim3(256), 0, E->stream, w.part, (int)chunks,
                       (int64_t)L * L, w.T);
    NMFX_HIP(hipGetLastError());
    T.resize((size_t)L * L);
    NMFX_HIP(hipMemcpyAsync(T.data(), w.T, T.size() * 8, hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

template <int NT>
int rotate(SvdWork& w, const double* Xin, const std::vector<double>& M, double* Xout, int64_t rows)
{
    nmfx_engine* E = w.E;
    NMFX_HIP(hipMemcpyAsync(w.M, M.data(), M.size() * 8, hipMemcpyHostToDevice, E->stream));
    hipLaunchKernelGGL((svd_rotate_kernel<NT>), dim3((unsigned)(rows / 64)), dim3(256), 0, E->stream, Xin, w.M, Xout);
    NMFX_HIP(hipGetLastError());
    NMFX_HIP(hipStreamSynchronize(E->stream));         // M (host vector) may go out of scope
    return NMFX_OK;
}

// X <- orthonormal basis of span(X) in the Ritz order of X^T X; returns eigenvalues (descending)
// and the rotation S (so that the caller can rotate a companion matrix).  Two passes: the
// second one repairs the orthogonality the first loses to cond(X^T X).
template <int NT>
int orthonormalize(SvdWork& w, double* X, double* tmp, int64_t rows, std::vector<double>& theta, std::vector<double>& S)
{
    constexpr int L = 16 * NT;
    std::vector<double> T, ev, M((size_t)L * L);
    int rc;
    if ((rc = gram_to_host<NT>(w, X, rows, T))) return rc;
    jacobi_eigh(L, T, S, theta);
    const double floor_ = std::max(theta[0], 0.0) * 1e-28;
    for (int j = 0; j < L; ++j) {
        const double sc = theta[j] > floor_ && theta[j] > 0.0 ? 1.0 / std::sqrt(theta[j]) : 0.0;
        for (int i = 0; i < L; ++i) M[(size_t)i * L + j] = S[(size_t)i * L + j] * sc;
    }
    if ((rc = rotate<NT>(w, X, M, tmp, rows))) return rc;
    // polish: tmp^T tmp = I + E  ->  X = tmp (I + E)^-1/2 via its own eigen-decomposition
    std::vector<double> S2, th2;
    if ((rc = gram_to_host<NT>(w, tmp, rows, T))) return rc;
    jacobi_eigh(L, T, S2, th2);
    std::vector<double> P((size_t)L * L, 0.0);
    for (int c = 0; c < L; ++c) {
        const double sc = th2[c] > 0.25 ? 1.0 / std::sqrt(th2[c]) : 0.0;        // (zeroed columns stay zero)
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < L; ++j) P[(size_t)i * L + j] += S2[(size_t)i * L + c] * sc * S2[(size_t)j * L + c];
    }
    return rotate<NT>(w, tmp, P, X, rows);
}

template <int NT>
int topk_svd(nmfx_engine* E, int k, double tol, int max_sweeps, uint64_t seed, double* u_out, double* s_out,
             double* vt_out, int* sweeps_out, double* resid_out)
{
    constexpr int L = 16 * NT;
    SvdWork w; w.E = E; w.L = L; w.NT = NT;
    const int64_t mp = E->mp, np = E->np;
    auto alloc = [&](double** p, int64_t count) -> int {
        NMFX_HIP(hipMalloc(reinterpret_cast<void**>(p), (size_t)count * 8));
        NMFX_HIP(hipMemsetAsync(*p, 0, (size_t)count * 8, E->stream));
        return NMFX_OK;
    };
    int rc;
    w.part_count = std::max<int64_t>({4 * std::max(mp, np) * L, (std::max(mp, np) / 128) * (int64_t)L * L, (int64_t)4096 * L});
    if ((rc = alloc(&w.Q, np * L)) || (rc = alloc(&w.Qr, np * L)) || (rc = alloc(&w.Z, np * L)) ||
        (rc = alloc(&w.Y, mp * L)) || (rc = alloc(&w.U, mp * L)) || (rc = alloc(&w.part, w.part_count)) ||
        (rc = alloc(&w.T, (int64_t)L * L)) || (rc = alloc(&w.M, (int64_t)L * L)) || (rc = alloc(&w.sig, L)) ||
        (rc = alloc(&w.C1, np * L)) || (rc = alloc(&w.C2, np * L)) || (rc = alloc(&w.C3, np * L))) return rc;
    {   // random start, zero in the padded rows
        std::vector<double> q0((size_t)np * L, 0.0);
        std::mt19937_64 gen(seed);
        std::normal_distribution<double> nd(0.0, 1.0);
        for (int64_t r = 0; r < E->n; ++r)
            for (int j = 0; j < L; ++j) q0[(size_t)r * L + j] = nd(gen);
        NMFX_HIP(hipMemcpyAsync(w.Q, q0.data(), q0.size() * 8, hipMemcpyHostToDevice, E->stream));
        NMFX_HIP(hipStreamSynchronize(E->stream));
    }
    std::vector<double> theta, S, sigma(L), M((size_t)L * L), rpart, res(L, 0.0);
    if ((rc = orthonormalize<NT>(w, w.Q, w.Z, np, theta, S))) return rc;
    int sweep = 0;
    double worst = INFINITY;
    for (;;) {
        ++sweep;
        // Y = V Q; Rayleigh-Ritz on Y^T Y
        if ((rc = launch_apply<NT>(w, false, w.Q, w.Y))) return rc;
        {   // U = orthonormal Ritz basis of span(Y); Qr = Q S
            std::vector<double> T;
            if ((rc = gram_to_host<NT>(w, w.Y, mp, T))) return rc;
            jacobi_eigh(L, T, S, theta);
            const double floor_ = std::max(theta[0], 0.0) * 1e-28;
            for (int j = 0; j < L; ++j) {
                sigma[j] = theta[j] > 0.0 ? std::sqrt(theta[j]) : 0.0;
                const double sc = theta[j] > floor_ && theta[j] > 0.0 ? 1.0 / sigma[j] : 0.0;
                for (int i = 0; i < L; ++i) M[(size_t)i * L + j] = S[(size_t)i * L + j] * sc;
            }
            if ((rc = rotate<NT>(w, w.Y, M, w.U, mp))) return rc;
            if ((rc = rotate<NT>(w, w.Q, S, w.Qr, np))) return rc;
        }
        // Z = V^T U; residuals of the Ritz triplets
        if ((rc = launch_apply<NT>(w, true, w.U, w.Z))) return rc;
        {
            NMFX_HIP(hipMemcpyAsync(w.sig, sigma.data(), (size_t)L * 8, hipMemcpyHostToDevice, E->stream));
            int64_t rpc = 512; while (np % rpc) rpc >>= 1;
            const int64_t chunks = np / rpc;
            hipLaunchKernelGGL(svd_resid_kernel, dim3((unsigned)chunks), dim3(256), 0, E->stream, w.Z, w.Qr, w.sig, L, rpc, w.part);
            NMFX_HIP(hipGetLastError());
            rpart.resize((size_t)chunks * L);
            NMFX_HIP(hipMemcpyAsync(rpart.data(), w.part, rpart.size() * 8, hipMemcpyDeviceToHost, E->stream));
            NMFX_HIP(hipStreamSynchronize(E->stream));
            worst = 0.0;
            for (int j = 0; j < L; ++j) {
                double s2 = 0.0;
                for (int64_t c = 0; c < chunks; ++c) s2 += rpart[(size_t)c * L + j];
                res[j] = std::sqrt(s2);
                if (j < k && sigma[j] > 0.0) worst = std::max(worst, res[j] / sigma[0]);
            }
        }
        if (worst <= tol || sweep >= max_sweeps) break;
        // next block.  Plain subspace iteration: Q = orth(Z) (= orth(A Qr), A = V^T V).  Its error
        // shrinks by (s_{L+1} / s_k)^2 per sweep, which is hopeless without a gap behind the k-th
        // singular value (noise bulk, real data): from the fourth sweep on the block is therefore
        // passed through a Chebyshev polynomial of A that is small on [0, theta_L] (the block's
        // smallest Ritz value bounds the unwanted spectrum) and grows fast above it (Zhou & Saad's
        // filtered subspace iteration; scaled three-term recurrence, degree 10: the error then
        // shrinks like exp(-2 d sqrt(gap)) instead of exp(-d gap) per d applications of A).
        // The leading Ritz pairs that have converged are LOCKED: non-negative data always has one
        // dominant singular value (here 30 x the next), and a polynomial that is large at theta_1 would
        // wipe every other direction out of the block in double precision.  The filter therefore acts
        // on the deflated operator A' = A - sum_locked theta_i q_i q_i^T and on the unlocked columns,
        // and its degree is capped by the dynamic range it creates between theta_nl and theta_k.
        int nl = 0;
        while (nl < k && sigma[nl] > 0.0 && res[nl] / sigma[0] <= tol) ++nl;
        const double cb = theta[L - 1], a0 = theta[std::min(nl, L - 1)];
        int degree = 10;
        if (cb > 1e-12 * a0 && a0 > 1.000001 * cb) {
            const double e0 = 0.5 * cb;
            auto grow = [&](double th) { const double xx = std::max((th - e0) / e0, 1.0); return xx + std::sqrt(xx * xx - 1.0); };
            const double ratio = grow(a0) / std::max(grow(theta[k - 1]), 1.0);
            if (ratio > 1.0) degree = (int)std::min(10.0, std::floor(std::log(1e9) / std::log(ratio)));
        }
        if (sweep < 3 || degree < 2 || !(cb > 1e-12 * a0) || !(a0 > 1.000001 * cb)) {
            std::vector<double> th2, S2;
            if ((rc = orthonormalize<NT>(w, w.Z, w.Q, np, th2, S2))) return rc;
            std::swap(w.Q, w.Z);                       // orthonormalize leaves its result in its first argument
        } else {
            const double e = 0.5 * cb, c = 0.5 * cb;
            double sg = e / (a0 - c);
            const double tau = 2.0 / sg;
            const int64_t count = np * L;
            const unsigned cgrid = (unsigned)((count + 255) / 256);
            std::vector<double> Cm, D((size_t)L * L);
            // A' X = A X - Qr D,  D[i][:] = theta_i (Qr^T X)[i][:] for the locked i
            auto deflate = [&](const double* xin, double* axbuf) -> int {
                if (nl == 0) return NMFX_OK;
                int r2;
                if ((r2 = cross_gram_to_host<NT>(w, w.Qr, xin, np, Cm))) return r2;
                std::fill(D.begin(), D.end(), 0.0);
                for (int i = 0; i < nl; ++i)
                    for (int j = 0; j < L; ++j) D[(size_t)i * L + j] = theta[i] * Cm[(size_t)i * L + j];
                if ((r2 = rotate<NT>(w, w.Qr, D, w.C3, np))) return r2;
                hipLaunchKernelGGL(svd_combine_kernel, dim3(cgrid), dim3(256), 0, E->stream, w.C3, (const double*)nullptr, axbuf,
                                   (const double*)nullptr, -1.0, 1.0, 0.0, L, count, axbuf);
                NMFX_HIP(hipGetLastError());
                return NMFX_OK;
            };
            // X0 = Qr;  A X0 = Z diag(sigma)  (made explicit in Z, then deflated)
            double *x0 = w.Qr, *x1 = w.C1, *x2 = w.C2, *ax = w.Z;
            hipLaunchKernelGGL(svd_combine_kernel, dim3(cgrid), dim3(256), 0, E->stream, ax, w.sig, ax, (const double*)nullptr,
                               1.0, 0.0, 0.0, L, count, ax);
            NMFX_HIP(hipGetLastError());
            if ((rc = deflate(x0, ax))) return rc;
            hipLaunchKernelGGL(svd_combine_kernel, dim3(cgrid), dim3(256), 0, E->stream, ax, (const double*)nullptr, x0,
                               (const double*)nullptr, sg / e, -c * sg / e, 0.0, L, count, x1);
            NMFX_HIP(hipGetLastError());
            double* xprev = x0;                        // x0 must stay intact (it is Qr, the deflation basis)
            for (int i = 2; i <= degree; ++i) {
                const double sn = 1.0 / (tau - sg);
                if ((rc = launch_apply<NT>(w, false, x1, w.Y))) return rc;           // V X1
                if ((rc = launch_apply<NT>(w, true, w.Y, ax))) return rc;            // A X1
                if ((rc = deflate(x1, ax))) return rc;
                double* dst = (xprev == w.Qr) ? x2 : xprev;                          // never overwrite Qr
                hipLaunchKernelGGL(svd_combine_kernel, dim3(cgrid), dim3(256), 0, E->stream, ax, (const double*)nullptr, x1, xprev,
                                   2.0 * sn / e, -2.0 * sn * c / e, -sg * sn, L, count, dst);
                NMFX_HIP(hipGetLastError());
                xprev = x1; x1 = dst;
                sg = sn;
            }
            // next block = [locked Ritz vectors | filtered columns], orthonormalised
            hipLaunchKernelGGL(svd_select_cols_kernel, dim3(cgrid), dim3(256), 0, E->stream, w.Qr, x1, nl, L, count, ax);
            NMFX_HIP(hipGetLastError());
            std::vector<double> th2, S2;
            if ((rc = orthonormalize<NT>(w, ax, w.C3, np, th2, S2))) return rc;
            std::swap(w.Q, w.Z);                       // ax == w.Z holds the new block
        }
    }
    // polish U (orthonormal to rounding) and hand out the k leading triplets
    {
        std::vector<double> hu((size_t)mp * L), hq((size_t)np * L);
        NMFX_HIP(hipMemcpyAsync(hu.data(), w.U, hu.size() * 8, hipMemcpyDeviceToHost, E->stream));
        NMFX_HIP(hipMemcpyAsync(hq.data(), w.Qr, hq.size() * 8, hipMemcpyDeviceToHost, E->stream));
        NMFX_HIP(hipStreamSynchronize(E->stream));
        for (int64_t r = 0; r < E->m; ++r)
            for (int j = 0; j < k; ++j) u_out[r * k + j] = hu[(size_t)r * L + j];
        for (int j = 0; j < k; ++j) {
            s_out[j] = sigma[j];
            for (int64_t c = 0; c < E->n; ++c) vt_out[(int64_t)j * E->n + c] = hq[(size_t)c * L + j];
        }
    }
    if (sweeps_out) *sweeps_out = sweep;
    if (resid_out) *resid_out = worst;
    return NMFX_OK;
}

}  // namespace

extern "C" int nmfx_topk_svd(nmfx_handle_t E, int k, int block, double tol, int max_sweeps, uint64_t seed,
                             double* u, double* s, double* vt, int* sweeps, double* resid)
{
    if (!E || !u || !s || !vt || k < 1) return NMFX_E_ARG;
    if (!E->have_v) { E->err = "upload V first"; return NMFX_E_STATE; }
    const int64_t lim = std::min(E->m, E->n);
    if (k > lim) { E->err = "topk_svd: k exceeds min(m, n)"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    if (block <= 0) block = std::max(k + 16, k + k / 2);
    block = std::max(block, k);
    static const int sizes[] = {32, 48, 64, 96, 128, 160, 192};
    int L = 0;
    for (int c : sizes) if (c >= block) { L = c; break; }
    if (!L) { if (k <= 192) L = 192; else { E->err = "topk_svd: k too large"; return NMFX_E_ARG; } }
    if (tol <= 0) tol = 1e-11;
    if (max_sweeps <= 0) max_sweeps = 4000;
    switch (L) {
        case 32: return topk_svd<2>(E, k, tol, max_sweeps, seed, u, s, vt, sweeps, resid);
        case 48: return topk_svd<3>(E, k, tol, max_sweeps, seed, u, s, vt, sweeps, resid);
        case 64: return topk_svd<4>(E, k, tol, max_sweeps, seed, u, s, vt, sweeps, resid);
        case 96: return topk_svd<6>(E, k, tol, max_sweeps, seed, u, s, vt, sweeps, resid);
        case 128: return topk_svd<8>(E, k, tol, max_sweeps, seed, u, s, vt, sweeps, resid);
        case 160: return topk_svd<10>(E, k, tol, max_sweeps, seed, u, s, vt, sweeps, resid);
        default: return topk_svd<12>(E, k, tol, max_sweeps, seed, u, s, vt, sweeps, resid);
    }
}

// (nmfx_create: forces this translation unit's code object onto the device under the library's start-up lock)
int nmfx_preload_svd() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(svd_sum_kernel)) == hipSuccess ? 0 : -1; }
