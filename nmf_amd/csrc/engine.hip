// C ABI of libnmfx.so: handle lifecycle, data movement, iteration state.
#include "nmfx_internal.h"
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <tuple>
#include <mutex>
#include <utility>

int nmfx_mur_eu_phase_a(nmfx_engine* E, double lambda_w, int64_t j);
int nmfx_mur_eu_phase_b(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_mur_eu_finish_a(nmfx_engine* E, int64_t j);
int nmfx_mur_kl_phase_a(nmfx_engine* E, double lambda_w, int64_t j);
int nmfx_mur_kl_phase_b(nmfx_engine* E, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j);
int nmfx_mur_kl_finish_a(nmfx_engine* E, int64_t j);

#ifndef NMFX_DEFAULT_PRECISION
#define NMFX_DEFAULT_PRECISION 1   /* split bf16 where available (k padded to 64, MUR-eu); NMFX_PRECISION=f32 for exact f32 */
#endif
static thread_local std::string g_err;

int nmfx_allow_lds(nmfx_engine* E, const void* kernel, int bytes) {
    if (bytes <= 64 * 1024) return NMFX_OK;            // within the default limit
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> granted;
    std::lock_guard<std::mutex> lock(mu);
    int& have = granted[std::make_pair(E->device, kernel)];
    if (have >= bytes) return NMFX_OK;
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
    return NMFX_OK;
}

// The HIP runtime loads a translation unit's code object onto a device lazily, at the first launch of one of its
// kernels.  So that two host threads never make those first launches at the same time, every handle passes through here
// before its own first launch: once per device, under a lock, one kernel of every translation unit is touched
// (hipFuncGetAttributes) and a launch is made and waited for.  (Precaution.  The failure that two concurrent creates did
// show -- "unknown error" on the second thread's first launch -- came from the unguarded attribute / occupancy queries of
// nmfx_phase_occupancy, which are serialised now.)
__global__ void nmfx_startup_kernel(int* p) { if (p) *p = 0; }
static int preload_once(nmfx_engine* E) {
    static std::mutex mu;
    static std::map<int, bool> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done[E->device]) return NMFX_OK;
    if (nmfx_preload_bf16() || nmfx_preload_products() || nmfx_preload_mur() || nmfx_preload_kl() || nmfx_preload_aoadmm() ||
        nmfx_preload_anls() || nmfx_preload_svd() || nmfx_preload_prox() || nmfx_preload_generic()) {
        E->err = "loading the kernels onto the device failed"; return NMFX_E_HIP; }
    hipLaunchKernelGGL(nmfx_startup_kernel, dim3(1), dim3(1), 0, 0, (int*)nullptr);
    NMFX_HIP(hipGetLastError());
    NMFX_HIP(hipDeviceSynchronize());
    done[E->device] = true;
    return NMFX_OK;
}

static int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

__global__ void cvt_f64_rows_kernel(const double* __restrict__ src, int64_t lds, int64_t n, float* __restrict__ dst,
                                    int64_t ldd, int64_t rows)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (c < n && r < rows) dst[r * ldd + c] = (float)src[r * lds + c];
}

__global__ void init_state_kernel(DevState* st) {
    st->flag = 0; st->stop_i = -1; st->n_obj = 0; st->obj_prev = 0.0;
    st->inner_stop = 0; st->inner_count = 0; st->notpd = 0; st->notpd_pending = 0; st->rho = 0.0; st->j_base = 0;
    st->nnls_evicted = 0; st->nnls_capped = 0; st->nnls_fallback = 0; st->nnls_noinv = 0;
    for (int i = 0; i < 4; ++i) { st->ao_hint[i >> 1][i & 1] = 0; st->ao_paths[i] = 0; }
    st->ao_continued = 0;
    for (int p = 0; p < 2; ++p) { st->pflag[p] = 0; st->pstop_i[p] = -1; st->pn_obj[p] = 0; }
    st->stop_guard = 0.0;
}

__global__ void set_guard_kernel(DevState* st, double g) { st->stop_guard = g; }
__global__ void resume_kernel(DevState* st) { st->flag = 0; st->stop_i = -1; }

__global__ void shift_iteration_base_kernel(DevState* st, long long delta) { st->j_base += delta; }

template <typename T>
static int dev_alloc(nmfx_engine* E, T** p, int64_t count) {
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T)));
    NMFX_HIP(hipMemsetAsync(*p, 0, (size_t)count * sizeof(T), E->stream));
    return NMFX_OK;
}

int nmfx_ensure_obj_capacity(nmfx_engine* E, int64_t need) {
    if (need <= E->obj_cap) return NMFX_OK;
    int64_t cap = E->obj_cap ? E->obj_cap : 4096;
    while (cap < need) cap *= 2;
    double* nbuf = nullptr;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&nbuf), (size_t)cap * sizeof(double)));
    NMFX_HIP(hipMemsetAsync(nbuf, 0, (size_t)cap * sizeof(double), E->stream));
    if (E->obj_hist) {
        NMFX_HIP(hipMemcpyAsync(nbuf, E->obj_hist, (size_t)E->obj_cap * sizeof(double),
                                hipMemcpyDeviceToDevice, E->stream));
        NMFX_HIP(hipStreamSynchronize(E->stream));
        NMFX_HIP(hipFree(E->obj_hist));
    }
    E->obj_hist = nbuf;
    E->obj_cap = cap;
    return NMFX_OK;
}

int nmfx_ensure_inner_capacity(nmfx_engine* E, int64_t need) {
    if (need <= E->inner_cap) return NMFX_OK;
    int64_t cap = E->inner_cap ? E->inner_cap : 4096;
    while (cap < need) cap *= 2;
    int32_t* nbuf = nullptr;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&nbuf), (size_t)cap * 2 * sizeof(int32_t)));
    NMFX_HIP(hipMemsetAsync(nbuf, 0, (size_t)cap * 2 * sizeof(int32_t), E->stream));
    if (E->inner_hist) {
        NMFX_HIP(hipMemcpyAsync(nbuf, E->inner_hist, (size_t)E->inner_cap * 2 * sizeof(int32_t),
                                hipMemcpyDeviceToDevice, E->stream));
        NMFX_HIP(hipStreamSynchronize(E->stream));
        NMFX_HIP(hipFree(E->inner_hist));
    }
    E->inner_hist = nbuf;
    E->inner_cap = cap;
    return NMFX_OK;
}

extern "C" {

int nmfx_version(void) { return 300; }      // round 3: nmfx_set_exchange_buffers takes sizes; nmfx_comm_*; k <= 256

int nmfx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* nmfx_last_error(nmfx_handle_t h) { return h ? h->err.c_str() : g_err.c_str(); }

int nmfx_create(nmfx_handle_t* out, int device, int64_t m, int64_t n, int k) {
    if (!out) { g_err = "out is NULL"; return NMFX_E_ARG; }
    *out = nullptr;
    if (m <= 0 || n <= 0 || k <= 0) { g_err = "m, n, k must be positive"; return NMFX_E_ARG; }
    if (k > 4096) { g_err = "k > 4096 is not supported"; return NMFX_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device"; return NMFX_E_HIP; }
    if (device < 0 || device >= ndev) { g_err = "device index out of range"; return NMFX_E_ARG; }
    nmfx_engine* E = new nmfx_engine();
    E->device = device; E->m = m; E->n = n; E->k = k;
    // (k > 128: a multiple of 128 -- the MUR solvers compose their iteration from the generic product kernel, kernels_generic.hip)
    E->kp = k <= 16 ? 16 : k <= 32 ? 32 : k <= 64 ? 64 : k <= 128 ? 128 : (int)round_up(k, 128);
    // 128: the split-bf16 kernel works on 128-row blocks of V and of V^T; the f32 kernels need 64
    E->mp = round_up(m, 2 * NMFX_TILE); E->np = round_up(n, 2 * NMFX_TILE);
    auto fail = [&](int rc) { g_err = E->err; nmfx_destroy(E); return rc; };
#define TRY(x) do { int rc_ = (x); if (rc_) return fail(rc_); } while (0)
#define TRYHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { E->err = std::string(#x) + ": " + hipGetErrorString(e_); return fail(NMFX_E_HIP); } } while (0)
    TRYHIP(hipSetDevice(device));
    TRY(preload_once(E));
    TRYHIP(hipStreamCreateWithFlags(&E->own_stream, hipStreamNonBlocking));
    E->stream = E->own_stream;
    // split configuration: enough workgroups to fill 256 CUs twice over
    const int64_t rb = E->mp / 64, cb = E->np / 64;
    // grids = a whole number of resident rounds: 256 CUs x blocks/CU of each kernel
    int wocc = 2, hocc = 2, ncu = 256;
    nmfx_phase_occupancy(E->kp, &wocc, &hocc);
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    E->ncu = ncu;
    { const char* pm = getenv("NMFX_PRECISION");
      E->precision = (pm && (!strcmp(pm, "f32") || !strcmp(pm, "fp32"))) ? 0 : (pm && !strcmp(pm, "bf16")) ? 1 : NMFX_DEFAULT_PRECISION; }
    const char* ev;
    const int64_t wtarget = (ev = getenv("NMFX_WBLOCKS")) ? atoll(ev) : (int64_t)ncu * wocc;
    const int64_t htarget = (ev = getenv("NMFX_HBLOCKS")) ? atoll(ev) : (int64_t)ncu * hocc;
    int64_t ws = std::max<int64_t>(1, (wtarget + rb / 2) / rb); ws = std::min<int64_t>(ws, std::max<int64_t>(1, cb / 4));
    int64_t hs = std::max<int64_t>(1, (htarget + cb / 2) / cb); hs = std::min<int64_t>(hs, rb);
    int64_t gs = std::min<int64_t>(32, std::max<int64_t>(1, std::min(E->mp, E->np) / 256));   // Gram kernels: gs x kp/16 blocks
    E->wsplit = (int)ws; E->hsplit = (int)hs; E->gsplit = (int)gs;
    const int64_t kp = E->kp, mp = E->mp, np = E->np;
    TRY(dev_alloc(E, &E->V, mp * np));
    TRY(dev_alloc(E, &E->W[0], mp * kp));
    TRY(dev_alloc(E, &E->W[1], mp * kp));
    TRY(dev_alloc(E, &E->H, kp * np));
    TRY(dev_alloc(E, &E->HHt, kp * kp));
    // (the split-bf16 path spreads its Gram by-products over up to 8 row blocks x splits slabs)
    TRY(dev_alloc(E, &E->HHt_part, std::max<int64_t>(std::max(gs, ws), 40) * kp * kp));
    TRY(dev_alloc(E, &E->G_part, std::max<int64_t>(std::max(gs, hs), 4 * std::max<int64_t>(hs, (int64_t)ncu) + 8) * kp * kp));
    TRY(dev_alloc(E, &E->A_part, ws * mp * kp));
    TRY(dev_alloc(E, &E->B_part, hs * kp * np));
    E->obj_part_cap = 2 * (std::max<int64_t>(rb * ws, cb * hs) + 64) + 2 * (int64_t)ncu;      // (+ 2 ncu: the segments of a stream-K plan, at most workers + row blocks)
    TRY(dev_alloc(E, &E->obj_part, E->obj_part_cap));      // (x 2: pair mode keeps two partials per block)
    TRY(dev_alloc(E, &E->xf32, kp * np + kp * kp + kp + NMFX_XTAIL));
    TRY(dev_alloc(E, &E->xf64, 8 + 4 * NMFX_MAX_FUSED_ROUNDS));
    TRY(dev_alloc(E, &E->state, 1));
    hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, E->stream, E->state);
    TRYHIP(hipGetLastError());
    TRY(nmfx_ensure_obj_capacity(E, 4096));
    TRYHIP(hipStreamSynchronize(E->stream));
    {   // Memory plan of the split-bf16 path: tile-major V and V^T next to the row-major V while they are built (three
        // V-sized buffers), two afterwards when the row-major one is dropped.  It is dropped for V >= 4 GiB (or
        // NMFX_DROP_V=1 / 0) and rebuilt on demand (nmfx_need_v).  If even the three-copy peak does not fit, the engine
        // runs the exact-f32 kernels (one copy) and SAYS so: nmfx_get_note().
        const double vbytes = (double)mp * (double)np * 4.0;
        const char* dv = getenv("NMFX_DROP_V");
        E->drop_v = dv ? atoi(dv) != 0 : vbytes >= 4.0 * 1024 * 1024 * 1024;
        size_t free_b = 0, total_b = 0;
        if (E->precision == 1 && nmfx_bf16_supported(E) && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double need = 2.0 * vbytes + 64.0 * (double)(mp + np) * kp + (256u << 20);
            if ((double)free_b < need) {
                E->precision = 0;
                char buf[256];
                snprintf(buf, sizeof buf, "split-bf16 products need %.1f GiB more device memory than the %.1f GiB that are free: "
                         "running the exact-f32 kernels (about half the speed)", need / 1073741824.0, (double)free_b / 1073741824.0);
                E->note = buf;
            }
        }
    }
#undef TRY
#undef TRYHIP
    *out = E;
    return NMFX_OK;
}

int nmfx_destroy(nmfx_handle_t E) {
    if (!E) return NMFX_OK;
    hipSetDevice(E->device);
    if (E->stream) hipStreamSynchronize(E->stream);
    nmfx_comm_free(E);
    for (auto& t : E->prof_pending) { hipEventDestroy(std::get<1>(t)); hipEventDestroy(std::get<2>(t)); }
    void* bufs[] = {E->V, E->W[0], E->W[1], E->H, E->HHt, E->HHt_part, E->G_part, E->A_part, E->B_part,
                    E->obj_part, E->own_x ? (void*)E->xf32 : nullptr, E->own_x ? (void*)E->xf64 : nullptr,
                    E->obj_hist, E->state, E->dualW, E->dualH, E->auxW, E->auxH, E->Minv, E->nrm_part,
                    E->inner_hist, E->Pw, E->Ph, E->Asum, E->S, E->DV, E->Vt, E->Vtile, E->Bt_part, E->Whi[0], E->Whi[1],
                    E->Wlo[0], E->Wlo[1], E->WThi, E->WTlo, E->Hhi, E->Hlo, E->HThi, E->HTlo, E->nrm_rounds, E->bkX, E->bkU,
                    E->nnls_ginv, E->nnls_todo, E->G_big, E->kl_part, E->Bt_chunk, E->gx_part, E->gx_d, E->gx_s, E->gx_r, E->gx_w64, E->gx_nrm, E->prox_keys, E->gx_nnls_work,
                    E->gxb_v[0], E->gxb_v[1], E->gxb_v[2], E->gxb_v[3], E->gxb_vt, E->gxb_q[0], E->gxb_q[1],
                    E->kl_S[0], E->kl_S[1], E->kl_DV[0], E->kl_DV[1], E->sk[0].seg, E->sk[0].first, E->sk[0].cnt, E->sk[0].slabs, E->sk[1].seg, E->sk[1].first, E->sk[1].cnt, E->sk[1].slabs, E->minv_img, E->gx_gimg};
    for (void* b : bufs) if (b) hipFree(b);
    if (E->own_stream) hipStreamDestroy(E->own_stream);
    delete E;
    return NMFX_OK;
}

int nmfx_set_stream(nmfx_handle_t E, void* s) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    E->stream = reinterpret_cast<hipStream_t>(s);      // NULL = HIP's default (null) stream
    return NMFX_OK;
}

int nmfx_reset_stream(nmfx_handle_t E) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    E->stream = E->own_stream;
    return NMFX_OK;
}

int nmfx_set_precision(nmfx_handle_t E, int mode) {
    if (E) { E->himg_both = false; E->wimg_ok = false; E->gxb_img_ready = false; }
    if (!E || (mode != 0 && mode != 1)) { if (E) E->err = "precision must be 0 (f32) or 1 (split bf16)"; return NMFX_E_ARG; }
    if (E->precision != mode) nmfx_comm_invalidate(E);
    E->precision = mode;
    return NMFX_OK;
}

const char* nmfx_get_note(nmfx_handle_t E) { return E ? E->note.c_str() : ""; }

int nmfx_get_precision(nmfx_handle_t E) {
    if (!E) return NMFX_E_ARG;
    return (E->precision == 1 && nmfx_bf16_supported(E)) ? 1 : 0;
}

int nmfx_synchronize(nmfx_handle_t E) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

int nmfx_upload_v(nmfx_handle_t E, const void* host, int dtype, int64_t ld, int64_t row0, int64_t rows) {
    if (E) E->anls_a_ready = false;
    if (!E) return NMFX_E_ARG;
    if (!host || row0 < 0 || rows < 0 || row0 + rows > E->m || ld < E->n) {
        E->err = "upload_v: bad row range or leading dimension"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    if (rows == 0) return NMFX_OK;
    if (E->have_v) { int rc_ = nmfx_need_v(E); if (rc_) return rc_; }       // (rows of a V whose row-major copy was dropped)
    float* dst = E->V + row0 * E->np;
    if (dtype == NMFX_F32) {
        NMFX_HIP(hipMemcpy2DAsync(dst, (size_t)E->np * 4, host, (size_t)ld * 4, (size_t)E->n * 4,
                                  (size_t)rows, hipMemcpyHostToDevice, E->stream));
        NMFX_HIP(hipStreamSynchronize(E->stream));
    } else if (dtype == NMFX_F64) {
        const int64_t chunk = std::max<int64_t>(1, (int64_t)(64 << 20) / (E->n * 8));
        double* stage = nullptr;
        NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&stage), (size_t)std::min(chunk, rows) * E->n * 8));
        const double* src = static_cast<const double*>(host);
        for (int64_t r = 0; r < rows; r += chunk) {
            const int64_t cnt = std::min(chunk, rows - r);
            hipError_t e = hipMemcpy2DAsync(stage, (size_t)E->n * 8, src + r * ld, (size_t)ld * 8,
                                            (size_t)E->n * 8, (size_t)cnt, hipMemcpyHostToDevice, E->stream);
            if (e != hipSuccess) { hipFree(stage); E->err = hipGetErrorString(e); return NMFX_E_HIP; }
            dim3 grid((unsigned)((E->n + 255) / 256), (unsigned)cnt);
            hipLaunchKernelGGL(cvt_f64_rows_kernel, grid, dim3(256), 0, E->stream, stage, E->n, E->n,
                               dst + r * E->np, E->np, cnt);
            e = hipStreamSynchronize(E->stream);
            if (e != hipSuccess) { hipFree(stage); E->err = hipGetErrorString(e); return NMFX_E_HIP; }
        }
        hipFree(stage);
    } else { E->err = "upload_v: dtype must be NMFX_F32 or NMFX_F64"; return NMFX_E_ARG; }
    E->have_v = true;
    E->bf_ready = false;
    E->gxb_v_ready = false;
    E->gxb_vt_ready = false;
    return NMFX_OK;
}

int nmfx_upload_v_device(nmfx_handle_t E, const void* dev, int dtype, int64_t ld, int64_t row0, int64_t rows) {
    if (E) E->anls_a_ready = false;
    if (!E) return NMFX_E_ARG;
    if (!dev || row0 < 0 || rows < 0 || row0 + rows > E->m || ld < E->n) {
        E->err = "upload_v_device: bad row range or leading dimension"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    if (rows == 0) return NMFX_OK;
    if (E->have_v) { int rc_ = nmfx_need_v(E); if (rc_) return rc_; }
    float* dst = E->V + row0 * E->np;
    if (dtype == NMFX_F32) {
        NMFX_HIP(hipMemcpy2DAsync(dst, (size_t)E->np * 4, dev, (size_t)ld * 4, (size_t)E->n * 4,
                                  (size_t)rows, hipMemcpyDeviceToDevice, E->stream));
    } else if (dtype == NMFX_F64) {
        for (int64_t r = 0; r < rows; r += 32768) {          // (grid.y limit)
            const int64_t cnt = std::min<int64_t>(32768, rows - r);
            dim3 grid((unsigned)((E->n + 255) / 256), (unsigned)cnt);
            hipLaunchKernelGGL(cvt_f64_rows_kernel, grid, dim3(256), 0, E->stream,
                               static_cast<const double*>(dev) + r * ld, ld, E->n, dst + r * E->np, E->np, cnt);
        }
        NMFX_HIP(hipGetLastError());
    } else { E->err = "upload_v_device: dtype must be NMFX_F32 or NMFX_F64"; return NMFX_E_ARG; }
    NMFX_HIP(hipStreamSynchronize(E->stream));
    E->have_v = true;
    E->bf_ready = false;
    E->gxb_v_ready = false;
    E->gxb_vt_ready = false;
    return NMFX_OK;
}

static int put_padded(nmfx_engine* E, float* dst, const double* src, int64_t rows, int64_t cols,
                      int64_t prow, int64_t pcol) {
    std::vector<float> tmp((size_t)prow * pcol, 0.f);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) tmp[(size_t)r * pcol + c] = (float)src[r * cols + c];
    NMFX_HIP(hipMemcpyAsync(dst, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

static int get_padded(nmfx_engine* E, const float* src, double* dst, int64_t rows, int64_t cols,
                      int64_t prow, int64_t pcol) {
    std::vector<float> tmp((size_t)prow * pcol);
    NMFX_HIP(hipMemcpyAsync(tmp.data(), src, tmp.size() * 4, hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) dst[r * cols + c] = (double)tmp[(size_t)r * pcol + c];
    return NMFX_OK;
}

int nmfx_set_factors(nmfx_handle_t E, const double* w, const double* hmat) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if (w) {
        if ((rc = put_padded(E, E->W[0], w, E->m, E->k, E->mp, E->kp))) return rc;
        NMFX_HIP(hipMemsetAsync(E->W[1], 0, (size_t)E->mp * E->kp * 4, E->stream));
    }
    if (hmat) { if ((rc = put_padded(E, E->H, hmat, E->k, E->n, E->kp, E->np))) return rc; }
    // (S, DV: the m x n auxiliaries of the KL-loss ADMM variants start from zero, ao_admm.py:17-30)
    float* zero[] = {E->dualW, E->dualH, E->auxW, E->auxH, E->S, E->DV, E->kl_S[0], E->kl_DV[0], E->kl_S[1], E->kl_DV[1]};
    const int64_t mn = E->mp * E->np;
    const int64_t zc[] = {E->mp * E->kp, E->kp * E->np, E->mp * E->kp, E->kp * E->np, mn, mn, mn, mn, mn, mn};
    E->kl_side = 0; E->kl_s_side = 0;                  // (split-bf16 form of the same state: zero in both orientations)
    for (int i = 0; i < 10; ++i)
        if (zero[i]) NMFX_HIP(hipMemsetAsync(zero[i], 0, (size_t)zc[i] * 4, E->stream));
    hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, E->stream, E->state);
    E->wsel = 0;
    E->have_f = true;
    E->bf_ready = false;
    E->gxb_img_ready = false;
    E->kl_h_iter = -2;
    E->wimg_ok = false;
    E->family = 0;
    E->pair = false;
    E->family_started = false;
    E->himg_both = false;
    E->lazy_objective = false;
    E->anls_a_ready = false;
    if (E->kp <= 128 && (rc = nmfx_launch_gram_nt(E, E->H, E->np, E->np, E->HHt_part, E->gsplit))) return rc;
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

static int read_state(nmfx_engine* E, DevState* hs) {
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_HIP(hipMemcpyAsync(hs, E->state, sizeof(DevState), hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

int nmfx_get_factors(nmfx_handle_t E, double* w, double* hmat) {
    if (!E) return NMFX_E_ARG;
    DevState hs; int rc;
    if ((rc = read_state(E, &hs))) return rc;
    if (hs.flag && !E->w_in_place) E->wsel = (int)((hs.stop_i + 1) & 1);
    if (w) { if ((rc = get_padded(E, E->W[E->wsel], w, E->m, E->k, E->mp, E->kp))) return rc; }
    if (hmat) { if ((rc = get_padded(E, E->H, hmat, E->k, E->n, E->kp, E->np))) return rc; }
    return NMFX_OK;
}

int nmfx_get_matrix(nmfx_handle_t E, const char* name, double* out) {
    if (!E || !name || !out) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    const std::string s(name);
    const float* src = nullptr; bool wlike = false;
    if (s == "dual_w") { src = E->dualW; wlike = true; }
    else if (s == "dual_h") { src = E->dualH; }
    else if (s == "w_aux") { src = E->auxW; wlike = true; }
    else if (s == "h_aux") { src = E->auxH; }
    else { E->err = "get_matrix: unknown name"; return NMFX_E_ARG; }
    if (!src) { E->err = "get_matrix: matrix not allocated for this solver"; return NMFX_E_STATE; }
    return wlike ? get_padded(E, src, out, E->m, E->k, E->mp, E->kp)
                 : get_padded(E, src, out, E->k, E->n, E->kp, E->np);
}

int nmfx_set_matrix(nmfx_handle_t E, const char* name, const double* in) {
    if (!E || !name || !in) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    int rc;
    if ((rc = nmfx_admm_state_alloc(E))) return rc;
    const std::string s(name);
    float* dst = nullptr; bool wlike = false;
    if (s == "dual_w") { dst = E->dualW; wlike = true; }
    else if (s == "dual_h") { dst = E->dualH; }
    else if (s == "w_aux") { dst = E->auxW; wlike = true; }
    else if (s == "h_aux") { dst = E->auxH; }
    else { E->err = "set_matrix: unknown name"; return NMFX_E_ARG; }
    return wlike ? put_padded(E, dst, in, E->m, E->k, E->mp, E->kp) : put_padded(E, dst, in, E->k, E->n, E->kp, E->np);
}

int nmfx_get_state(nmfx_handle_t E, int* stop_rule, int64_t* stop_i, int64_t* n_obj) {
    if (!E) return NMFX_E_ARG;
    DevState hs; int rc;
    if ((rc = read_state(E, &hs))) return rc;
    if (hs.notpd) { E->err = "Gram + rho I is not positive definite"; return NMFX_E_NOTPD; }
    if (stop_rule) *stop_rule = hs.flag;
    if (stop_i) *stop_i = hs.stop_i;
    if (n_obj) *n_obj = hs.n_obj;
    // the iterate the reference would return: W_{stop_i+1} once stopped
    if (hs.flag && !E->w_in_place) E->wsel = (int)((hs.stop_i + 1) & 1);
    return NMFX_OK;
}

int nmfx_get_diagnostics(nmfx_handle_t E, int64_t* nnls_evicted, int64_t* nnls_capped) {
    if (!E) return NMFX_E_ARG;
    DevState hs; int rc;
    if ((rc = read_state(E, &hs))) return rc;
    if (nnls_evicted) *nnls_evicted = hs.nnls_evicted;
    if (nnls_capped) *nnls_capped = hs.nnls_capped;
    return NMFX_OK;
}

int nmfx_get_nnls_fallbacks(nmfx_handle_t E, int64_t* problems, int64_t* half_steps) {
    if (!E) return NMFX_E_ARG;
    DevState hs; int rc;
    if ((rc = read_state(E, &hs))) return rc;
    if (problems) *problems = hs.nnls_fallback;
    if (half_steps) *half_steps = hs.nnls_noinv;
    return NMFX_OK;
}

int nmfx_get_inner_paths(nmfx_handle_t E, int64_t out[4]) {
    if (!E || !out) return NMFX_E_ARG;
    DevState hs; int rc;
    if ((rc = read_state(E, &hs))) return rc;
    for (int i = 0; i < 4; ++i) out[i] = hs.ao_paths[i];
    return NMFX_OK;
}

int nmfx_get_objectives(nmfx_handle_t E, int64_t first, int64_t count, double* out) {
    if (!E || !out || first < 0 || count < 0 || first + count > E->obj_cap) {
        if (E) E->err = "get_objectives: range"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    if (count == 0) return NMFX_OK;
    NMFX_HIP(hipMemcpyAsync(out, E->obj_hist + first, (size_t)count * 8, hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

int nmfx_get_inner_counts(nmfx_handle_t E, int64_t first, int64_t count, int32_t* out) {
    if (!E || !out || first < 0 || count < 0 || first + count > E->inner_cap) {
        if (E) E->err = "get_inner_counts: range"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    if (count == 0) return NMFX_OK;
    NMFX_HIP(hipMemcpyAsync(out, E->inner_hist + first * 2, (size_t)count * 8, hipMemcpyDeviceToHost, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

// ---- exchange buffers ----------------------------------------------------
int nmfx_exchange_sizes(nmfx_handle_t E, int64_t* n_f32, int64_t* n_f64) {
    if (!E) return NMFX_E_ARG;
    if (n_f32) *n_f32 = (int64_t)E->kp * E->np + (int64_t)E->kp * E->kp + E->kp + NMFX_XTAIL;
    if (n_f64) *n_f64 = 8 + 4 * NMFX_MAX_FUSED_ROUNDS;
    return NMFX_OK;
}

int nmfx_set_exchange_rank(nmfx_handle_t E, int rank, int world) {
    if (!E) return NMFX_E_ARG;
    if (world == 0) { E->xrank = -1; E->xworld = 0; return NMFX_OK; }           // back to the separate f64 exchange
    if (world < 1 || world > NMFX_XTAIL_RANKS || rank < 0 || rank >= world) { E->err = "set_exchange_rank: 1 <= world <= 64, 0 <= rank < world"; return NMFX_E_ARG; }
    if (!(E->precision == 1 && nmfx_bf16_supported(E))) { E->err = "set_exchange_rank: only with the split-bf16 epilogues (k padded to 64 / 128)"; return NMFX_E_STATE; }
    E->xrank = rank; E->xworld = world;
    return NMFX_OK;
}

int nmfx_set_exchange_buffers(nmfx_handle_t E, void* f32, int64_t n_f32, void* f64, int64_t n_f64) {
    if (!E || !f32 || !f64) return NMFX_E_ARG;
    int64_t need32, need64;
    nmfx_exchange_sizes(E, &need32, &need64);
    if (n_f32 < need32 || n_f64 < need64) {
        E->err = "set_exchange_buffers: allocations smaller than nmfx_exchange_sizes (the buffers grew in round 2: f64 tail of 4 x 64 "
                 "norm sums, f32 tail of 256 floats)";
        return NMFX_E_ARG;
    }
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    if (E->own_x) { hipFree(E->xf32); hipFree(E->xf64); }
    E->xf32 = static_cast<float*>(f32); E->xf64 = static_cast<double*>(f64); E->own_x = false;
    return NMFX_OK;
}

int nmfx_get_exchange_buffers(nmfx_handle_t E, void** f32, void** f64) {
    if (!E) return NMFX_E_ARG;
    if (f32) *f32 = E->xf32;
    if (f64) *f64 = E->xf64;
    return NMFX_OK;
}

// ---- the f64 referee of the stop rule (kernels_generic.hip: nmfx_objective_f64) ---------------
int nmfx_set_stop_guard(nmfx_handle_t E, double guard) {
    if (!E || !(guard >= 0.0)) { if (E) E->err = "set_stop_guard: guard >= 0"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    hipLaunchKernelGGL(set_guard_kernel, dim3(1), dim3(1), 0, E->stream, E->state, guard);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

int nmfx_resume(nmfx_handle_t E) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    hipLaunchKernelGGL(resume_kernel, dim3(1), dim3(1), 0, E->stream, E->state);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- iteration base (hipGraph replay support) --------------------------------
int nmfx_reserve_objectives(nmfx_handle_t E, int64_t count) {
    if (!E || count < 0) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    return nmfx_ensure_obj_capacity(E, count + 2);
}

int nmfx_shift_iteration_base(nmfx_handle_t E, int64_t delta) {
    if (!E) return NMFX_E_ARG;
    NMFX_HIP(hipSetDevice(E->device));
    hipLaunchKernelGGL(shift_iteration_base_kernel, dim3(1), dim3(1), 0, E->stream, E->state, (long long)delta);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- MUR -------------------------------------------------------------------
static int check_ready(nmfx_engine* E, int64_t first, int64_t count, bool mur_entry = true) {
    E->anls_a_ready = false;                           // (another solver's products overwrite A_part)
    if (!E->have_v || !E->have_f) { E->err = "upload V and set factors first"; return NMFX_E_STATE; }
    if (mur_entry) {
        int rc_ = nmfx_enter_family(E, 1); if (rc_) return rc_;
        if (E->pair) { E->err = "this handle runs two stacked problems (nmfx_mur_pair_run): nmfx_set_factors before a single-problem run"; return NMFX_E_STATE; }
        E->family_started = true;
    }
    if (first < 0 || count < 0) { E->err = "negative iteration range"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    return nmfx_ensure_obj_capacity(E, first + count + 2);
}

// (kl_h_iter: the H images and row-sum partials the KL H epilogue of iteration j leaves for the KL W phase of iteration j + 1 are
// only good if NOTHING else has written H or those buffers in between -- every entry point of another solver, and the
// Euclidean MUR phases, void them; ADVICE r2: only nmfx_set_factors did)
int nmfx_mur_phase_a(nmfx_handle_t E, int distance, double lambda_w, int64_t j) {
    if (E) E->himg_both = false;
    if (!E) return NMFX_E_ARG;
    if (distance != NMFX_KL) E->kl_h_iter = -2;
    int rc = check_ready(E, j, 1); if (rc) return rc;
    if (E->kp > 128 && (distance == NMFX_EU || distance == NMFX_KL)) return nmfx_generic_mur_phase_a(E, distance, lambda_w, j);
    if (distance == NMFX_EU)
        return (E->precision == 1 && nmfx_bf16_supported(E)) ? nmfx_mur_eu_phase_a_bf16(E, lambda_w, j)
                                                             : nmfx_mur_eu_phase_a(E, lambda_w, j);
    if (distance == NMFX_KL)
        return (E->precision == 1 && nmfx_bf16_supported(E)) ? nmfx_mur_kl_phase_a_bf16(E, lambda_w, j)
                                                             : nmfx_mur_kl_phase_a(E, lambda_w, j);
    E->err = "Unknown distance type."; return NMFX_E_ARG;
}

// Phase A in pieces (chunked exchange; Euclidean loss on the split-bf16 path only -- nmfx_mur_chunk_info says whether)
static bool mur_chunkable(nmfx_engine* E, int distance) {
    return distance == NMFX_EU && E->precision == 1 && nmfx_bf16_supported(E);
}
int nmfx_mur_chunk_info(nmfx_handle_t E, int distance, int64_t* unit, int64_t* n_padded, int64_t* k_padded) {
    if (!E) return NMFX_E_ARG;
    const bool ok = mur_chunkable(E, distance);
    if (unit) *unit = ok ? 128 : 0;
    if (n_padded) *n_padded = E->np;
    if (k_padded) *k_padded = E->kp;
    return NMFX_OK;
}
int nmfx_mur_phase_a_head(nmfx_handle_t E, int distance, double lambda_w, int64_t j) {
    if (!E) return NMFX_E_ARG;
    E->himg_both = false;
    E->kl_h_iter = -2;
    int rc = check_ready(E, j, 1); if (rc) return rc;
    if (!mur_chunkable(E, distance)) { E->err = "phase_a_head: Euclidean loss on the split-bf16 path only (nmfx_mur_chunk_info)"; return NMFX_E_ARG; }
    return nmfx_mur_eu_phase_a_head_bf16(E, lambda_w, j);
}
int nmfx_mur_phase_a_cols(nmfx_handle_t E, int distance, int64_t c0, int64_t c1) {
    if (!E) return NMFX_E_ARG;
    if (!mur_chunkable(E, distance)) { E->err = "phase_a_cols: Euclidean loss on the split-bf16 path only (nmfx_mur_chunk_info)"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    return nmfx_mur_eu_phase_a_cols_bf16(E, c0, c1);
}

int nmfx_mur_phase_b(nmfx_handle_t E, int distance, double lambda_h, int64_t min_iter, double tol1,
                     double tol2, int64_t j) {
    if (E) E->himg_both = false;
    if (!E) return NMFX_E_ARG;
    if (distance != NMFX_KL) E->kl_h_iter = -2;
    int rc = check_ready(E, j, 1); if (rc) return rc;
    E->wsel = (int)((j + 1) & 1);
    E->w_in_place = false;
    if (E->kp > 128 && (distance == NMFX_EU || distance == NMFX_KL)) return nmfx_generic_mur_phase_b(E, distance, lambda_h, min_iter, tol1, tol2, j);
    if (distance == NMFX_EU)
        return (E->precision == 1 && nmfx_bf16_supported(E))
                   ? nmfx_mur_eu_phase_b_bf16(E, lambda_h, min_iter, tol1, tol2, j)
                   : nmfx_mur_eu_phase_b(E, lambda_h, min_iter, tol1, tol2, j);
    if (distance == NMFX_KL)
        return (E->precision == 1 && nmfx_bf16_supported(E))
                   ? nmfx_mur_kl_phase_b_bf16(E, lambda_h, min_iter, tol1, tol2, j)
                   : nmfx_mur_kl_phase_b(E, lambda_h, min_iter, tol1, tol2, j);
    E->err = "Unknown distance type."; return NMFX_E_ARG;
}

// Phase B by column slices (reduce-scatter / all-gather exchange, include/nmfx.h)
int nmfx_mur_slice_info(nmfx_handle_t E, int distance, int world, int64_t* cols, int64_t* elems) {
    if (!E || world < 1) return NMFX_E_ARG;
    const bool ok = mur_chunkable(E, distance) && E->xworld == world && (E->np / 64) % world == 0;
    const int64_t c = ok ? E->np / world : 0;
    if (cols) *cols = c;
    if (elems) *elems = c * E->kp;
    return NMFX_OK;
}
static int slice_args(nmfx_engine* E, int distance, int64_t c0, int64_t c1, const char* who) {
    if (!mur_chunkable(E, distance) || E->xworld <= 0) { E->err = std::string(who) + ": Euclidean loss on the split-bf16 path with nmfx_set_exchange_rank in force only (nmfx_mur_slice_info)"; return NMFX_E_ARG; }
    if (c0 < 0 || c1 < c0 || c1 > E->np || c0 % 64 || c1 % 64) { E->err = std::string(who) + ": column range must be whole 64-column blocks inside the padded n"; return NMFX_E_ARG; }
    return NMFX_OK;
}
int nmfx_mur_phase_b_slice(nmfx_handle_t E, int distance, double lambda_h, int64_t min_iter, double tol1, double tol2, int64_t j,
                           int64_t c0, int64_t c1) {
    if (!E) return NMFX_E_ARG;
    E->himg_both = false;
    E->kl_h_iter = -2;
    int rc = check_ready(E, j, 1); if (rc) return rc;
    if ((rc = slice_args(E, distance, c0, c1, "phase_b_slice"))) return rc;
    E->wsel = (int)((j + 1) & 1);
    E->w_in_place = false;
    return nmfx_mur_eu_phase_b_slice_bf16(E, lambda_h, min_iter, tol1, tol2, j, (int)(c0 / 64), (int)((c1 - c0) / 64));
}
int nmfx_mur_phase_b_rest(nmfx_handle_t E, int distance, int64_t c0, int64_t c1) {
    if (!E) return NMFX_E_ARG;
    int rc = slice_args(E, distance, c0, c1, "phase_b_rest"); if (rc) return rc;
    NMFX_HIP(hipSetDevice(E->device));
    return nmfx_mur_eu_phase_b_rest_bf16(E, (int)(c0 / 64), (int)((c1 - c0) / 64));
}

int nmfx_mur_finish_a(nmfx_handle_t E, int distance, int64_t j) {
    if (!E) return NMFX_E_ARG;
    int rc = check_ready(E, j, 1); if (rc) return rc;
    if (E->kp > 128 && (distance == NMFX_EU || distance == NMFX_KL)) return nmfx_generic_mur_finish_a(E, distance, j);
    if (distance == NMFX_EU) return nmfx_mur_eu_finish_a(E, j);
    if (distance == NMFX_KL) return nmfx_mur_kl_finish_a(E, j);
    E->err = "Unknown distance type."; return NMFX_E_ARG;
}

int nmfx_mur_finish_b(nmfx_handle_t E, int64_t min_iter, double tol1, double tol2, int64_t j) {
    if (!E) return NMFX_E_ARG;
    int rc = check_ready(E, j, 1, false); if (rc) return rc;       // (the closing step of EVERY row-sharded solver, not only MUR's)
    return nmfx_finish_b(E, min_iter, tol1, tol2, j);
}

int nmfx_mur_run(nmfx_handle_t E, int distance, double lambda_w, double lambda_h, int64_t min_iter,
                 double tol1, double tol2, int64_t first, int64_t count) {
    if (!E) return NMFX_E_ARG;
    int rc = check_ready(E, first, count); if (rc) return rc;
    E->fused_pack = true;            // nothing is exchanged between the phases here
    for (int64_t j = first; j < first + count && !rc; ++j) {
        if ((rc = nmfx_mur_phase_a(E, distance, lambda_w, j))) break;
        rc = nmfx_mur_phase_b(E, distance, lambda_h, min_iter, tol1, tol2, j);
    }
    E->fused_pack = false;
    return rc;
}

int nmfx_mur_finish(nmfx_handle_t E, int distance, int64_t min_iter, double tol1, double tol2,
                    int64_t iters_done) {
    int rc;
    if ((rc = nmfx_mur_finish_a(E, distance, iters_done))) return rc;
    return nmfx_mur_finish_b(E, min_iter, tol1, tol2, iters_done);
}

// ---- profiling -------------------------------------------------------------
static void drain_profile(nmfx_engine* E) {
    if (E->prof_pending.empty()) return;
    hipStreamSynchronize(E->stream);
    for (auto& t : E->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, std::get<1>(t), std::get<2>(t)) == hipSuccess) {
            ProfSlot& s = E->prof_slots[std::get<0>(t)];
            s.ms += ms; s.n += 1;
        }
        hipEventDestroy(std::get<1>(t)); hipEventDestroy(std::get<2>(t));
    }
    E->prof_pending.clear();
}

int nmfx_profile_enable(nmfx_handle_t E, int on) {
    if (!E) return NMFX_E_ARG;
    drain_profile(E);
    E->prof = on != 0;
    return NMFX_OK;
}

int nmfx_profile_get(nmfx_handle_t E, const char* name, double* total_ms, int64_t* launches) {
    if (!E || !name) return NMFX_E_ARG;
    drain_profile(E);
    auto it = E->prof_slots.find(name);
    if (total_ms) *total_ms = it == E->prof_slots.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == E->prof_slots.end() ? 0 : it->second.n;
    return NMFX_OK;
}

// `reps` back-to-back launches of ONE product kernel of the current state between a single pair of HIP events:
// the per-launch time of the dominant kernel without an event (= a command-processor barrier) in front of
// every launch, which costs the W phase ~10 us that no real iteration pays.
int nmfx_profile_repeat(nmfx_handle_t E, const char* which, int distance, int reps, double* ms_per_launch) {
    if (!E || !which || reps <= 0 || !ms_per_launch) return NMFX_E_ARG;
    const std::string w(which);
    int rc = check_ready(E, 0, 0, !(w.rfind("ao_", 0) == 0 || w.rfind("sk_", 0) == 0)); if (rc) return rc;
    const bool wph = w == "wphase";
    // r4, the two V-sized products of AO-ADMM at k padded to 128 (after at least one nmfx_aoadmm_run on the handle, so that the images
    // and Gram slabs exist): "ao_hphase" / "ao_wphase" = the (row block) x (split) launches, "sk_hphase" / "sk_wphase" = the stream-K
    // form without a side job, "..._side" = with the inversion of the Gram slabs beside it
    const bool ao = w.rfind("ao_", 0) == 0 || w.rfind("sk_", 0) == 0;
    if (ao && !(E->precision == 1 && E->kp == 128 && E->bf_ready && E->HThi)) { E->err = "profile_repeat: AO-ADMM products need a split-bf16 k = 128 run first"; return NMFX_E_STATE; }
    if (!ao && !wph && w != "hphase") { E->err = "profile_repeat: which must be wphase or hphase"; return NMFX_E_ARG; }
    if (distance != NMFX_EU && distance != NMFX_KL) { E->err = "Unknown distance type."; return NMFX_E_ARG; }
    const bool bf = E->precision == 1 && nmfx_bf16_supported(E);
    const bool kl = distance == NMFX_KL;
    if (bf) {
        if (!E->bf_ready) E->wsel = 0;
        if ((rc = nmfx_bf16_prepare(E))) return rc;
        if (kl && (rc = nmfx_bf16_images_h(E, true))) return rc;          // H^T images: the Z operand of the KL H phase
    }
    const bool was = E->prof;
    E->prof = false;
    hipEvent_t a, b;
    NMFX_HIP(hipEventCreate(&a)); NMFX_HIP(hipEventCreate(&b));
#ifdef NMFX_EXP_REVERSE
    extern int nmfx_debug_set_reverse(void* stream, int v);
    static const int altrev = getenv("NMFX_EXP_ALTREV") ? atoi(getenv("NMFX_EXP_ALTREV")) : 0;
#endif
    for (int i = -2; i < reps && !rc; ++i) {                               // two untimed launches first
        if (i == 0) NMFX_HIP(hipEventRecord(a, E->stream));
#ifdef NMFX_EXP_REVERSE
        nmfx_debug_set_reverse(E->stream, altrev ? (i & 1) : 0);          // (the setter launch is there in both variants)
#endif
        if (ao) {
            const bool side = w.find("_side") != std::string::npos, hside = w.find("hphase") != std::string::npos;
            if (w.rfind("ao_", 0) == 0) rc = hside ? nmfx_bf16_vtw(E, true, "hphase", false, 3) : nmfx_bf16_vht(E, false, 0, "wphase_noobj", false, 3);
            else rc = nmfx_bf16_sk_product(E, hside ? 0 : 1, hside, side ? (hside ? E->G_part : E->HHt_part) : nullptr, hside ? 64 : 32, -1.0,
                                           hside ? "hphase" : "wphase_noobj");
        }
        else if (bf) rc = wph ? nmfx_bf16_vht(E, true, E->wsel, "wphase", kl, 3) : nmfx_bf16_vtw(E, false, "hphase", kl, 3);
        else if (kl) { E->err = "profile_repeat: KL only in the split-bf16 mode"; rc = NMFX_E_ARG; }
        else rc = wph ? nmfx_launch_wphase(E, E->W[E->wsel], true, true) : nmfx_launch_hphase(E, E->W[E->wsel], nmfx_hphase_can_fuse_gram(E));
    }
    E->prof = was;
    if (rc) { hipEventDestroy(a); hipEventDestroy(b); return rc; }
    NMFX_HIP(hipEventRecord(b, E->stream));
    NMFX_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    NMFX_HIP(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    *ms_per_launch = (double)ms / reps;
    return NMFX_OK;
}

int nmfx_profile_reset(nmfx_handle_t E) {
    if (!E) return NMFX_E_ARG;
    drain_profile(E);
    E->prof_slots.clear();
    return NMFX_OK;
}

}  // extern "C"
