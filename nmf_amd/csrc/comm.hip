// The exchange step of the row-sharded solvers behind the C ABI: an RCCL communicator per handle (one process per GPU),
// sum-all-reduces of the handle's exchange buffers IN PLACE on the handle's stream, and the sharded MUR loop as ONE C call
// (phase A . all-reduce . phase B per outer iteration, optionally replayed as a hipGraph).
//
//   reference: none -- the reference is single-process numpy.  This is north_star's "shard rows of V and W across the 8 GPUs
//   of one node with an RCCL all-reduce over xGMI of the k x k Gram W^T W and the k x n product W^T V each outer iteration"
//   (nmf/mur.py:45 needs w.T @ x and w.T @ w over ALL rows), SURVEY 8b's nmfx_create(dev_ids[], ...) / NMFX_E_RCCL sketch
//   in the one-process-per-GPU form.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME both ROCm's and PyTorch's copy carry: inside a torch process
// the copy torch has already mapped is the one that is found, so there is ONE RCCL and one HIP runtime per process), hence
// libnmfx.so itself links only the HIP runtime and single-GPU users never need RCCL installed.
#include "nmfx_internal.h"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string why;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("NMFX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm || !*nm) continue;
            api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
            api.why = dlerror();
        }
        if (!api.lib) return;
        auto sym = [&](const char* s) { void* p = dlsym(api.lib, s); if (!p) { api.why = std::string("missing symbol ") + s; } return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.ReduceScatter = reinterpret_cast<decltype(api.ReduceScatter)>(sym("ncclReduceScatter"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.ReduceScatter || !api.AllGather ||
            !api.GroupStart || !api.GroupEnd || !api.GetErrorString) { dlclose(api.lib); api.lib = nullptr; }
    });
    return api;
}

}  // namespace

struct nmfx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t side = nullptr;           // the chunked exchange runs its collectives here, behind events of the handle's stream
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    int64_t* bar = nullptr;               // device word of nmfx_comm_barrier
    // settled by nmfx_comm_negotiate (identical on every rank)
    bool negotiated = false;
    bool merged = false;                  // objective partial inside the f32 buffer: one collective per MUR-eu iteration
    int64_t chunk_unit = 0;
    int exchange = 0;                     // nmfx_comm_set_exchange: 0 = one sum-all-reduce per iteration, 1 = reduce-scatter . sliced H update . all-gather
    // hipGraph replay of iteration pairs (nmfx_comm_set_graph)
    bool want_graph = false, graph_failed = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    double g_lw = 0, g_lh = 0, g_t1 = 0, g_t2 = 0; int64_t g_min = 0; int g_dist = -1; int g_chunks = 0; int g_precision = -1; int g_exchange = -1;
    const void* g_hist = nullptr; hipStream_t g_stream = nullptr;     // what the captured nodes point at
    bool g_klfresh = false;               // captured with "the previous iteration's KL epilogue left its images and sums" (kl_h_iter)
    int64_t replays = 0;
};

#define NMFX_RCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
    E->err = std::string(#expr) + ": " + rccl().GetErrorString(r_) + " (" __FILE__ ":" NMFX_STR(__LINE__) ")"; return NMFX_E_RCCL; } } while (0)

static int need_rccl(nmfx_engine* E) {
    if (rccl().lib) return NMFX_OK;
    if (E) E->err = "RCCL is not available (dlopen librccl.so.1: " + rccl().why + "); set NMFX_RCCL_LIB to its path";
    return NMFX_E_RCCL;
}

static void drop_graph(nmfx_comm* c) {
    if (c->exec) { hipGraphExecDestroy(c->exec); c->exec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
}

// nmfx_set_precision: what nmfx_comm_negotiate settled -- the merged objective, the chunk unit -- and a captured graph
// belong to the arithmetic mode they were settled in.  The exact-f32 epilogues put the objective partial into xf64, which a merged
// iteration does not reduce; a replayed graph would run the old mode's kernels.
void nmfx_comm_invalidate(nmfx_engine* E) {
    nmfx_comm* c = E->comm;
    if (!c) return;
    c->negotiated = false; c->merged = false;
    drop_graph(c);
    nmfx_set_exchange_rank(E, 0, 0);
}

void nmfx_comm_free(nmfx_engine* E) {      // nmfx_destroy
    nmfx_comm* c = E->comm;
    if (!c) return;
    hipSetDevice(E->device);
    hipStreamSynchronize(E->stream);
    drop_graph(c);
    if (c->comm && rccl().lib) rccl().CommDestroy(c->comm);
    if (c->side) hipStreamDestroy(c->side);
    if (c->ev_ready) hipEventDestroy(c->ev_ready);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->bar) hipFree(c->bar);
    delete c;
    E->comm = nullptr;
}

extern "C" int nmfx_comm_unique_id(void* id128) {
    if (!id128) return NMFX_E_ARG;
    if (need_rccl(nullptr)) return NMFX_E_RCCL;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return NMFX_E_RCCL;
    std::memcpy(id128, &id, sizeof(id));
    return NMFX_OK;
}

extern "C" int nmfx_comm_init_rank(nmfx_handle_t E, const void* id128, int rank, int world) {
    if (!E) return NMFX_E_ARG;
    if (!id128 || world < 1 || rank < 0 || rank >= world) { E->err = "comm_init_rank: 0 <= rank < world, id from nmfx_comm_unique_id of rank 0"; return NMFX_E_ARG; }
    int rc = need_rccl(E); if (rc) return rc;
    if (E->comm) { E->err = "comm_init_rank: this handle already has a communicator (nmfx_comm_destroy first)"; return NMFX_E_STATE; }
    NMFX_HIP(hipSetDevice(E->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    nmfx_comm* c = new nmfx_comm;
    c->rank = rank; c->world = world;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        E->err = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r);
        delete c;
        return NMFX_E_RCCL;
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->bar), 64) != hipSuccess || hipMemset(c->bar, 0, 64) != hipSuccess) {
        E->err = "comm_init_rank: stream / event creation failed";
        E->comm = c; nmfx_comm_free(E);
        return NMFX_E_HIP;
    }
    E->comm = c;
    return NMFX_OK;
}

extern "C" int nmfx_comm_destroy(nmfx_handle_t E) {
    if (!E) return NMFX_E_ARG;
    nmfx_comm_free(E);
    nmfx_set_exchange_rank(E, 0, 0);
    return NMFX_OK;
}

extern "C" int nmfx_comm_info(nmfx_handle_t E, int* rank, int* world, int* merged, int* rccl_version) {
    if (!E) return NMFX_E_ARG;
    if (rank) *rank = E->comm ? E->comm->rank : 0;
    if (world) *world = E->comm ? E->comm->world : 1;
    if (merged) *merged = E->comm && E->comm->merged ? 1 : 0;
    if (rccl_version) { *rccl_version = 0; if (rccl().lib && rccl().GetVersion) rccl().GetVersion(rccl_version); }
    return NMFX_OK;
}

static int have_comm(nmfx_engine* E) {
    if (!E) return NMFX_E_ARG;
    if (!E->comm) { E->err = "no communicator on this handle: nmfx_comm_init_rank first"; return NMFX_E_STATE; }
    return NMFX_OK;
}

// in-place sum over [first, first + count) of the f32 (which = 0) or f64 (which = 1) exchange buffer, on `stream`
static int reduce_range(nmfx_engine* E, int which, int64_t first, int64_t count, hipStream_t stream) {
    int64_t n32, n64;
    nmfx_exchange_sizes(E, &n32, &n64);
    const int64_t cap = which ? n64 : n32;
    if ((which != 0 && which != 1) || first < 0 || count < 0 || first + count > cap) { E->err = "comm_all_reduce: range outside the exchange buffer"; return NMFX_E_ARG; }
    if (count == 0) return NMFX_OK;
    void* p = which ? static_cast<void*>(E->xf64 + first) : static_cast<void*>(E->xf32 + first);
    NMFX_RCCL(rccl().AllReduce(p, p, (size_t)count, which ? ncclDouble : ncclFloat, ncclSum, E->comm->comm, stream));
    return NMFX_OK;
}

extern "C" int nmfx_comm_all_reduce(nmfx_handle_t E, int which, int64_t first, int64_t count) {
    int rc = have_comm(E); if (rc) return rc;
    NMFX_HIP(hipSetDevice(E->device));
    return reduce_range(E, which, first, count, E->stream);
}

// MIN over the ranks of a few host integers (mode negotiation, test rigs); blocking
extern "C" int nmfx_comm_all_min(nmfx_handle_t E, int64_t* vals, int n) {
    int rc = have_comm(E); if (rc) return rc;
    if (!vals || n < 1 || n > 64) { E->err = "comm_all_min: 1 .. 64 values"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    int64_t* d = nullptr;
    NMFX_HIP(hipMalloc(reinterpret_cast<void**>(&d), (size_t)n * 8));
    int out = NMFX_OK;
    if (hipMemcpyAsync(d, vals, (size_t)n * 8, hipMemcpyHostToDevice, E->stream) != hipSuccess) out = NMFX_E_HIP;
    if (!out && rccl().AllReduce(d, d, (size_t)n, ncclInt64, ncclMin, E->comm->comm, E->stream) != ncclSuccess) { E->err = "comm_all_min: ncclAllReduce failed"; out = NMFX_E_RCCL; }
    if (!out && hipMemcpyAsync(vals, d, (size_t)n * 8, hipMemcpyDeviceToHost, E->stream) != hipSuccess) out = NMFX_E_HIP;
    if (hipStreamSynchronize(E->stream) != hipSuccess && !out) out = NMFX_E_HIP;
    hipFree(d);
    if (out == NMFX_E_HIP) E->err = "comm_all_min: HIP error";
    return out;
}

// Every rank's queued work on the handle's stream has completed and every rank has arrived: a one-word all-reduce on the
// communicator the data path uses (a barrier through ANOTHER communicator -- torch.distributed's -- wakes that one up from idle:
// ~0.5 ms at the end of a timed region), then a stream synchronisation.
extern "C" int nmfx_comm_barrier(nmfx_handle_t E) {
    int rc = have_comm(E); if (rc) return rc;
    NMFX_HIP(hipSetDevice(E->device));
    NMFX_RCCL(rccl().AllReduce(E->comm->bar, E->comm->bar, 1, ncclInt64, ncclSum, E->comm->comm, E->stream));
    NMFX_HIP(hipStreamSynchronize(E->stream));
    return NMFX_OK;
}

// Agree on what fixes the SEQUENCE of collectives (see nmf_amd/dist.py: DeviceShard.negotiate): merged objective exchange, chunk
// unit, arithmetic mode.  NMFX_E_STATE when the ranks' arithmetic modes differ.
extern "C" int nmfx_comm_negotiate(nmfx_handle_t E) {
    int rc = have_comm(E); if (rc) return rc;
    nmfx_comm* c = E->comm;
    const int64_t bf16 = nmfx_get_precision(E) == 1 ? 1 : 0;
    const char* mg = getenv("NMFX_DIST_MERGE");
    int64_t unit = 0;
    nmfx_mur_chunk_info(E, NMFX_EU, &unit, nullptr, nullptr);
    int64_t offer[4] = {(bf16 && c->world <= NMFX_XTAIL_RANKS && !(mg && atoi(mg) == 0)) ? 1 : 0, unit, bf16, -bf16};
    if ((rc = nmfx_comm_all_min(E, offer, 4))) return rc;
    if (offer[2] != -offer[3]) {
        E->err = "row-sharded run: the ranks run different arithmetic modes (one fell back to the exact-f32 kernels, nmfx_get_note); "
                 "set NMFX_PRECISION=f32 on every rank or free that GPU's memory";
        return NMFX_E_STATE;
    }
    c->merged = offer[0] != 0;
    c->chunk_unit = offer[1];
    c->negotiated = true;
    drop_graph(c);
    return c->merged ? nmfx_set_exchange_rank(E, c->rank, c->world) : nmfx_set_exchange_rank(E, 0, 0);
}

extern "C" int nmfx_comm_set_graph(nmfx_handle_t E, int enable) {
    int rc = have_comm(E); if (rc) return rc;
    E->comm->want_graph = enable != 0;
    if (!enable) drop_graph(E->comm);
    return NMFX_OK;
}

// The exchange of the sharded MUR loop (nmfx_mur_run_sharded): 0 = one sum-all-reduce of the f32 buffer per iteration (default),
// 1 = reduce-scatter of its W^T V part . H update of this rank's column slice . all-gather (on xGMI's point-to-point
// links the all-reduce IS those two steps; between them every rank now updates n / world columns instead of all n).  Iterations
// for which the sliced update is not available (nmfx_mur_slice_info: KL loss, exact-f32 epilogues, separate objective exchange,
// n not a multiple of 64 * world) take the all-reduce.  Must be the same on every rank (as NMFX_DIST_EXCHANGE is).
extern "C" int nmfx_comm_set_exchange(nmfx_handle_t E, int mode) {
    int rc = have_comm(E); if (rc) return rc;
    if (mode != 0 && mode != 1) { E->err = "comm_set_exchange: 0 (all-reduce) or 1 (reduce-scatter + all-gather)"; return NMFX_E_ARG; }
    if (E->comm->exchange != mode) drop_graph(E->comm);
    E->comm->exchange = mode;
    return NMFX_OK;
}

extern "C" int nmfx_comm_get_exchange(nmfx_handle_t E, int* mode) {
    if (!E || !mode) return NMFX_E_ARG;
    *mode = E->comm ? E->comm->exchange : 0;
    return NMFX_OK;
}

static int chunks_from_env() {
    const char* s = getenv("NMFX_DIST_CHUNKS");
    const int n = s ? atoi(s) : 1;
    return n < 1 ? 1 : n;
}

// one sharded outer iteration j, queued on the handle's stream
static int sharded_iteration(nmfx_engine* E, int distance, double lw, double lh, int64_t min_iter, double tol1, double tol2, int64_t j,
                             int chunks) {
    nmfx_comm* c = E->comm;
    int64_t n32, n64;
    nmfx_exchange_sizes(E, &n32, &n64);
    int rc;
    const bool merged = c->merged && distance == NMFX_EU;
    int64_t unit = 0, npad = 0, kpad = 0;
    if (chunks > 1 && distance == NMFX_EU) nmfx_mur_chunk_info(E, distance, &unit, &npad, &kpad);
    if (unit > c->chunk_unit) unit = c->chunk_unit;
    int64_t step = 0;
    if (unit > 0) {
        step = ((npad + chunks - 1) / chunks) / unit * unit;
        if (step < 512) step = 512;
        if (step >= npad) step = 0;
    }
    int64_t scols = 0, selems = 0;
    if (c->exchange == 1 && merged) nmfx_mur_slice_info(E, distance, c->world, &scols, &selems);
    if (scols > 0) {
        // reduce-scatter . sliced update . all-gather, all in place in the f32 exchange buffer ([column][factor]: a column range is contiguous)
        if ((rc = nmfx_mur_phase_a(E, distance, lw, j))) return rc;
        const int64_t head = selems * c->world;       // = kp * np: the W^T V part
        NMFX_RCCL(rccl().GroupStart());
        ncclResult_t r1 = rccl().ReduceScatter(E->xf32, E->xf32 + c->rank * selems, (size_t)selems, ncclFloat, ncclSum, c->comm, E->stream);
        ncclResult_t r2 = rccl().AllReduce(E->xf32 + head, E->xf32 + head, (size_t)(n32 - head), ncclFloat, ncclSum, c->comm, E->stream);
        NMFX_RCCL(rccl().GroupEnd());
        NMFX_RCCL(r1); NMFX_RCCL(r2);
        if ((rc = nmfx_mur_phase_b_slice(E, distance, lh, min_iter, tol1, tol2, j, c->rank * scols, (c->rank + 1) * scols))) return rc;
        NMFX_RCCL(rccl().AllGather(E->xf32 + c->rank * selems, E->xf32, (size_t)selems, ncclFloat, c->comm, E->stream));
        return nmfx_mur_phase_b_rest(E, distance, c->rank * scols, (c->rank + 1) * scols);
    }
    if (step == 0) {
        if ((rc = nmfx_mur_phase_a(E, distance, lw, j))) return rc;
        if (merged) { if ((rc = reduce_range(E, 0, 0, n32, E->stream))) return rc; }
        else {
            NMFX_RCCL(rccl().GroupStart());
            rc = reduce_range(E, 0, 0, n32, E->stream);
            if (!rc) rc = reduce_range(E, 1, 0, 8, E->stream);
            NMFX_RCCL(rccl().GroupEnd());
            if (rc) return rc;
        }
        return nmfx_mur_phase_b(E, distance, lh, min_iter, tol1, tol2, j);
    }
    // chunked: each column range of W^T V is reduced on the side stream while the next range is computed
    if ((rc = nmfx_mur_phase_a_head(E, distance, lw, j))) return rc;
    for (int64_t c0 = 0; c0 < npad;) {
        int64_t c1 = c0 + step;
        if (c1 > npad || npad - c1 < 512) c1 = npad;
        if ((rc = nmfx_mur_phase_a_cols(E, distance, c0, c1))) return rc;
        NMFX_HIP(hipEventRecord(c->ev_ready, E->stream));
        NMFX_HIP(hipStreamWaitEvent(c->side, c->ev_ready, 0));
        if ((rc = reduce_range(E, 0, c0 * kpad, (c1 < npad ? c1 * kpad : n32) - c0 * kpad, c->side))) return rc;
        c0 = c1;
    }
    if (!merged && (rc = reduce_range(E, 1, 0, 8, c->side))) return rc;
    NMFX_HIP(hipEventRecord(c->ev_done, c->side));
    NMFX_HIP(hipStreamWaitEvent(E->stream, c->ev_done, 0));
    return nmfx_mur_phase_b(E, distance, lh, min_iter, tol1, tol2, j);
}

// Capture iterations (0, 1) + "base += 2" into one graph (the launches carry indices relative to DevState::j_base).
static int capture_pair(nmfx_engine* E, int distance, double lw, double lh, int64_t min_iter, double tol1, double tol2, int chunks) {
    nmfx_comm* c = E->comm;
    drop_graph(c);
    NMFX_HIP(hipStreamSynchronize(E->stream));
    if (hipStreamBeginCapture(E->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return NMFX_E_HIP; }
    int rc = NMFX_OK;
    for (int64_t j = 0; j < 2 && !rc; ++j) rc = sharded_iteration(E, distance, lw, lh, min_iter, tol1, tol2, j, chunks);
    if (!rc) rc = nmfx_shift_iteration_base(E, 2);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(E->stream, &g);
    if (rc || e != hipSuccess || !g) { if (g) hipGraphDestroy(g); (void)hipGetLastError(); return rc ? rc : NMFX_E_HIP; }
    if (hipGraphInstantiate(&c->exec, g, nullptr, nullptr, 0) != hipSuccess) { hipGraphDestroy(g); c->exec = nullptr; (void)hipGetLastError(); return NMFX_E_HIP; }
    c->graph = g;
    c->g_hist = E->obj_hist; c->g_stream = E->stream;
    c->g_dist = distance; c->g_lw = lw; c->g_lh = lh; c->g_min = min_iter; c->g_t1 = tol1; c->g_t2 = tol2; c->g_chunks = chunks; c->g_precision = E->precision; c->g_exchange = c->exchange;
    return NMFX_OK;
}

// The reference's MUR loop body (nmf/mur.py:119-131) for a row shard: `count` outer iterations from index `first`, each
// phase A (W update of this rank's rows, this rank's [W^T V | W^T W | objective partial]) . RCCL sum-all-reduce . phase B
// (the replicated H update and the stop rule, identical on every rank) -- queued, no host synchronisation.
extern "C" int nmfx_mur_run_sharded(nmfx_handle_t E, int distance, double lambda_w, double lambda_h, int64_t min_iter, double tol1,
                                    double tol2, int64_t first, int64_t count) {
    int rc = have_comm(E); if (rc) return rc;
    if (distance != NMFX_EU && distance != NMFX_KL) { E->err = "Unknown distance type."; return NMFX_E_ARG; }
    if (first < 0 || count < 0) { E->err = "negative iteration range"; return NMFX_E_ARG; }
    NMFX_HIP(hipSetDevice(E->device));
    nmfx_comm* c = E->comm;
    if (!c->negotiated && (rc = nmfx_comm_negotiate(E))) return rc;
    const int chunks = chunks_from_env();
    int64_t j = first;
    const int64_t end = first + count;
    const bool graph_ok = c->want_graph && !c->graph_failed && !E->prof;
    if (graph_ok) {
        // the first pair runs eagerly (lazy allocations and the communicator's own start-up happen there)
        while (j < end && j < 2) { if ((rc = sharded_iteration(E, distance, lambda_w, lambda_h, min_iter, tol1, tol2, j, chunks))) return rc; ++j; }
        if (j < end && (j & 1)) { if ((rc = sharded_iteration(E, distance, lambda_w, lambda_h, min_iter, tol1, tol2, j, chunks))) return rc; ++j; }
        int64_t pairs = (end - j) / 2;
        // MUR-KL (split bf16): an iteration starts from the H images / row-sum partials its predecessor's epilogue left behind
        // (kl_h_iter == j - 1), or rebuilds them.  The captured pair must make the choice the eager loop makes at that point,
        // or the replays would rebuild in every other iteration (slower, and another summation order): capture iteration "0" as
        // the successor of iteration "-1" when that holds now, and bring a reused graph's assumption about by an eager pair.
        const bool klfresh = distance == NMFX_KL && E->kl_h_iter == j - 1;
        if (pairs > 0 && c->exec && c->g_klfresh && !klfresh) {
            for (int e = 0; e < 2; ++e, ++j) if ((rc = sharded_iteration(E, distance, lambda_w, lambda_h, min_iter, tol1, tol2, j, chunks))) return rc;
            pairs = (end - j) / 2;
        }
        if (pairs > 0) {
            if ((rc = nmfx_ensure_obj_capacity(E, end + 4))) return rc;           // (may move the history: checked below)
            const bool same = c->exec && c->g_dist == distance && c->g_lw == lambda_w && c->g_lh == lambda_h && c->g_min == min_iter &&
                              c->g_t1 == tol1 && c->g_t2 == tol2 && c->g_chunks == chunks && c->g_hist == E->obj_hist &&
                              c->g_stream == E->stream && c->g_precision == E->precision && c->g_exchange == c->exchange;
            if (!same) {
                const bool fresh_now = distance == NMFX_KL && E->kl_h_iter == j - 1;
                if (fresh_now) E->kl_h_iter = -1;
                if (capture_pair(E, distance, lambda_w, lambda_h, min_iter, tol1, tol2, chunks) != NMFX_OK) {
                    c->graph_failed = true;       // e.g. a collective that cannot be captured: the eager loop takes over for good
                    drop_graph(c);
                    E->kl_h_iter = -2;
                } else c->g_klfresh = fresh_now;
            }
            if (c->exec) {
                if ((rc = nmfx_shift_iteration_base(E, j))) return rc;
                for (int64_t p = 0; p < pairs; ++p) NMFX_HIP(hipGraphLaunch(c->exec, E->stream));
                c->replays += pairs;
                j += 2 * pairs;
                if ((rc = nmfx_shift_iteration_base(E, -j))) return rc;
                E->kl_h_iter = (distance == NMFX_KL && E->kl_h_iter >= 0) ? j - 1 : -2;      // (the capture left the index of ITS second iteration)
            }
        }
    }
    for (; j < end; ++j)
        if ((rc = sharded_iteration(E, distance, lambda_w, lambda_h, min_iter, tol1, tol2, j, chunks))) return rc;
    return NMFX_OK;
}

// objective of the last pair and the final stop-rule evaluation (nmf/mur.py:127-131 for i = max_iter - 1)
extern "C" int nmfx_mur_finish_sharded(nmfx_handle_t E, int distance, int64_t min_iter, double tol1, double tol2, int64_t iters_done) {
    int rc = have_comm(E); if (rc) return rc;
    NMFX_HIP(hipSetDevice(E->device));
    if ((rc = nmfx_mur_finish_a(E, distance, iters_done))) return rc;
    if ((rc = reduce_range(E, 1, 0, 8, E->stream))) return rc;
    return nmfx_mur_finish_b(E, min_iter, tol1, tol2, iters_done);
}

extern "C" int nmfx_comm_graph_replays(nmfx_handle_t E, int64_t* replays) {
    if (!E || !replays) return NMFX_E_ARG;
    *replays = E->comm ? E->comm->replays : 0;
    return NMFX_OK;
}
