// dp_tail.hip -- the data-parallel tail of the training step behind two C calls.
//
// Rays shard over the GPUs of a node; the only exchange of a step is the average of the hash-table gradient (52 MB at
// bound 3) and of the MLP weight gradients (37 KB) over RCCL / xGMI before Adam.  Driven from Python through
// torch.distributed that tail is a dozen dispatcher round trips per step (an async all-reduce per piece of the table,
// a wait and an optimizer call per piece, ...): 0.2 - 0.4 ms of host time beside a 0.35 ms device step, i.e. the host
// becomes the limit as soon as there is more than one rank.  Here the whole tail is enqueued by
//
//   enerf_dp_begin   the collectives, on this library's own communicator and stream (ordered after everything queued on
//                    the training stream so far):  mode 0: all-reduce (AVG) of the table gradient in `pieces` pieces;
//                    mode 1: reduce-scatter (AVG) of it, this rank keeping its slice; then the all-reduce of the flat MLP
//                    gradient buffer.  Returns at once: the caller issues whatever should run UNDER the collectives (the
//                    next batch's march) between the two calls.
//   enerf_dp_finish  the consumers, on the training stream: mode 0: Adam on each piece of the table as it lands (the
//                    optimizer pass hides under the remaining collectives), gradients cleared by the same kernel;
//                    mode 1: Adam on this rank's slice only, the rest of the gradient buffer cleared, and the all-gather of
//                    the updated slices into every replica's table (collective stream again), which the training stream
//                    waits for.  Either way it ends with the wait for the MLP gradients' all-reduce.
//
// The communicator is RCCL's, through the copy of librccl the process already holds (torch's: resolved at run time, no
// second runtime is loaded); its unique id is minted on rank 0 (enerf_dp_unique_id) and handed round by the caller
// (torch.distributed broadcast).  One communicator per process (one process per GPU).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>

#include <cstring>

#include "common.h"

using namespace enerf;

namespace {

// the few RCCL entry points used, with the types of rccl.h spelled out (ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7,
// ncclSum = 0, ncclAvg = 4, ncclSuccess = 0)
struct UniqueId {
    char internal[128];
};
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*ReduceScatterFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat32 = 7, kAvg = 4, kSum = 0;
constexpr uint32_t kMaxPieces = 16;

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    ReduceScatterFn reduce_scatter = nullptr;
    AllGatherFn all_gather = nullptr;
    GetErrorStringFn error_string = nullptr;
};
std::mutex g_mu;
Rccl g_rccl;
Comm g_comm = nullptr;
int g_rank = 0, g_world = 1;
hipStream_t g_cs = nullptr;                      // the collectives' stream
hipEvent_t g_ev_main = nullptr, g_ev_piece[kMaxPieces] = {}, g_ev_dw = nullptr, g_ev_gather = nullptr;
struct Pending {                                // what enerf_dp_begin queued and enerf_dp_finish consumes
    bool open = false;
    int mode = 0;
    uint32_t pieces = 0;
    size_t n = 0, lo[kMaxPieces] = {}, hi[kMaxPieces] = {};
    float* g = nullptr;
} g_pending;

int load_rccl() {
    if (g_rccl.handle) return 0;
    // the copy already in the process first (torch links "librccl.so"; its soname is librccl.so.1): never a second runtime
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        set_error("dp: librccl not found (%s)", dlerror());
        return ENERF_E_BADARG;
    }
    Rccl r;
    r.handle = h;
    r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.reduce_scatter = (ReduceScatterFn)dlsym(h, "ncclReduceScatter");
    r.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
    r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.reduce_scatter || !r.all_gather) {
        set_error("dp: librccl lacks an entry point");
        return ENERF_E_BADARG;
    }
    g_rccl = r;
    return 0;
}

int check_rccl(int rc, const char* what) {
    if (rc == 0) return 0;
    set_error("dp: %s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "rccl error");
    return ENERF_E_UNSUPPORTED;
}

// [lo, hi) of every piece: equal pieces rounded up to multiples of 4 elements (the Adam kernel's vector width)
void cut(size_t n, uint32_t pieces, size_t* lo, size_t* hi, uint32_t& count) {
    size_t step = (n + pieces - 1) / pieces;
    step += (4 - step % 4) % 4;
    count = 0;
    for (size_t a = 0; a < n && count < kMaxPieces; a += step) {
        lo[count] = a;
        hi[count] = a + step < n ? a + step : n;
        count++;
    }
}

__global__ void __launch_bounds__(256) k_clear(float4* __restrict__ p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
void clear_async(float* p, size_t n, hipStream_t s) {        // n and p are multiples of 4 elements / 16 bytes here
    if (n == 0) return;
    k_clear<<<512, 256, 0, s>>>(reinterpret_cast<float4*>(p), n / 4);
}

void release_stream_and_events() {
    if (g_cs) {
        (void)hipStreamDestroy(g_cs);
        g_cs = nullptr;
    }
    hipEvent_t* single[] = {&g_ev_main, &g_ev_dw, &g_ev_gather};
    for (hipEvent_t* e : single)
        if (*e) {
            (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    for (uint32_t k = 0; k < kMaxPieces; k++)
        if (g_ev_piece[k]) {
            (void)hipEventDestroy(g_ev_piece[k]);
            g_ev_piece[k] = nullptr;
        }
}

}  // namespace

extern "C" {

int enerf_dp_unique_id(void* out, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out || bytes < sizeof(UniqueId)) ENERF_BADARG("dp_unique_id: need a buffer of %zu bytes", sizeof(UniqueId));
    if (int e = load_rccl()) return e;
    return check_rccl(g_rccl.get_unique_id((UniqueId*)out), "ncclGetUniqueId");
}

// 0 when librccl can be reached (what enerf_dp_unique_id does first; ranks other than 0 call this instead of minting an id)
int enerf_dp_probe(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return load_rccl();
}

int enerf_dp_init(const void* unique_id, size_t bytes, int rank, int world) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!unique_id || bytes < sizeof(UniqueId) || world < 1 || rank < 0 || rank >= world)
        ENERF_BADARG("dp_init: bad arguments (rank %d of %d)", rank, world);
    if (g_comm) ENERF_BADARG("dp_init: already initialised (enerf_dp_shutdown first)");
    if (int e = load_rccl()) return e;
    UniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    if (int e = check_hip(hipStreamCreateWithFlags(&g_cs, hipStreamNonBlocking), "dp_init(stream)")) return e;
    auto ev = [](hipEvent_t* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming); };
    hipError_t he = ev(&g_ev_main);
    if (he == hipSuccess) he = ev(&g_ev_dw);
    if (he == hipSuccess) he = ev(&g_ev_gather);
    for (uint32_t k = 0; k < kMaxPieces && he == hipSuccess; k++) he = ev(&g_ev_piece[k]);
    if (int e = check_hip(he, "dp_init(events)")) {
        release_stream_and_events();
        return e;
    }
    if (int e = check_rccl(g_rccl.comm_init_rank(&g_comm, world, id, rank), "ncclCommInitRank")) {
        g_comm = nullptr;
        release_stream_and_events();
        return e;
    }
    g_rank = rank;
    g_world = world;
    g_pending = Pending();
    return 0;
}

int enerf_dp_world(int* rank, int* world) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (rank) *rank = g_comm ? g_rank : -1;
    if (world) *world = g_comm ? g_world : 0;
    return 0;
}

int enerf_dp_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm) {
        (void)hipStreamSynchronize(g_cs);
        (void)g_rccl.comm_destroy(g_comm);
        g_comm = nullptr;
    }
    release_stream_and_events();
    g_pending = Pending();
    return 0;
}

int enerf_dp_begin(int mode, float* table_grad, size_t n, uint32_t pieces, float* mlp_grad, size_t n_mlp,
                   enerf_stream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) ENERF_BADARG("dp_begin: enerf_dp_init has not run");
    if (g_pending.open) ENERF_BADARG("dp_begin: the previous step's enerf_dp_finish has not run");
    // mode 2: mode 1 with a SUM reduce-scatter (the caller's optimizer pass applies 1 / ranks: enerf_grid_owner_range)
    if (!table_grad || n == 0 || n % 4 || mode < 0 || mode > 2) ENERF_BADARG("dp_begin: bad arguments");
    if (mode == 0 && (pieces < 1 || pieces > kMaxPieces)) ENERF_BADARG("dp_begin: 1..%u pieces", kMaxPieces);
    if (mode >= 1 && (n % (size_t)g_world || (n / (size_t)g_world) % 4))
        ENERF_BADARG("dp_begin: the sharded tail needs the table to divide over the ranks in multiples of 4 elements");
    hipStream_t s = (hipStream_t)stream;
    Pending pd;
    pd.mode = mode == 2 ? 1 : mode;
    pd.n = n;
    pd.g = table_grad;
    // the collectives come after everything the training stream holds so far (the backward that filled the buffers)
    if (int e = check_hip(hipEventRecord(g_ev_main, s), "dp_begin(record)")) return e;
    if (int e = check_hip(hipStreamWaitEvent(g_cs, g_ev_main, 0), "dp_begin(wait)")) return e;
    if (mode == 0) {
        cut(n, pieces, pd.lo, pd.hi, pd.pieces);
        for (uint32_t k = 0; k < pd.pieces; k++) {
            float* p = table_grad + pd.lo[k];
            if (int e = check_rccl(g_rccl.all_reduce(p, p, pd.hi[k] - pd.lo[k], kFloat32, kAvg, g_comm, g_cs), "ncclAllReduce"))
                return e;
            if (int e = check_hip(hipEventRecord(g_ev_piece[k], g_cs), "dp_begin(piece event)")) return e;
        }
    } else {
        const size_t shard = n / (size_t)g_world;
        pd.pieces = 1;
        pd.lo[0] = shard * (size_t)g_rank;
        pd.hi[0] = pd.lo[0] + shard;
        // in place: this rank's slice of the buffer receives the average of everybody's slice
        if (int e = check_rccl(g_rccl.reduce_scatter(table_grad, table_grad + pd.lo[0], shard, kFloat32, mode == 2 ? kSum : kAvg,
                                                     g_comm, g_cs),
                               "ncclReduceScatter"))
            return e;
        if (int e = check_hip(hipEventRecord(g_ev_piece[0], g_cs), "dp_begin(slice event)")) return e;
    }
    if (mlp_grad && n_mlp) {
        if (int e = check_rccl(g_rccl.all_reduce(mlp_grad, mlp_grad, n_mlp, kFloat32, kAvg, g_comm, g_cs), "ncclAllReduce(mlp)"))
            return e;
    }
    if (int e = check_hip(hipEventRecord(g_ev_dw, g_cs), "dp_begin(mlp event)")) return e;
    pd.open = true;
    g_pending = pd;
    return 0;
}

int enerf_dp_finish(float* p, float* m, float* v, float lr, float beta1, float beta2, float eps, uint32_t step,
                    enerf_stream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm || !g_pending.open) ENERF_BADARG("dp_finish: no enerf_dp_begin is pending");
    if (!p || !m || !v || step == 0) ENERF_BADARG("dp_finish: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const Pending pd = g_pending;
    g_pending.open = false;
    for (uint32_t k = 0; k < pd.pieces; k++) {
        if (int e = check_hip(hipStreamWaitEvent(s, g_ev_piece[k], 0), "dp_finish(wait piece)")) return e;
        float* pp = p + pd.lo[k];
        float* gg = pd.g + pd.lo[k];
        float* mm = m + pd.lo[k];
        float* vv = v + pd.lo[k];
        const size_t cnt = pd.hi[k] - pd.lo[k];
        // Adam on the piece; the kernel clears the gradients it has read (the next step's backward adds into the buffer)
        if (int e = enerf_adam_step_multi(1, &pp, &gg, &mm, &vv, &cnt, &lr, &step, beta1, beta2, eps, 1, stream)) return e;
    }
    if (pd.mode == 1) {
        // this rank's contributions to the other slices are spent
        clear_async(pd.g, pd.lo[0], s);
        clear_async(pd.g + pd.hi[0], pd.n - pd.hi[0], s);
        // updated slices -> every replica's table (in place), on the collectives' stream behind the Adam launch
        if (int e = check_hip(hipEventRecord(g_ev_main, s), "dp_finish(record)")) return e;
        if (int e = check_hip(hipStreamWaitEvent(g_cs, g_ev_main, 0), "dp_finish(wait)")) return e;
        if (int e = check_rccl(g_rccl.all_gather(p + pd.lo[0], p, pd.hi[0] - pd.lo[0], kFloat32, g_comm, g_cs), "ncclAllGather"))
            return e;
        if (int e = check_hip(hipEventRecord(g_ev_gather, g_cs), "dp_finish(gather event)")) return e;
        if (int e = check_hip(hipStreamWaitEvent(s, g_ev_gather, 0), "dp_finish(wait gather)")) return e;
    }
    if (int e = check_hip(hipStreamWaitEvent(s, g_ev_dw, 0), "dp_finish(wait mlp)")) return e;
    ENERF_LAUNCH_CHECK("dp_finish");
    return 0;
}

// The sharded tail with the optimizer pass left to the caller (enerf_grid_owner_range + enerf_grid_adam_from_records_ex:
// this rank's slice keeps its record lists): after enerf_dp_begin(mode 1, ...)
//   enerf_dp_wait       `stream` waits for the reduce-scatter and for the MLP gradients' all-reduce -- the caller's
//                       optimizer launch goes behind it;
//   enerf_dp_allgather  the updated slices -> every replica's table (in place, on the collectives' stream behind what
//                       `stream` holds so far), which `stream` then waits for; closes the step enerf_dp_begin opened.
int enerf_dp_wait(enerf_stream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm || !g_pending.open) ENERF_BADARG("dp_wait: no enerf_dp_begin is pending");
    hipStream_t s = (hipStream_t)stream;
    for (uint32_t k = 0; k < g_pending.pieces; k++)
        if (int e = check_hip(hipStreamWaitEvent(s, g_ev_piece[k], 0), "dp_wait(piece)")) return e;
    return check_hip(hipStreamWaitEvent(s, g_ev_dw, 0), "dp_wait(mlp)");
}

int enerf_dp_allgather(float* p, enerf_stream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm || !g_pending.open) ENERF_BADARG("dp_allgather: no enerf_dp_begin is pending");
    if (!p || g_pending.mode != 1) ENERF_BADARG("dp_allgather: needs the table and a sharded (mode 1) enerf_dp_begin");
    hipStream_t s = (hipStream_t)stream;
    const Pending pd = g_pending;
    g_pending.open = false;
    if (int e = check_hip(hipEventRecord(g_ev_main, s), "dp_allgather(record)")) return e;
    if (int e = check_hip(hipStreamWaitEvent(g_cs, g_ev_main, 0), "dp_allgather(wait)")) return e;
    if (int e = check_rccl(g_rccl.all_gather(p + pd.lo[0], p, pd.hi[0] - pd.lo[0], kFloat32, g_comm, g_cs), "ncclAllGather"))
        return e;
    if (int e = check_hip(hipEventRecord(g_ev_gather, g_cs), "dp_allgather(event)")) return e;
    return check_hip(hipStreamWaitEvent(s, g_ev_gather, 0), "dp_allgather(wait gather)");
}

}  // extern "C"
