// runtime.hip -- error strings, library-owned workspaces, per-kernel hipEvent timing.
#include <stdarg.h>
#include <mutex>
#include <vector>

#include "common.h"

namespace enerf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- workspaces -----------------------------------------------------------
// Grow-only scratch buffers, one set per device.  Growing a slot retires the old buffer (drain, free) and bumps a
// generation counter: anything that baked a workspace pointer into a captured HIP graph compares the counter before
// replaying and re-captures when it moved (enerf_workspace_generation; TrainHarness does).
static const int kMaxDevices = 16;
static std::mutex g_ws_mu;
static void* g_ws_ptr[kMaxDevices][WS_SLOTS] = {};
static size_t g_ws_bytes[kMaxDevices][WS_SLOTS] = {};
static uint64_t g_ws_generation = 0;

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}

void* workspace(int slot, size_t bytes) {
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (g_ws_bytes[dev][slot] >= bytes && g_ws_ptr[dev][slot]) return g_ws_ptr[dev][slot];
    size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        set_error("workspace: hipMalloc(%zu) failed", want);
        return nullptr;
    }
    // The old buffer may still be in use by kernels in flight on some stream: drain before freeing.
    if (g_ws_ptr[dev][slot]) {
        (void)hipDeviceSynchronize();
        (void)hipFree(g_ws_ptr[dev][slot]);
    }
    g_ws_ptr[dev][slot] = p;
    g_ws_bytes[dev][slot] = want;
    g_ws_generation++;
    return p;
}

uint64_t workspace_generation() {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    return g_ws_generation;
}

static bool g_ws_ordering = true;      // enerf_debug_workspace_ordering

int workspace_family_enter(int family, hipStream_t s) {
    constexpr int kFamilies = 2;
    if (!g_ws_ordering) return 0;
    static hipStream_t last[kMaxDevices][kFamilies] = {};
    static bool used[kMaxDevices][kFamilies] = {};
    static hipEvent_t ev[kMaxDevices][kFamilies] = {};
    if (family < 0 || family >= kFamilies) return 0;
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_ws_mu);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capturing);
    // (a capture stream is no "previous user": its work runs when the graph is replayed, not now -- recording an event on
    // it after the capture has ended would fail.  Graph replays are ordered against side-stream users by the harness:
    // TrainHarness never runs the side-stream march in graph mode.)
    if (capturing != hipStreamCaptureStatusNone) return 0;
    // (a stream under graph capture cannot wait for an event of a stream outside the capture; captures start from a
    // drained device -- torch.cuda.graph synchronises first -- so there is nothing in flight to wait for)
    if (used[dev][family] && last[dev][family] != s) {
        if (!ev[dev][family] && hipEventCreateWithFlags(&ev[dev][family], hipEventDisableTiming) != hipSuccess) {
            ev[dev][family] = nullptr;
            set_error("workspace_family_enter: hipEventCreate failed");
            return ENERF_E_NOMEM;
        }
        if (hipEventRecord(ev[dev][family], last[dev][family]) != hipSuccess ||
            hipStreamWaitEvent(s, ev[dev][family], 0) != hipSuccess) {
            set_error("workspace_family_enter: could not order stream after the family's previous user");
            return ENERF_E_NOMEM;
        }
    }
    used[dev][family] = true;
    last[dev][family] = s;
    return 0;
}

// The library's session state -- the table backward's pending record lists and overflow counters, the MLP kernels'
// valid-row pointer / reduce signal / deferred sums, the marcher's occupied box -- is per PROCESS: one process drives one
// GPU.  The first device that uses it owns it; a call from another device is refused instead of
// interleaving two models' sessions.
int single_device_guard(const char* what) {
    static int owner = -1;
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (owner < 0) owner = dev;
    if (owner != dev) {
        set_error("%s: this process drives device %d (one process per GPU); called with device %d current", what, owner, dev);
        return ENERF_E_UNSUPPORTED;
    }
    return 0;
}

uint32_t num_cus() {
    static uint32_t n[kMaxDevices] = {};
    const int dev = current_device();
    if (n[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n[dev] = (uint32_t)cus;
    }
    return n[dev];
}

// ---- event timing ---------------------------------------------------------
struct EvPair {
    hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static uint32_t g_prof_mask = 0;     // bit k: time kernel family k
static std::vector<EvPair> g_prof_ev[ENERF_K_COUNT];      // created once, reused after enerf_prof_reset
static size_t g_prof_used[ENERF_K_COUNT] = {};
static const size_t kMaxPairs = 1 << 16;
static uint32_t g_prof_every = 1;                        // enerf_prof_sample_every: time one call in g_prof_every
static uint64_t g_prof_calls[ENERF_K_COUNT] = {};        // eligible calls seen since the last reset
static double g_prof_units[ENERF_K_COUNT] = {};          // work units (ProfScope::units) of the TIMED calls

ProfScope::ProfScope(int kernel_id, hipStream_t stream, bool ext_) : id(kernel_id), s(stream), slot(nullptr), ext(ext_) {
    if (!((g_prof_mask >> id) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_calls[id]++ % g_prof_every != 0) return;
    if (g_prof_used[id] >= kMaxPairs) return;
    if (g_prof_used[id] == g_prof_ev[id].size()) {
        EvPair p;
        if (hipEventCreate(&p.a) != hipSuccess) return;
        if (hipEventCreate(&p.b) != hipSuccess) {
            (void)hipEventDestroy(p.a);
            return;
        }
        g_prof_ev[id].push_back(p);
    }
    if (!ext) (void)hipEventRecord(g_prof_ev[id][g_prof_used[id]].a, s);
    g_prof_used[id]++;
    slot = (void*)(uintptr_t)g_prof_used[id];  // 1-based index
}

hipEvent_t ProfScope::start() const {
    if (!slot) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return g_prof_ev[id][(size_t)(uintptr_t)slot - 1].a;
}
hipEvent_t ProfScope::stop() const {
    if (!slot) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return g_prof_ev[id][(size_t)(uintptr_t)slot - 1].b;
}

void ProfScope::units(double n) const {
    if (!slot) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_units[id] += n;
}

ProfScope::~ProfScope() {
    if (!slot || ext) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    size_t i = (size_t)(uintptr_t)slot - 1;
    if (i < g_prof_used[id]) (void)hipEventRecord(g_prof_ev[id][i].b, s);
}

}  // namespace enerf

extern "C" {

int enerf_debug_workspace_ordering(int on) {
    enerf::g_ws_ordering = on != 0;
    return 0;
}

const char* enerf_last_error(void) { return enerf::g_err; }
int enerf_abi_version(void) { return ENERF_ABI_VERSION; }      // include/enerf_hip.h
uint64_t enerf_workspace_generation(void) { return enerf::workspace_generation(); }

int enerf_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(enerf::g_prof_mu);
    enerf::g_prof_mask = on ? 0xffffffffu : 0u;
    return 0;
}

int enerf_prof_enable_mask(uint32_t mask) {
    std::lock_guard<std::mutex> lk(enerf::g_prof_mu);
    enerf::g_prof_mask = mask;
    return 0;
}

int enerf_prof_reset(void) {
    std::lock_guard<std::mutex> lk(enerf::g_prof_mu);
    for (int k = 0; k < ENERF_K_COUNT; k++) {
        enerf::g_prof_used[k] = 0;                      // the events themselves are kept for reuse
        enerf::g_prof_calls[k] = 0;
        enerf::g_prof_units[k] = 0.0;
    }
    return 0;
}

int enerf_prof_sample_every(uint32_t n) {
    std::lock_guard<std::mutex> lk(enerf::g_prof_mu);
    enerf::g_prof_every = n ? n : 1u;
    return 0;
}

int enerf_prof_read_units(int kernel_id, double* units, uint64_t* calls_seen) {
    if (kernel_id < 0 || kernel_id >= ENERF_K_COUNT) ENERF_BADARG("prof_read_units: bad kernel id %d", kernel_id);
    std::lock_guard<std::mutex> lk(enerf::g_prof_mu);
    if (units) *units = enerf::g_prof_units[kernel_id];
    if (calls_seen) *calls_seen = enerf::g_prof_calls[kernel_id];
    return 0;
}

int enerf_prof_read(int kernel_id, double* total_ms, uint64_t* launches) {
    if (kernel_id < 0 || kernel_id >= ENERF_K_COUNT) ENERF_BADARG("prof_read: bad kernel id %d", kernel_id);
    std::lock_guard<std::mutex> lk(enerf::g_prof_mu);
    double tot = 0;
    uint64_t n = 0;
    for (size_t i = 0; i < enerf::g_prof_used[kernel_id]; i++) {
        auto& p = enerf::g_prof_ev[kernel_id][i];
        if (hipEventSynchronize(p.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            tot += ms;
            n++;
        }
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return 0;
}

}  // extern "C"
