// common.h -- shared host/device helpers for libenerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/enerf_hip.h"

namespace enerf {

constexpr int kWave = 64;  // CDNA wavefront

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define ENERF_BADARG(...)            \
    do {                             \
        enerf::set_error(__VA_ARGS__); \
        return ENERF_E_BADARG;       \
    } while (0)

#define ENERF_LAUNCH_CHECK(name)                                   \
    do {                                                           \
        int _e = enerf::check_hip(hipGetLastError(), name);        \
        if (_e) return _e;                                         \
    } while (0)

// ---- per-kernel-family event timing (include/enerf_hip.h: enerf_prof_*) ---
struct ProfScope {
    int id;
    hipStream_t s;
    void* slot;
    bool ext;
    // ext = true: the caller launches ONE kernel with hipExtLaunchKernelGGL(..., start(), stop(), ...), which stamps the
    // events with the kernel's own begin / end (what rocprofv3 reports) instead of the time between two event packets
    ProfScope(int kernel_id, hipStream_t stream, bool ext = false);
    ~ProfScope();
    hipEvent_t start() const;      // nullptr when this launch is not being timed
    hipEvent_t stop() const;
    void units(double n) const;    // work units of this call (points, samples): summed over the TIMED calls only
};

// ---- loss scaling of the fp16 regime (include/enerf_hip.h: enerf_amp_begin / enerf_amp_end, csrc/optim.hip) ----------
// Between the two calls: the closed-form loss gradient is multiplied by *scale (compositing / event-loss kernels), the MLP
// weight-gradient reduce launch raises *found_inf when a sum is not finite, and the table / MLP Adam launch divides the
// gradients by *scale, counts its step as (host step - *skipped) and leaves p / m / v alone when *found_inf is set.
struct AmpState {
    const float* scale;        // nullptr: loss scaling is off
    uint32_t* found_inf;
    const uint32_t* skipped;
};
AmpState amp_state();

// ---- workspace owned by the library (grow-only, per process) --------------
// Returns a device buffer of at least `bytes`; nullptr on failure.  Slots are independent.
void* workspace(int slot, size_t bytes);
uint64_t workspace_generation();    // bumped whenever a slot is re-allocated (its old pointer dies)
uint32_t num_cus();                    // compute units of the current device (256 when it cannot be asked)
// The workspaces of one kernel family are shared by every stream of the device.  An entry point that is about to use
// them on stream `s` calls this first: if the family's previous user was another stream, `s` is made to wait for
// everything queued on that stream so far (an event recorded there now).  Costs nothing while one stream keeps the
// family to itself.  `family`: 0 = the marcher (chunk log, scan tiles, compaction counters, occupied box).
int workspace_family_enter(int family, hipStream_t s);
// process-wide session state belongs to the first device that used it: != 0 (error set) when another device is current
int single_device_guard(const char* what);
enum { WS_SCAN = 0, WS_COMPACT = 1, WS_FFMLP = 2, WS_GRIDBWD = 3, WS_MARCH = 4, WS_DENSITY = 5, WS_MLP32_DEFER = 6, WS_AABB = 7, WS_AABB_CALL = 8, WS_FFMLP_W = 9, WS_NERF_FRAGS = 10, WS_NERF_PART = 11, WS_MARCH2 = 12, WS_SLOTS = 13 };

__host__ __device__ inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---- a small job that another kernel's launch carries in a few extra workgroups ------------------------------------
// Gather-and-split: thread t makes values 8t .. 8t+7, value i = src[map[i] >> 16][map[i] & 0xffff] (map 0xffffffff: 0.0f),
// as bf16 hi (round to nearest even) and lo = bf16(value - hi); the 16 bytes of hi go to out + (t / 64) * 512 + (t % 64) * 4
// (in 32-bit words), the 16 bytes of lo 256 words further.  (What csrc/nerf_mlp.hip's operand fragments are: the training
// step's grid forward builds them beside its own work instead of a 5 us launch in front of the MLP forward.)
struct SplitJob {
    const float* src[5];
    const uint32_t* map;
    uint32_t* out;
    uint32_t threads;
};
// gridencoder.hip: the next fp32 D = 3, C = 2 forward launch carries `job` (one-shot; nullptr disarms).  Returns whether a
// job armed earlier was still waiting (= no launch took it).
bool grid_fwd_carry(const SplitJob* job);
// gridencoder.hip: while set (device_count != nullptr), enerf_grid_encode_forward / _backward(_ex) treat their B rows as a
// budget of which only base + min(*device_count, cap) (cap == 0: *device_count), rounded up to 32, are real -- the
// convention of enerf_mlp32_valid_rows(_ex), whose kernels sit between the two and skip the same rows.  Results for real
// rows are unchanged; the rest is neither encoded nor binned.  Set and cleared by the whole-step entry points only.
void grid_valid_rows(const int32_t* device_count, uint32_t base, uint32_t cap);
// mlp32.hip: the job that builds the fragments enerf_nerf_mlp_forward / _backward would build for these weights, and the
// promise that it runs before them on `s`: the calls that follow with flags bit 0 use the fragments as they are.
int nerf_mlp_frag_job(const float* const* wseg_s, const float* const* wseg_c, uint32_t w0_cols_c, uint32_t out_c,
                      hipStream_t s, SplitJob* job);
// ... and the promise withdrawn: no launch took the job (or the launch that should have failed), the book-keeping of
// "fragments current for these weight pointers" is void and the next MLP call rebuilds them whatever its flags say.
void nerf_mlp_frags_invalidate();
// Per-workgroup partial sums that the table optimizer's launch reduces on the fly: value i (< n) = sum over b < parts of
// partial[b * stride + i], the gradient of element map[i] & 0xffffff of that launch's small tensor map[i] >> 24
// (0xffffffff: of nobody).  (The fused MLP backward's weight gradients: k_mlp32_reduce_w2 and its 5 us leave the chain.)
struct PartialSums {
    const float* partial;
    const uint32_t* map;
    uint32_t parts, stride, n;
};
// gridencoder.hip: the next enerf_grid_adam_from_records(_ex) call sums `job` for its small tensors' gradients (and
// stores them where it would have read them) -- one-shot; nullptr disarms.  Returns whether a job was still waiting.
bool grid_adam_partial_sums(const PartialSums* job);
// mlp32.hip: the job for the enerf_nerf_mlp_backward call that follows on `s` with flags bit 1 (B rows, same gradient
// segments); small_g / small_n: the optimizer call's small tensors.  0: `job` is set; 1: does not apply (the small tensors
// are not exactly the five gradient matrices, or loss scaling is armed); < 0: error.
int nerf_mlp_partial_job(float* const* dwseg_s, float* const* dwseg_c, uint32_t w0_cols_c, uint32_t out_c,
                         const float* const* small_g, const uint32_t* small_n, uint32_t n_small, uint32_t B, hipStream_t s,
                         PartialSums* job);

// The count pass of the wave-per-ray lattice marcher as data (csrc/march_lattice.h: march_count_block runs it as workgroup
// `bid` of `blocks` workgroups of 256 threads).  In the one-call training step the NEXT batch's count pass rides in the
// table optimizer's launch -- one queue, no second stream, none of the two cross-stream hand-overs (a signal behind the MLP
// backward, an event wait at the head of the next step: ~17 us of idle queue per step) -- and one small launch behind the
// optimizer scans the counts and writes the samples.
struct MarchCountJob {
    const float* rays_o;
    const float* rays_d;
    const uint8_t* grid;
    float bound;
    uint32_t max_steps, N, C, H;
    const float* nears;          // (nf_nears / nf_fars alias them when near / far are computed in the count pass)
    const float* fars;
    int32_t* rays;
    uint32_t perturb;
    void* log;                   // ChunkEntry[N][kLogCap]
    uint32_t* nlog;
    const int* occ_keys;
    const float* nf_aabb;
    float nf_min_near;
    float* nf_nears;
    float* nf_fars;
    uint32_t blocks;
};
// gridencoder.hip: the next enerf_grid_adam_from_records(_ex) launch of the plain fp32 C = 2 form carries `job` in
// job->blocks extra workgroups.  Up to TWO jobs may wait (the event step's two renders): each call with a job appends;
// nullptr disarms.  Returns whether a job armed earlier was still waiting.
bool tile_adam_carry_count(const MarchCountJob* job);
// raymarching.hip: march_rays_train_ex(...) split around a carrying launch.  begin: 0 = *job is the call's count pass (the
// workspace is prepared, the one-shot near / far request consumed) and the call's scan + write are remembered for
// march_carry_end; 1 = this call cannot be served that way (another marcher, a count mirror armed, a kept counter ...):
// nothing consumed, make the ordinary call; < 0 = error.  end: scan + write on `s` (behind the launch that carried the job).
// Up to two marches may be pending (begun one after the other: the second uses a chunk log of its own, WS_MARCH2); end
// finishes all of them in the order they were begun.  `share`: how many marches will ride in the launch (splits the workgroups).
int march_carry_begin(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                      uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                      const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                      uint32_t perturb, uint32_t flags, hipStream_t s, MarchCountJob* job, uint32_t share = 1);
int march_carry_end(hipStream_t s);
int march_carry_count_now(const MarchCountJob* job, hipStream_t s);     // (the carrying launch did not take the job)
void march_carry_abort();                                               // (a step failed between begin and end)

// ---- wave-level primitives (wave64) ----------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Inclusive scans of a wavefront with DPP lane exchanges (no LDS crossbar, no per-step select): Hillis-Steele steps
// 1, 2, 4, 8 inside each 16-lane row (row_shr; a lane without a source keeps the operation's identity), then the row
// totals: lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15, row mask 0xA), lane 31 into rows 2 and 3 (row_bcast31,
// row mask 0xC).  All 64 lanes must be active.
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int dpp_take(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWS, 0xf, false);
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ float dpp_take(float identity, float v) {
    return __int_as_float(dpp_take<CTRL, ROWS>(__float_as_int(identity), __float_as_int(v)));
}
__device__ __forceinline__ float wave_incl_scan_add(float v, int) {
    v += dpp_take<0x111>(0.0f, v);
    v += dpp_take<0x112>(0.0f, v);
    v += dpp_take<0x114>(0.0f, v);
    v += dpp_take<0x118>(0.0f, v);
    v += dpp_take<0x142, 0xa>(0.0f, v);
    v += dpp_take<0x143, 0xc>(0.0f, v);
    return v;
}
__device__ __forceinline__ float wave_incl_scan_mul(float v, int) {
    v *= dpp_take<0x111>(1.0f, v);
    v *= dpp_take<0x112>(1.0f, v);
    v *= dpp_take<0x114>(1.0f, v);
    v *= dpp_take<0x118>(1.0f, v);
    v *= dpp_take<0x142, 0xa>(1.0f, v);
    v *= dpp_take<0x143, 0xc>(1.0f, v);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_add_u32(uint32_t x, int) {
    int v = (int)x;
    v += dpp_take<0x111>(0, v);
    v += dpp_take<0x112>(0, v);
    v += dpp_take<0x114>(0, v);
    v += dpp_take<0x118>(0, v);
    v += dpp_take<0x142, 0xa>(0, v);
    v += dpp_take<0x143, 0xc>(0, v);
    return (uint32_t)v;
}
// value of the lane below (wave_shr:1); lane 0 gets `first`
__device__ __forceinline__ float wave_prev(float v, float first) { return dpp_take<0x138>(first, v); }
// a lane's value in every lane (v_readlane: `src` is wave-uniform)
__device__ __forceinline__ float wave_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

}  // namespace enerf
