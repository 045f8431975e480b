// train_step.hip -- one closed-form training step as ONE call (host code only: every launch is one of the library's
// own entry points, in the order the Python harness issues them -- enerf_amd/fused_render.train_step_mse +
// fused_network.nerf_forward / nerf_backward + FusedAdam.step_grid_table -- so the two routes are bit-identical).
//
// Why: with the kernels of a 4096-ray step at ~0.32 ms, the ~28 launches of a step cost the Python harness ~0.4 ms of
// host time (argument marshalling, tensor checks, dispatcher calls between the launches): the host, not the device,
// bounds the step -- and under data parallelism it has the collectives to issue as well.  Here the host's share of a
// step is one struct and one call; what remains on its side is the launches themselves.
//
//   render of batch i (its samples were marched by the previous call, on the side stream):
//     grid_encode_forward -> mlp32 forward (sigma net, + SH columns) -> mlp32 forward (colour net)
//     -> composite forward + MSE + composite backward (one launch)
//     -> mlp32 backward (colour net) -> mlp32 backward (sigma net; its reduce launch carries the signal the side stream
//        waits for) -> [side stream: near_far + march_rays_train of batch i+1] -> grid_encode_backward (record lists)
//     -> table Adam from the records + the MLP weights' Adam (one launch)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>

#include "common.h"

using namespace enerf;

// development aid (enerf_debug_step_timing): host microseconds spent in each call of the step, summed over steps
static double g_host_us[16];
static uint64_t g_host_steps = 0;
static bool g_host_timing = false;
// the fused MLP's operand fragments built inside the grid forward's launch (ENERF_NO_CARRY_FRAGS: by their own launch)
static bool g_carry_frags = getenv("ENERF_NO_CARRY_FRAGS") == nullptr;      // enerf_debug_carry_frags
// the fused MLP's weight-gradient partial sums summed by the optimizer's launch (ENERF_NO_FOLD_REDUCE: by k_mlp32_reduce_w2)
static bool g_fold_reduce = getenv("ENERF_NO_FOLD_REDUCE") == nullptr;      // enerf_debug_fold_reduce
// the next batch's march without a second stream: its count pass rides in the table optimizer's launch, scan + write are one
// launch behind it (common.h MarchCountJob; ENERF_NO_CARRY_COUNT / enerf_debug_carry_count(0): the side-stream march)
static bool g_carry_count = getenv("ENERF_NO_CARRY_COUNT") == nullptr;
static int g_carried_steps = 0;          // steps whose next batch was marched that way so far (enerf_debug_carry_count(-2))
extern "C" int enerf_debug_carry_count(int on) {
    if (on == -2) return g_carried_steps;
    const int prev = g_carry_count ? 1 : 0;
    if (on >= 0) g_carry_count = on != 0;
    return prev;
}

extern "C" int enerf_train_step_mse(const enerf_train_step_args* a) {
    if (!a) ENERF_BADARG("train_step_mse: null arguments");
    if (a->struct_bytes != sizeof(enerf_train_step_args))
        ENERF_BADARG("train_step_mse: struct of %u bytes, this library expects %zu", a->struct_bytes,
                     sizeof(enerf_train_step_args));
    if (a->M == 0 || a->N == 0) return 0;
    enerf_stream_t s = a->stream;
    const uint32_t M = a->M, N = a->N;
    const float in_add = a->bound, in_mul = a->inv_two_bound;
    int prev_prec = -1;
    if (a->mlp_precision >= 0) prev_prec = enerf_mlp32_precision(a->mlp_precision);
    int rc = 0;
    bool rows_set = false, defer_set = false, signal_set = false, fused_mlp = false, carry_set = false, own_sums = false;
    bool count_carried = false;
    MarchCountJob count_job{};
    uint32_t frags_built = 0;
    PartialSums sums{nullptr, nullptr, 0, 0, 0};
    int slot = 0;
    auto t_prev = std::chrono::steady_clock::now();
    if (g_host_timing) g_host_steps++;
#define STEP(call)                                                                       \
    do {                                                                                 \
        rc = (call);                                                                     \
        if (g_host_timing) {                                                             \
            const auto t_now = std::chrono::steady_clock::now();                         \
            g_host_us[slot < 15 ? slot : 15] += std::chrono::duration<double, std::micro>(t_now - t_prev).count(); \
            t_prev = t_now;                                                              \
            slot++;                                                                      \
        }                                                                                \
        if (rc) goto done;                                                               \
    } while (0)
    // ---- forward
    // (both nets as one launch each way when the arithmetic is the split-bf16 default: csrc/nerf_mlp.hip; their operand
    //  fragments are then built by sixteen extra workgroups of the grid forward's launch -- common.h SplitJob)
    fused_mlp = a->nh_s == 1 && a->nh_c == 2 && enerf_nerf_mlp_available() != 0;
    if (fused_mlp && g_carry_frags) {
        SplitJob job;
        STEP(nerf_mlp_frag_job(a->wseg_s, a->wseg_c, a->w0_cols_c, a->out_c, (hipStream_t)s, &job));
        if (g_host_timing) slot--;
        grid_fwd_carry(&job);
        carry_set = true;
    }
    if (a->counter) grid_valid_rows(a->counter, 0, 0);      // (forward .. backward: the budget's unfilled rows are skipped)
    STEP(enerf_grid_encode_forward(a->xyzs, a->embeddings, a->offsets, a->feats, M, 3, 2, 16, a->level_scale_log2,
                                   a->base_resolution, 0, a->feats, a->gridtype, ENERF_F32, 2, in_add, in_mul, s));
    if (carry_set) {
        frags_built = grid_fwd_carry(nullptr) ? 0u : 1u;      // (taken along: the MLP calls are told so)
        if (!frags_built) nerf_mlp_frags_invalidate();
        carry_set = false;
    }
    if (a->counter) {
        enerf_mlp32_valid_rows(a->counter);
        rows_set = true;
    }
    if (fused_mlp) {
        STEP(enerf_nerf_mlp_forward(a->feats, a->dirs, a->wseg_s, a->wseg_c, a->w0_cols_c, M, a->out_c, a->sigma, a->rgb,
                                    frags_built, s));
        if (g_host_timing) slot++;
    } else {
        STEP(enerf_mlp32_forward_p(a->feats, a->wseg_s, 32, 0, M, 32, 16, a->nh_s, 0, 6, a->fb_s, a->h32, 1, 32, a->sigma,
                                   a->dirs, s));
        STEP(enerf_mlp32_forward_p(a->h32, a->wseg_c, a->w0_cols_c, 1, M, 32, a->out_c, a->nh_c, 0, 3, a->fb_c, a->rgb, 0, 0,
                                   nullptr, nullptr, s));
    }
    // ---- compositing forward + loss gradient + compositing backward
    STEP(enerf_composite_rays_train_fwd_bwd_mse(a->sigma, a->rgb, a->deltas, a->rays, M, N, a->weights_sum, a->image,
                                                nullptr, 0, a->bg_scalar, a->out_image, a->target, a->grad_scale,
                                                a->counter, a->g_sigmas, a->g_rgbs, a->loss, s));
    // ---- the next batch's march: carried by the optimizer's launch where that applies (decided here, because the
    //      side-stream form needs its signal armed on the MLP backward)
    if (a->next_rays_o && g_carry_count && !(a->flags & 1u) && !(a->march_flags & 16u)) {
        enerf_march_fuse_near_far(a->aabb, a->min_near);           // (near / far inside the count pass)
        const int b = march_carry_begin(a->next_rays_o, a->next_rays_d, a->bitfield, a->bound, a->dt_gamma, a->max_steps,
                                        a->next_N, a->cascade, a->grid_size, a->next_M, a->next_nears, a->next_fars,
                                        a->next_xyzs, a->next_dirs, a->next_deltas, a->next_rays, a->next_counter,
                                        a->perturb, a->march_flags, (hipStream_t)s, &count_job);
        if (b < 0) { rc = b; goto done; }
        count_carried = b == 0;
        if (!count_carried) enerf_march_fuse_near_far(nullptr, 0.0f);    // (nothing consumed: the ordinary call arms it again)
    }
    // ---- MLP backward (the colour net's partial sums wait for the sigma net's reduce launch)
    if (a->next_rays_o && !count_carried) {
        enerf_mlp32_signal_next_reduce(1);
        signal_set = true;
    }
    if (fused_mlp) {
        // (the weight gradients' partial sums are summed by the optimizer's launch when it follows in this call and its small
        //  tensors are those gradients: no reduce launch in between -- common.h PartialSums)
        if (g_fold_reduce && !(a->flags & 1u))
            own_sums = nerf_mlp_partial_job(a->dwseg_s, a->dwseg_c, a->w0_cols_c, a->out_c, a->small_g, a->small_n, a->n_small,
                                            M, (hipStream_t)s, &sums) == 0;
        STEP(enerf_nerf_mlp_backward(a->g_rgbs, a->g_sigmas, 1.0f, a->feats, a->dirs, a->rgb, a->wseg_s, a->wseg_c,
                                     a->dwseg_s, a->dwseg_c, a->w0_cols_c, 1, M, a->out_c, a->dfeat, own_sums ? 3u : 1u, s));
        if (g_host_timing) slot++;
    } else {
        enerf_mlp32_defer_reduce(1);
        defer_set = true;
        STEP(enerf_mlp32_backward_p(a->g_rgbs, a->h32, a->wseg_c, a->dwseg_c, a->w0_cols_c, 1, 1, a->fb_c, M, 32, a->out_c,
                                    a->nh_c, 0, nullptr, a->dx32, 0, 0, a->rgb, a->out_c, nullptr, nullptr, 0, s));
        STEP(enerf_mlp32_backward_p(a->dx32, a->feats, a->wseg_s, a->dwseg_s, 32, 0, 1, a->fb_s, M, 32, 16, a->nh_s, 0,
                                    nullptr, a->dfeat, 1, 32, nullptr, 0, a->g_sigmas, a->h32, 32, s));
        enerf_mlp32_defer_reduce(0);
        defer_set = false;
    }
    if (rows_set) {
        enerf_mlp32_valid_rows(nullptr);
        rows_set = false;
    }
    // ---- the next batch's march, on the side stream, behind the MLP backward (it reads no parameter)
    if (a->next_rays_o && !count_carried) {
        enerf_mlp32_signal_next_reduce(0);
        signal_set = false;
        enerf_stream_t ss = a->side_stream;
        STEP(enerf_stream_wait_mlp32_signal(ss));
        // (near / far of the next batch inside the count pass: one launch less at the head of the chain the next step waits for)
        STEP(enerf_march_fuse_near_far(a->aabb, a->min_near));
        STEP(enerf_march_rays_train_ex(a->next_rays_o, a->next_rays_d, a->bitfield, a->bound, a->dt_gamma, a->max_steps,
                                       a->next_N, a->cascade, a->grid_size, a->next_M, a->next_nears, a->next_fars,
                                       a->next_xyzs, a->next_dirs, a->next_deltas, a->next_rays, a->next_counter, a->perturb,
                                       a->march_flags, ss));
    }
    // ---- table backward (record lists) and the optimizer
    if (a->flags & 1u) {
        // data parallel: the gradient has to exist to be averaged -- the backward's own flush into the dense buffer
        // (flags bit 1: the sharded tail with an owner range set -- enerf_grid_owner_range -- keeps this rank's own
        //  slice as record lists for the optimizer pass and flushes the rest)
        STEP(enerf_grid_encode_backward_ex(a->dfeat, a->xyzs, a->embeddings, a->offsets, a->table_grad, M, 3, 2, 16,
                                           a->level_scale_log2, a->base_resolution, 0, a->dfeat, a->dfeat, a->gridtype,
                                           ENERF_F32, 2, in_add, in_mul, (a->flags & 2u) ? 1u : 0u, (a->flags & 2u) ? M : 0u,
                                           s));
        goto done;
    }
    STEP(enerf_grid_encode_backward_ex(a->dfeat, a->xyzs, a->embeddings, a->offsets, a->table_grad, M, 3, 2, 16,
                                       a->level_scale_log2, a->base_resolution, 0, a->dfeat, a->dfeat, a->gridtype,
                                       ENERF_F32, 2, in_add, in_mul, 1, M, s));
    if (own_sums) grid_adam_partial_sums(&sums);
    if (count_carried) tile_adam_carry_count(&count_job);
    STEP(enerf_grid_adam_from_records_ex(a->table, a->table_grad, a->table_m, a->table_v, a->offsets, 16, 2, a->lr,
                                         a->beta1, a->beta2, a->eps, a->table_step, a->n_small, a->small_p, a->small_g,
                                         a->small_m, a->small_v, a->small_n, a->small_lr, a->small_step, s));
    if (own_sums && grid_adam_partial_sums(nullptr)) {
        set_error("train_step_mse: the optimizer launch did not take the weight gradients' partial sums");
        rc = ENERF_E_BADARG;
    }
    own_sums = false;
    if (count_carried && !rc) {
        // (an optimizer form that carries nothing -- loss scaling armed -- leaves the job waiting: counted by its own launch)
        if (tile_adam_carry_count(nullptr)) STEP(march_carry_count_now(&count_job, (hipStream_t)s));
        STEP(march_carry_end((hipStream_t)s));
        count_carried = false;
        g_carried_steps++;
    }
done:
#undef STEP
    // (one-shot march requests never outlive the step they were armed for -- csrc/raymarching.hip: MarchOneShot)
    enerf_march_fuse_near_far(nullptr, 0.0f);
    enerf_march_mirror_count(nullptr);
    grid_valid_rows(nullptr, 0, 0);
    if (count_carried) {                   // (failed between march_carry_begin and march_carry_end)
        tile_adam_carry_count(nullptr);
        march_carry_abort();
    }
    if (own_sums) grid_adam_partial_sums(nullptr);
    if (carry_set) {                       // (the launch that should have carried the fragments' build never ran)
        grid_fwd_carry(nullptr);
        nerf_mlp_frags_invalidate();
    }
    if (defer_set) enerf_mlp32_defer_reduce(0);
    if (signal_set) enerf_mlp32_signal_next_reduce(0);
    if (rows_set) enerf_mlp32_valid_rows(nullptr);
    if (prev_prec >= 0) enerf_mlp32_precision(prev_prec);
    return rc;
}

// The event-only step with both renders' samples as ONE batch of 2 M rows (enerf_event_step_args.flags bit 1)
static int train_step_events_merged(const enerf_event_step_args* a) {
    enerf_stream_t s = a->stream;
    const enerf_step_render &r0 = a->r[0], &r1 = a->r[1];
    const uint32_t N = r0.N, M = r0.M, M2 = 2 * M;
    const float in_add = a->bound, in_mul = a->inv_two_bound;
    if (r1.M != M || r1.xyzs != r0.xyzs + (size_t)3 * M || r1.dirs != r0.dirs + (size_t)3 * M ||
        r1.deltas != r0.deltas + (size_t)2 * M)
        ENERF_BADARG("train_step_events(merged): the second render's samples must follow the first's M rows");
    if (!a->m_feats || !a->m_h32 || !a->m_sigma || !a->m_rgb || !a->m_g_sigmas || !a->m_g_rgbs || !a->m_dx32 || !a->m_dfeat)
        ENERF_BADARG("train_step_events(merged): the m_* scratch buffers are required");
    int prev_prec = -1;
    if (a->mlp_precision >= 0) prev_prec = enerf_mlp32_precision(a->mlp_precision);
    int rc = 0;
    bool rows_set = false, defer_set = false, signal_set = false, carry_set = false, own_sums = false;
    uint32_t counts_carried = 0;           // the next step's marches riding in this step's optimizer launch
    MarchCountJob count_jobs[2] = {};
    uint32_t frags_built = 0;
    PartialSums sums{nullptr, nullptr, 0, 0, 0};
    const bool march_next = r0.next_rays_o != nullptr || r1.next_rays_o != nullptr;
    const bool skip = r0.counter != nullptr && r1.counter != nullptr;
    const bool fused_mlp = a->nh_s == 1 && a->nh_c == 2 && enerf_nerf_mlp_available() != 0;
#define STEP(call)           \
    do {                     \
        rc = (call);         \
        if (rc) goto done;   \
    } while (0)
    // (as in enerf_train_step_mse: the operand fragments' build rides in the grid forward's launch, the weight gradients'
    //  partial sums are summed by the optimizer's launch)
    if (fused_mlp && g_carry_frags) {
        SplitJob job;
        STEP(nerf_mlp_frag_job(a->wseg_s, a->wseg_c, a->w0_cols_c, a->out_c, (hipStream_t)s, &job));
        grid_fwd_carry(&job);
        carry_set = true;
    }
    if (skip) grid_valid_rows(r1.counter, M, M);            // (real rows: the first render's M + min(counter_2, M))
    STEP(enerf_grid_encode_forward(r0.xyzs, a->embeddings, a->offsets, a->m_feats, M2, 3, 2, 16, a->level_scale_log2,
                                   a->base_resolution, 0, a->m_feats, a->gridtype, ENERF_F32, 2, in_add, in_mul, s));
    if (carry_set) {
        frags_built = grid_fwd_carry(nullptr) ? 0u : 1u;
        if (!frags_built) nerf_mlp_frags_invalidate();
        carry_set = false;
    }
    if (skip) {
        enerf_mlp32_valid_rows_ex(r1.counter, M, M);         // real rows: the first render's M + min(counter_2, M)
        rows_set = true;
    }
    if (fused_mlp) {
        STEP(enerf_nerf_mlp_forward(a->m_feats, r0.dirs, a->wseg_s, a->wseg_c, a->w0_cols_c, M2, a->out_c, a->m_sigma,
                                    a->m_rgb, frags_built, s));
    } else {
        STEP(enerf_mlp32_forward_p(a->m_feats, a->wseg_s, 32, 0, M2, 32, 16, a->nh_s, 0, 6, a->m_fb_s, a->m_h32, 1, 32,
                                   a->m_sigma, r0.dirs, s));
        STEP(enerf_mlp32_forward_p(a->m_h32, a->wseg_c, a->w0_cols_c, 1, M2, 32, a->out_c, a->nh_c, 0, 3, a->m_fb_c, a->m_rgb,
                                   0, 0, nullptr, nullptr, s));
    }
    if (rows_set) {
        enerf_mlp32_valid_rows(nullptr);
        rows_set = false;
    }
    for (int k = 0; k < 2; k++) {
        const enerf_step_render& r = a->r[k];
        STEP(enerf_composite_rays_train_forward_blend(a->m_sigma + (size_t)k * M, a->m_rgb + (size_t)k * M * a->out_c, r.deltas,
                                                      r.rays, M, N, r.weights_sum, nullptr, r.image, a->bg_color, 0, 0.0f,
                                                      r.out_image, s));
    }
    STEP(enerf_event_loss_fwd_bwd(r0.out_image, r1.out_image, a->pols, N, a->use_luma, a->linlog, a->C_thres, a->log_thres,
                                  a->upstream, r0.g_image, r1.g_image, a->delta, a->loss, s));
    for (int k = 0; k < 2; k++) {
        const enerf_step_render& r = a->r[k];
        // (its tail blocks zero the gradients of rows [counter, M): the padding between the two renders' samples)
        STEP(enerf_composite_rays_train_backward_mse(r.g_image, nullptr, 1.0f, a->bg_color, 0, 0.0f, r.counter,
                                                     a->m_sigma + (size_t)k * M, a->m_rgb + (size_t)k * M * a->out_c, r.deltas,
                                                     r.rays, r.weights_sum, r.image, M, N, a->m_g_sigmas + (size_t)k * M,
                                                     a->m_g_rgbs + (size_t)k * M * a->out_c, nullptr, s));
    }
    if (skip) {
        enerf_mlp32_valid_rows_ex(r1.counter, M, M);
        rows_set = true;
    }
    // ---- the next step's two marches: carried by the optimizer's launch where that applies (both or neither; decided here,
    //      because the side-stream form needs its signal armed on the MLP backward)
    if (march_next && g_carry_count && !(a->march_flags & 16u)) {
        const uint32_t want = (r0.next_rays_o ? 1u : 0u) + (r1.next_rays_o ? 1u : 0u);
        for (int q = 0; q < 2; q++) {
            const enerf_step_render& n = a->r[q];
            if (!n.next_rays_o) continue;
            enerf_march_fuse_near_far(a->aabb, a->min_near);       // (near / far inside the count pass)
            const int b = march_carry_begin(n.next_rays_o, n.next_rays_d, a->bitfield, a->bound, a->dt_gamma, a->max_steps,
                                            n.next_N, a->cascade, a->grid_size, n.next_M, n.next_nears, n.next_fars,
                                            n.next_xyzs, n.next_dirs, n.next_deltas, n.next_rays, n.next_counter, a->perturb,
                                            a->march_flags, (hipStream_t)s, &count_jobs[counts_carried], want);
            if (b < 0) { rc = b; goto done; }
            if (b != 0) {                                           // not this way: both marches take the side stream
                enerf_march_fuse_near_far(nullptr, 0.0f);
                march_carry_abort();
                counts_carried = 0;
                break;
            }
            counts_carried++;
        }
    }
    if (march_next && !counts_carried) {
        enerf_mlp32_signal_next_reduce(1);
        signal_set = true;
    }
    if (fused_mlp) {
        if (g_fold_reduce)
            own_sums = nerf_mlp_partial_job(a->dwseg_s, a->dwseg_c, a->w0_cols_c, a->out_c, a->small_g, a->small_n, a->n_small,
                                            M2, (hipStream_t)s, &sums) == 0;
        STEP(enerf_nerf_mlp_backward(a->m_g_rgbs, a->m_g_sigmas, 1.0f, a->m_feats, r0.dirs, a->m_rgb, a->wseg_s, a->wseg_c,
                                     a->dwseg_s, a->dwseg_c, a->w0_cols_c, 1u, M2, a->out_c, a->m_dfeat, own_sums ? 3u : 1u, s));
    } else {
        enerf_mlp32_defer_reduce(1);
        defer_set = true;
        STEP(enerf_mlp32_backward_p(a->m_g_rgbs, a->m_h32, a->wseg_c, a->dwseg_c, a->w0_cols_c, 1, 1u, a->m_fb_c, M2, 32,
                                    a->out_c, a->nh_c, 0, nullptr, a->m_dx32, 0, 0, a->m_rgb, a->out_c, nullptr, nullptr, 0, s));
        STEP(enerf_mlp32_backward_p(a->m_dx32, a->m_feats, a->wseg_s, a->dwseg_s, 32, 0, 1u, a->m_fb_s, M2, 32, 16, a->nh_s, 0,
                                    nullptr, a->m_dfeat, 1, 32, nullptr, 0, a->m_g_sigmas, a->m_h32, 32, s));
        enerf_mlp32_defer_reduce(0);
        defer_set = false;
    }
    if (rows_set) {
        enerf_mlp32_valid_rows(nullptr);
        rows_set = false;
    }
    if (march_next && !counts_carried) {
        enerf_mlp32_signal_next_reduce(0);
        signal_set = false;
        enerf_stream_t ss = a->side_stream;
        STEP(enerf_stream_wait_mlp32_signal(ss));
        for (int q = 0; q < 2; q++) {
            const enerf_step_render& n = a->r[q];
            if (!n.next_rays_o) continue;
            STEP(enerf_march_fuse_near_far(a->aabb, a->min_near));       // (near / far inside the march's count pass)
            STEP(enerf_march_rays_train_ex(n.next_rays_o, n.next_rays_d, a->bitfield, a->bound, a->dt_gamma, a->max_steps,
                                           n.next_N, a->cascade, a->grid_size, n.next_M, n.next_nears, n.next_fars,
                                           n.next_xyzs, n.next_dirs, n.next_deltas, n.next_rays, n.next_counter, a->perturb,
                                           a->march_flags, ss));
        }
    }
    STEP(enerf_grid_encode_backward_ex(a->m_dfeat, r0.xyzs, a->embeddings, a->offsets, a->table_grad, M2, 3, 2, 16,
                                       a->level_scale_log2, a->base_resolution, 0, a->m_dfeat, a->m_dfeat, a->gridtype,
                                       ENERF_F32, 2, in_add, in_mul, 1, M2, s));
    if (own_sums) grid_adam_partial_sums(&sums);
    for (uint32_t q = 0; q < counts_carried; q++) tile_adam_carry_count(&count_jobs[q]);
    STEP(enerf_grid_adam_from_records_ex(a->table, a->table_grad, a->table_m, a->table_v, a->offsets, 16, 2, a->lr, a->beta1,
                                         a->beta2, a->eps, a->table_step, a->n_small, a->small_p, a->small_g, a->small_m,
                                         a->small_v, a->small_n, a->small_lr, a->small_step, s));
    if (own_sums && grid_adam_partial_sums(nullptr)) {
        set_error("train_step_events: the optimizer launch did not take the weight gradients' partial sums");
        rc = ENERF_E_BADARG;
    }
    own_sums = false;
    if (counts_carried && !rc) {
        // (an optimizer form that carries nothing leaves the jobs waiting: counted by launches of their own)
        if (tile_adam_carry_count(nullptr))
            for (uint32_t q = 0; q < counts_carried; q++) STEP(march_carry_count_now(&count_jobs[q], (hipStream_t)s));
        STEP(march_carry_end((hipStream_t)s));
        counts_carried = 0;
        g_carried_steps++;
    }
done:
#undef STEP
    // (one-shot march requests never outlive the step they were armed for -- csrc/raymarching.hip: MarchOneShot)
    enerf_march_fuse_near_far(nullptr, 0.0f);
    enerf_march_mirror_count(nullptr);
    grid_valid_rows(nullptr, 0, 0);
    if (counts_carried) {                  // (failed between march_carry_begin and march_carry_end)
        tile_adam_carry_count(nullptr);
        march_carry_abort();
    }
    if (own_sums) grid_adam_partial_sums(nullptr);
    if (carry_set) {                       // (the launch that should have carried the fragments' build never ran)
        grid_fwd_carry(nullptr);
        nerf_mlp_frags_invalidate();
    }
    if (defer_set) enerf_mlp32_defer_reduce(0);
    if (signal_set) enerf_mlp32_signal_next_reduce(0);
    if (rows_set) enerf_mlp32_valid_rows(nullptr);
    if (prev_prec >= 0) enerf_mlp32_precision(prev_prec);
    return rc;
}

// The event-only step (two renders, one loss, one optimizer pass): events.train_step_events_manual +
// FusedAdam.step_grid_table, call for call.
extern "C" int enerf_train_step_events(const enerf_event_step_args* a) {
    if (!a) ENERF_BADARG("train_step_events: null arguments");
    if (a->struct_bytes != sizeof(enerf_event_step_args))
        ENERF_BADARG("train_step_events: struct of %u bytes, this library expects %zu", a->struct_bytes,
                     sizeof(enerf_event_step_args));
    if (a->r[0].N == 0 || a->r[0].N != a->r[1].N || a->r[0].M == 0 || a->r[1].M == 0)
        ENERF_BADARG("train_step_events: both renders take the same (non-zero) number of rays and a sample budget");
    if (!a->bg_color || !a->pols) ENERF_BADARG("train_step_events: bg_color and pols are required");
    if (a->flags & 2u) return train_step_events_merged(a);
    enerf_stream_t s = a->stream;
    const uint32_t N = a->r[0].N, total = a->r[0].M + a->r[1].M;
    const float in_add = a->bound, in_mul = a->inv_two_bound;
    int prev_prec = -1;
    if (a->mlp_precision >= 0) prev_prec = enerf_mlp32_precision(a->mlp_precision);
    int rc = 0;
    bool rows_set = false, defer_set = false, signal_set = false;
    const bool march_next = a->r[0].next_rays_o != nullptr || a->r[1].next_rays_o != nullptr;
    const bool fused_mlp = a->nh_s == 1 && a->nh_c == 2 && enerf_nerf_mlp_available() != 0;
#define STEP(call)           \
    do {                     \
        rc = (call);         \
        if (rc) goto done;   \
    } while (0)
    // ---- the two renders' forward
    for (int k = 0; k < 2; k++) {
        const enerf_step_render& r = a->r[k];
        grid_valid_rows(r.counter, 0, 0);
        STEP(enerf_grid_encode_forward(r.xyzs, a->embeddings, a->offsets, r.feats, r.M, 3, 2, 16, a->level_scale_log2,
                                       a->base_resolution, 0, r.feats, a->gridtype, ENERF_F32, 2, in_add, in_mul, s));
        if (r.counter) {
            enerf_mlp32_valid_rows(r.counter);
            rows_set = true;
        }
        if (fused_mlp) {
            STEP(enerf_nerf_mlp_forward(r.feats, r.dirs, a->wseg_s, a->wseg_c, a->w0_cols_c, r.M, a->out_c, r.sigma, r.rgb,
                                        k == 0 ? 0u : 1u, s));
        } else {
            STEP(enerf_mlp32_forward_p(r.feats, a->wseg_s, 32, 0, r.M, 32, 16, a->nh_s, 0, 6, r.fb_s, r.h32, 1, 32, r.sigma,
                                       r.dirs, s));
            STEP(enerf_mlp32_forward_p(r.h32, a->wseg_c, a->w0_cols_c, 1, r.M, 32, a->out_c, a->nh_c, 0, 3, r.fb_c, r.rgb, 0,
                                       0, nullptr, nullptr, s));
        }
        if (rows_set) {
            enerf_mlp32_valid_rows(nullptr);
            rows_set = false;
        }
        STEP(enerf_composite_rays_train_forward_blend(r.sigma, r.rgb, r.deltas, r.rays, r.M, N, r.weights_sum, nullptr,
                                                      r.image, a->bg_color, 0, 0.0f, r.out_image, s));
    }
    // ---- the loss and its gradient with respect to the two images
    STEP(enerf_event_loss_fwd_bwd(a->r[0].out_image, a->r[1].out_image, a->pols, N, a->use_luma, a->linlog, a->C_thres,
                                  a->log_thres, a->upstream, a->r[0].g_image, a->r[1].g_image, a->delta, a->loss, s));
    // ---- the two renders' backward (the second adds its weight gradients to the first's)
    for (int k = 0; k < 2; k++) {
        const enerf_step_render& r = a->r[k];
        STEP(enerf_composite_rays_train_backward_mse(r.g_image, nullptr, 1.0f, a->bg_color, 0, 0.0f, r.counter, r.sigma,
                                                     r.rgb, r.deltas, r.rays, r.weights_sum, r.image, r.M, N, r.g_sigmas,
                                                     r.g_rgbs, nullptr, s));
        if (r.counter) {
            enerf_mlp32_valid_rows(r.counter);
            rows_set = true;
        }
        if (k == 1 && march_next) {
            enerf_mlp32_signal_next_reduce(1);
            signal_set = true;
        }
        const uint32_t overwrite = k == 0 ? 1u : 0u;
        if (fused_mlp) {
            STEP(enerf_nerf_mlp_backward(r.g_rgbs, r.g_sigmas, 1.0f, r.feats, r.dirs, r.rgb, a->wseg_s, a->wseg_c, a->dwseg_s,
                                         a->dwseg_c, a->w0_cols_c, overwrite, r.M, a->out_c, r.dfeat, 1, s));
        } else {
            enerf_mlp32_defer_reduce(1);
            defer_set = true;
            STEP(enerf_mlp32_backward_p(r.g_rgbs, r.h32, a->wseg_c, a->dwseg_c, a->w0_cols_c, 1, overwrite, r.fb_c, r.M, 32,
                                        a->out_c, a->nh_c, 0, nullptr, r.dx32, 0, 0, r.rgb, a->out_c, nullptr, nullptr, 0, s));
            STEP(enerf_mlp32_backward_p(r.dx32, r.feats, a->wseg_s, a->dwseg_s, 32, 0, overwrite, r.fb_s, r.M, 32, 16, a->nh_s,
                                        0, nullptr, r.dfeat, 1, 32, nullptr, 0, r.g_sigmas, r.h32, 32, s));
            enerf_mlp32_defer_reduce(0);
            defer_set = false;
        }
        if (rows_set) {
            enerf_mlp32_valid_rows(nullptr);
            rows_set = false;
        }
        if (k == 1 && march_next) {
            // the next step's two marches, on the side stream, behind this step's last MLP backward
            enerf_mlp32_signal_next_reduce(0);
            signal_set = false;
            enerf_stream_t ss = a->side_stream;
            STEP(enerf_stream_wait_mlp32_signal(ss));
            for (int q = 0; q < 2; q++) {
                const enerf_step_render& n = a->r[q];
                if (!n.next_rays_o) continue;
                STEP(enerf_march_fuse_near_far(a->aabb, a->min_near));       // (near / far inside the march's count pass)
                STEP(enerf_march_rays_train_ex(n.next_rays_o, n.next_rays_d, a->bitfield, a->bound, a->dt_gamma,
                                               a->max_steps, n.next_N, a->cascade, a->grid_size, n.next_M, n.next_nears,
                                               n.next_fars, n.next_xyzs, n.next_dirs, n.next_deltas, n.next_rays,
                                               n.next_counter, a->perturb, a->march_flags, ss));
            }
        }
        grid_valid_rows(r.counter, 0, 0);
        STEP(enerf_grid_encode_backward_ex(r.dfeat, r.xyzs, a->embeddings, a->offsets, a->table_grad, r.M, 3, 2, 16,
                                           a->level_scale_log2, a->base_resolution, 0, r.dfeat, r.dfeat, a->gridtype,
                                           ENERF_F32, 2, in_add, in_mul, 1, total, s));
    }
    STEP(enerf_grid_adam_from_records_ex(a->table, a->table_grad, a->table_m, a->table_v, a->offsets, 16, 2, a->lr,
                                         a->beta1, a->beta2, a->eps, a->table_step, a->n_small, a->small_p, a->small_g,
                                         a->small_m, a->small_v, a->small_n, a->small_lr, a->small_step, s));
done:
#undef STEP
    // (one-shot march requests never outlive the step they were armed for -- csrc/raymarching.hip: MarchOneShot)
    enerf_march_fuse_near_far(nullptr, 0.0f);
    enerf_march_mirror_count(nullptr);
    grid_valid_rows(nullptr, 0, 0);
    if (defer_set) enerf_mlp32_defer_reduce(0);
    if (signal_set) enerf_mlp32_signal_next_reduce(0);
    if (rows_set) enerf_mlp32_valid_rows(nullptr);
    if (prev_prec >= 0) enerf_mlp32_precision(prev_prec);
    return rc;
}

extern "C" int enerf_debug_fold_reduce(int on) {
    const int prev = g_fold_reduce ? 1 : 0;
    if (on >= 0) g_fold_reduce = on != 0;
    return prev;
}

extern "C" int enerf_debug_carry_frags(int on) {
    const int prev = g_carry_frags ? 1 : 0;
    if (on >= 0) g_carry_frags = on != 0;
    return prev;
}

// development aid: on != 0 starts (and clears) the per-call host timers of enerf_train_step_mse; out (16 doubles, may
// be NULL) receives the microseconds per call slot, in call order, averaged over the steps since the last start
extern "C" int enerf_debug_step_timing(int on, double* out) {
    if (out)
        for (int k = 0; k < 16; k++) out[k] = g_host_steps ? g_host_us[k] / (double)g_host_steps : 0.0;
    if (on >= 0) {
        g_host_timing = on != 0;
        for (int k = 0; k < 16; k++) g_host_us[k] = 0.0;
        g_host_steps = 0;
    }
    return 0;
}
