// event_pairs.hip -- one launch per training step for what EventNeRFDataset.collate does per event pair on the host
// (nerf/provider.py:1364-1441):
//   accumulate_evs branch (:1367-1398): drawn event -> step back if it is the last at its pixel -> window end among
//     its next min(num_successor, acc_max_num_evs + 1) events -> polarity sum over the window (prefix sums);
//   "computing poses online" branch (:1411-1420): camera pose at the two event times from the pose track -- rotation by
//     scipy's Slerp (R_i * exp(alpha * log(R_i^T R_{i+1}))), translation by interp1d(kind="cubic") -- evaluated here
//     from per-segment tables prepared once (enerf_amd/pose_interp.py), in double like scipy, then rounded to fp32 as
//     `torch.Tensor(get_hom_trafos(...))` does;
//   get_event_rays (nerf/utils.py:184-216): pixel -> unit camera direction -> world direction / origin at both poses.
// One thread per pair; everything a pair needs is 2 table rows + 2 track segments.
#include "common.h"

namespace {

struct Intr {
    float fx, fy, cx, cy;
};

// index of the track segment [knots[i], knots[i+1]] holding t (last segment for t == knots[K-1]); -1 outside the track
__device__ __forceinline__ int find_segment(const double* __restrict__ knots, uint32_t K, double t) {
    if (!(t >= knots[0]) || !(t <= knots[K - 1])) return -1;
    uint32_t lo = 0, hi = K - 1;                      // invariant: knots[lo] <= t <= knots[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (knots[mid] <= t) lo = mid; else hi = mid;
    }
    return (int)lo;
}

// c2w[3][4] (fp32) at time t
__device__ __forceinline__ bool pose_at(const double* __restrict__ knots, const double* __restrict__ rot,
                                        const double* __restrict__ rotvec, const double* __restrict__ tcoef, uint32_t K,
                                        double t, float (&m)[3][4]) {
    const int s = find_segment(knots, K, t);
    if (s < 0) return false;
    const double h = knots[s + 1] - knots[s];
    const double alpha = (t - knots[s]) / h;
    // Rodrigues: exp(alpha * w)
    const double wx = alpha * rotvec[s * 3], wy = alpha * rotvec[s * 3 + 1], wz = alpha * rotvec[s * 3 + 2];
    const double th2 = wx * wx + wy * wy + wz * wz;
    const double th = sqrt(th2);
    double a, b;                                      // exp = I + a [w]x + b [w]x^2
    if (th < 1e-6) {
        a = 1.0 - th2 / 6.0;
        b = 0.5 - th2 / 24.0;
    } else {
        a = sin(th) / th;
        b = (1.0 - cos(th)) / th2;
    }
    double E[3][3];
    E[0][0] = 1.0 - b * (wy * wy + wz * wz); E[0][1] = -a * wz + b * wx * wy;        E[0][2] = a * wy + b * wx * wz;
    E[1][0] = a * wz + b * wx * wy;          E[1][1] = 1.0 - b * (wx * wx + wz * wz); E[1][2] = -a * wx + b * wy * wz;
    E[2][0] = -a * wy + b * wx * wz;         E[2][1] = a * wx + b * wy * wz;          E[2][2] = 1.0 - b * (wx * wx + wy * wy);
    const double* R = rot + (size_t)s * 9;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            m[i][j] = (float)(R[i * 3] * E[0][j] + R[i * 3 + 1] * E[1][j] + R[i * 3 + 2] * E[2][j]);
    // translation: cubic in (t - knots[s]), coefficients highest power first: tcoef[s][k][axis]
    const double u = t - knots[s];
    const double* c = tcoef + (size_t)s * 12;
    for (int ax = 0; ax < 3; ax++) m[ax][3] = (float)(((c[ax] * u + c[3 + ax]) * u + c[6 + ax]) * u + c[9 + ax]);
    return true;
}

__device__ __forceinline__ void ray_of(const float (&m)[3][4], float dx, float dy, float dz, float* o, float* d) {
    o[0] = m[0][3]; o[1] = m[1][3]; o[2] = m[2][3];
    // torch.sum(dirs_cams[..., None, :] * c2w[..., :3, :3], axis=-1): products rounded, then summed left to right
    for (int i = 0; i < 3; i++) d[i] = (dx * m[i][0] + dy * m[i][1]) + dz * m[i][2];
}

__global__ void __launch_bounds__(256) k_event_pair_rays(
    const float* __restrict__ events, const uint8_t* __restrict__ no_successor, const int64_t* __restrict__ num_successor,
    const double* __restrict__ pol_cumsum, uint32_t N, const int64_t* __restrict__ start_draw,
    const double* __restrict__ u_end, uint32_t M, uint32_t acc_max_num_evs, const double* __restrict__ knots,
    const double* __restrict__ rot, const double* __restrict__ rotvec, const double* __restrict__ tcoef, uint32_t K,
    Intr in, float* __restrict__ o1, float* __restrict__ d1, float* __restrict__ o2, float* __restrict__ d2,
    float* __restrict__ pols, int64_t* __restrict__ start_out, int64_t* __restrict__ end_out, int* __restrict__ outside) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    int64_t s = start_draw[k];
    s -= (int64_t)no_successor[s];                                   // last event of its pixel: take the one before
    int64_t ns = num_successor[s];
    if (acc_max_num_evs && ns > (int64_t)acc_max_num_evs + 1) ns = (int64_t)acc_max_num_evs + 1;
    int64_t step = (int64_t)floor(u_end[k] * (double)ns);            // randint(s + 1, s + 1 + ns) from a uniform in [0,1)
    if (step > ns - 1) step = ns - 1;
    const int64_t e = s + 1 + step;
    start_out[k] = s;
    end_out[k] = e;
    pols[k] = (float)(pol_cumsum[e + 1] - pol_cumsum[s + 1]);
    const float x = events[(size_t)s * 4], y = events[(size_t)s * 4 + 1];
    // camera direction (get_event_rays): fp32, z = 1
    const float us = (x - in.cx) / in.fx, vs = (y - in.cy) / in.fy;
    const float nrm = sqrtf((us * us + vs * vs) + 1.0f);
    const float dx = us / nrm, dy = vs / nrm, dz = 1.0f / nrm;
    float m[3][4];
    const bool ok1 = pose_at(knots, rot, rotvec, tcoef, K, (double)events[(size_t)s * 4 + 2], m);
    ray_of(m, dx, dy, dz, o1 + (size_t)k * 3, d1 + (size_t)k * 3);
    const bool ok2 = pose_at(knots, rot, rotvec, tcoef, K, (double)events[(size_t)e * 4 + 2], m);
    ray_of(m, dx, dy, dz, o2 + (size_t)k * 3, d2 + (size_t)k * 3);
    if (!(ok1 && ok2)) atomicAdd(outside, 1);                        // interp1d(bounds_error=True) would raise
}


// ---- event loss, forward and gradient in one launch (nerf/utils.py:499-516 with C_thres != -1) ------------------------
// Per ray: (luma of) the two renders -> lin-log / log intensities -> delta = p2 - p1 -> (delta - pol * C)^2, mean over
// rays x channels.  Writes delta, d loss / d image1, d loss / d image2 and the loss (one workgroup, fixed summation
// order).  The arithmetic follows events.py (= utils/event_utils.py:23-66) operation by operation.
struct EventLossCfg {
    uint32_t use_luma, linlog;
    float c_thres, log_thres, upstream;
};
__device__ __forceinline__ float ev_luma(const float* rgb) {
    // torch.sum(rgb * factors, axis=-1): ((r*w0 + g*w1) + b*w2)
    return (rgb[0] * 0.299f + rgb[1] * 0.587f) + rgb[2] * 0.114f;
}
// intensity p(l) and dp/dl for l in image units (the reference scales by 255 first)
__device__ __forceinline__ void ev_intensity(float l, const EventLossCfg& c, float& p, float& dp) {
    const float x = l * 255.0f;
    if (c.linlog) {
        const float slope = (float)(2.995732273553991 / 20.0);          // np.log(20) / 20, rounded to fp32 by torch
        if (x < 20.0f) { p = slope * x; dp = slope * 255.0f; }
        else { p = logf(x); dp = (1.0f / x) * 255.0f; }
    } else {
        if (x > c.log_thres) { p = logf(x); dp = (1.0f / x) * 255.0f; }
        else { p = logf(c.log_thres); dp = 0.0f; }
    }
}
__global__ void __launch_bounds__(1024) k_event_loss(const float* __restrict__ img1, const float* __restrict__ img2,
                                                     const float* __restrict__ pols, uint32_t N, EventLossCfg c,
                                                     float* __restrict__ g1, float* __restrict__ g2,
                                                     float* __restrict__ delta, float* __restrict__ loss) {
    __shared__ double red[1024];
    const uint32_t ch = c.use_luma ? 1u : 3u;
    const float inv_count = 1.0f / (float)(N * ch);
    double acc = 0.0;
    for (uint32_t r = threadIdx.x; r < N; r += 1024u) {
        const float* a = img1 + (size_t)r * 3;
        const float* b = img2 + (size_t)r * 3;
        const float target = pols[r] * c.c_thres;
        float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f};
        if (c.use_luma) {
            float p1, d1, p2, d2;
            ev_intensity(ev_luma(a), c, p1, d1);
            // without lin-log the reference evaluates the second term on the FIRST luma (nerf/utils.py:500): delta = 0
            ev_intensity(c.linlog ? ev_luma(b) : ev_luma(a), c, p2, d2);
            const float dl = p2 - p1, res = dl - target;
            delta[r] = dl;
            acc += (double)(res * res);
            const float gd = c.upstream * (2.0f * res * inv_count);
            const float w[3] = {0.299f, 0.587f, 0.114f};
            for (int k = 0; k < 3; k++) {
                if (c.linlog) { ga[k] = -gd * d1 * w[k]; gb[k] = gd * d2 * w[k]; }
                else ga[k] = (gd * d2 - gd * d1) * w[k];
            }
        } else {
            for (int k = 0; k < 3; k++) {
                float p1, d1, p2, d2;
                ev_intensity(a[k], c, p1, d1);
                ev_intensity(b[k], c, p2, d2);
                const float dl = p2 - p1, res = dl - target;
                delta[(size_t)r * 3 + k] = dl;
                acc += (double)(res * res);
                const float gd = c.upstream * (2.0f * res * inv_count);
                ga[k] = -gd * d1;
                gb[k] = gd * d2;
            }
        }
        for (int k = 0; k < 3; k++) {
            g1[(size_t)r * 3 + k] = ga[k];
            g2[(size_t)r * 3 + k] = gb[k];
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = (float)(red[0] * (double)inv_count) * c.upstream;
}
}  // namespace

extern "C" int enerf_event_pair_rays(const float* events, const uint8_t* no_successor, const int64_t* num_successor,
                                     const double* pol_cumsum, uint32_t N, const int64_t* start_draw,
                                     const double* u_end, uint32_t M, uint32_t acc_max_num_evs, const double* knots,
                                     const double* rot, const double* rotvec, const double* tcoef, uint32_t K, float fx,
                                     float fy, float cx, float cy, float* rays_o1, float* rays_d1, float* rays_o2,
                                     float* rays_d2, float* pols, int64_t* start_out, int64_t* end_out,
                                     int32_t* outside_track, enerf_stream_t stream) {
    if (M == 0) return 0;
    if (N < 2 || K < 2) ENERF_BADARG("event_pair_rays: need >= 2 events and >= 2 track poses (N=%u K=%u)", N, K);
    const Intr in = {fx, fy, cx, cy};
    k_event_pair_rays<<<enerf::div_up(M, 256), 256, 0, (hipStream_t)stream>>>(
        events, no_successor, num_successor, pol_cumsum, N, start_draw, u_end, M, acc_max_num_evs, knots, rot, rotvec,
        tcoef, K, in, rays_o1, rays_d1, rays_o2, rays_d2, pols, start_out, end_out, (int*)outside_track);
    ENERF_LAUNCH_CHECK("event_pair_rays");
    return 0;
}

extern "C" int enerf_event_loss_fwd_bwd(const float* image1, const float* image2, const float* pols, uint32_t N,
                                        uint32_t use_luma, uint32_t linlog, float C_thres, float log_thres, float upstream,
                                        float* grad_image1, float* grad_image2, float* delta, float* loss,
                                        enerf_stream_t stream) {
    if (N == 0) return 0;
    if (!image1 || !image2 || !pols || !grad_image1 || !grad_image2 || !delta)
        ENERF_BADARG("event_loss_fwd_bwd: images, pols, gradients and delta are required");
    if (C_thres == -1.0f) ENERF_BADARG("event_loss_fwd_bwd: the normalised loss (C_thres == -1) is not fused");
    const EventLossCfg c = {use_luma, linlog, C_thres, log_thres, upstream};
    k_event_loss<<<1, 1024, 0, (hipStream_t)stream>>>(image1, image2, pols, N, c, grad_image1, grad_image2, delta, loss);
    ENERF_LAUNCH_CHECK("event_loss_fwd_bwd");
    return 0;
}
