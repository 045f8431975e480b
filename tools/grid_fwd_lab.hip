// Development lab for the hash-grid forward gather on MI355X (not part of the library): times variants of the
// D=3, C=2, fp32 kernel on (a) uniform random points, (b) ray-ordered points, and checks every variant against the
// baseline bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/grid_fwd_lab.hip -o /tmp/grid_fwd_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int L = 16, C = 2, D = 3;
struct Tab { float scale[L]; uint32_t res[L]; uint32_t off[L + 1]; };

__device__ __forceinline__ uint32_t row_of(uint32_t size, uint32_t res, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t stride = 1, index = 0;
    const uint32_t p[3] = {x, y, z};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        if (stride <= size) { index += p[d] * stride; stride *= (res + 1); }
    }
    if (stride > size) index = x ^ (y * 2654435761u) ^ (z * 805459861u);
    return index % size;
}

struct __attribute__((packed, aligned(8))) F4u { float v[4]; };

__device__ __forceinline__ bool decode(uint32_t nchunks, uint32_t& level, uint32_t& chunk) {
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, j = bid >> 3;
    level = xcd + 8u * (j / nchunks);
    chunk = j % nchunks;
    return level < L;
}

// VAR 0: eight 8-byte gathers.  VAR 1: x-neighbour corners that sit in adjacent rows are fetched by one 16-byte load.
// NT: non-temporal input loads / output stores.
template <int VAR, bool NT>
__device__ __forceinline__ void point_level(const float* __restrict__ in, const float2* __restrict__ grid, float2* outp,
                                            const Tab& tab, uint32_t level, uint32_t b) {
    const uint32_t off0 = tab.off[level], size = tab.off[level + 1] - off0, res = tab.res[level];
    const float scale = tab.scale[level];
    const float2* rows = grid + off0;
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; d++) x[d] = NT ? __builtin_nontemporal_load(in + (size_t)b * 3 + d) : in[(size_t)b * 3 + d];
    float pos[3]; uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        pos[d] = fmaf(x[d], scale, 0.5f);
        const float fl = floorf(pos[d]);
        pg[d] = (uint32_t)fl;
        pos[d] -= (float)pg[d];
    }
    float2 f[8];
    if (VAR == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = rows[row_of(size, res, pg[0] + (i & 1), pg[1] + ((i >> 1) & 1), pg[2] + (i >> 2))];
    } else {
        uint32_t r0[4], r1[4];
        bool adj[4];
        F4u q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r0[k] = row_of(size, res, pg[0], pg[1] + (k & 1), pg[2] + (k >> 1));
            r1[k] = row_of(size, res, pg[0] + 1, pg[1] + (k & 1), pg[2] + (k >> 1));
            const uint32_t lo = r0[k] < r1[k] ? r0[k] : r1[k];
            adj[k] = (r0[k] ^ r1[k]) == 1u || r1[k] == r0[k] + 1u;      // same aligned pair, or dense neighbours
            // non-adjacent lanes fetch the aligned pair that holds r0
            const uint32_t base = adj[k] ? lo : (r0[k] & ~1u);
            q[k] = *reinterpret_cast<const F4u*>(rows + base);
        }
        float2 e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            e[k] = make_float2(0.f, 0.f);
            if (!adj[k]) e[k] = rows[r1[k]];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool first = adj[k] ? (r0[k] < r1[k]) : ((r0[k] & 1u) == 0u);
            const float2 a = make_float2(q[k].v[0], q[k].v[1]), c = make_float2(q[k].v[2], q[k].v[3]);
            f[2 * k] = first ? a : c;
            f[2 * k + 1] = adj[k] ? (first ? c : a) : e[k];
        }
    }
    float r[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float w = 1;
#pragma unroll
        for (int d = 0; d < 3; d++) w *= ((i >> d) & 1) ? pos[d] : 1 - pos[d];
        r[0] = fmaf(w, f[i].x, r[0]);
        r[1] = fmaf(w, f[i].y, r[1]);
    }
    if (NT) {
        __builtin_nontemporal_store(r[0], &outp->x);
        __builtin_nontemporal_store(r[1], &outp->y);
    } else {
        *outp = make_float2(r[0], r[1]);
    }
}

// Round 6: P points per thread (paired loads), every point's gathers requested before the first is used -- more loads in
// flight per wavefront at the price of registers (occupancy).  Points b, b + 256, ... of a P * 256-point chunk.
template <int P>
__device__ __forceinline__ void point_level_multi(const float* __restrict__ in, const float2* __restrict__ grid, float2* outp,
                                                  size_t ostride, const Tab& tab, uint32_t level, uint32_t b0, uint32_t B) {
    const uint32_t off0 = tab.off[level], size = tab.off[level + 1] - off0, res = tab.res[level];
    const float scale = tab.scale[level];
    const float2* rows = grid + off0;
    float x[P][3];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const uint32_t b = b0 + 256u * p < B ? b0 + 256u * p : B - 1;
#pragma unroll
        for (int d = 0; d < 3; d++) x[p][d] = in[(size_t)b * 3 + d];
    }
    float pos[P][3];
    uint32_t r0[P][4], r1[P][4];
    bool adj[P][4];
    F4u q[P][4];
#pragma unroll
    for (int p = 0; p < P; p++) {
        uint32_t pg[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            pos[p][d] = fmaf(x[p][d], scale, 0.5f);
            const float fl = floorf(pos[p][d]);
            pg[d] = (uint32_t)fl;
            pos[p][d] -= (float)pg[d];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r0[p][k] = row_of(size, res, pg[0], pg[1] + (k & 1), pg[2] + (k >> 1));
            r1[p][k] = row_of(size, res, pg[0] + 1, pg[1] + (k & 1), pg[2] + (k >> 1));
            const uint32_t lo = r0[p][k] < r1[p][k] ? r0[p][k] : r1[p][k];
            adj[p][k] = (r0[p][k] ^ r1[p][k]) == 1u || r1[p][k] == r0[p][k] + 1u;
            q[p][k] = *reinterpret_cast<const F4u*>(rows + (adj[p][k] ? lo : (r0[p][k] & ~1u)));
        }
    }
    float2 e[P][4];
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            e[p][k] = make_float2(0.f, 0.f);
            if (!adj[p][k]) e[p][k] = rows[r1[p][k]];
        }
#pragma unroll
    for (int p = 0; p < P; p++) {
        float2 f[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool first = adj[p][k] ? (r0[p][k] < r1[p][k]) : ((r0[p][k] & 1u) == 0u);
            const float2 a = make_float2(q[p][k].v[0], q[p][k].v[1]), c = make_float2(q[p][k].v[2], q[p][k].v[3]);
            f[2 * k] = first ? a : c;
            f[2 * k + 1] = adj[p][k] ? (first ? c : a) : e[p][k];
        }
        float r[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float w = 1;
#pragma unroll
            for (int d = 0; d < 3; d++) w *= ((i >> d) & 1) ? pos[p][d] : 1 - pos[p][d];
            r[0] = fmaf(w, f[i].x, r[0]);
            r[1] = fmaf(w, f[i].y, r[1]);
        }
        if (b0 + 256u * p < B) outp[(size_t)p * 256 * ostride] = make_float2(r[0], r[1]);
    }
}
template <int P, int GX>
__global__ void __launch_bounds__(256) k_group_multi(const float* __restrict__ in, const float2* __restrict__ grid,
                                                     float2* __restrict__ out, uint32_t B, Tab tab, int layout, uint32_t nchunks) {
    constexpr uint32_t NG = 8 / GX;
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, j = bid >> 3;
    const uint32_t group = xcd / GX, member = xcd % GX;
    const uint32_t per = (nchunks + GX - 1) / GX;      // chunks (of P * 256 points) per member per level
    const uint32_t li = j / per;
    const uint32_t chunk = (j % per) * GX + member;
    const uint32_t round = li, posn = (round & 1u) ? (NG - 1 - group) : group;
    const int level = (int)L - 1 - (int)(round * NG + posn);
    if (level < 0 || chunk >= nchunks) return;
    const uint32_t b = chunk * 256 * P + threadIdx.x;
    if (b >= B) return;
    float2* o = out + (layout == 0 ? (size_t)level * B + b : (size_t)b * L + level);
    point_level_multi<P>(in, grid, o, layout == 0 ? 1 : L, tab, (uint32_t)level, b, B);
}

template <int VAR, bool NT>
__global__ void __launch_bounds__(256) k_static(const float* __restrict__ in, const float2* __restrict__ grid,
                                                float2* __restrict__ out, uint32_t B, Tab tab, int layout, uint32_t nchunks) {
    uint32_t level, chunk;
    if (!decode(nchunks, level, chunk)) return;
    const uint32_t b = chunk * 256 + threadIdx.x;
    if (b >= B) return;
    float2* o = out + (layout == 0 ? (size_t)level * B + b : (size_t)b * L + level);
    point_level<VAR, NT>(in, grid, o, tab, level, b);
}


// XCD-pair groups: XCDs (2g, 2g+1) share the levels dealt to group g in snake order from the finest level down, and
// split every level's chunks between them; both walk the group's levels in the same order.
template <int VAR, bool NT, int GX>
__global__ void __launch_bounds__(256) k_group(const float* __restrict__ in, const float2* __restrict__ grid,
                                               float2* __restrict__ out, uint32_t B, Tab tab, int layout, uint32_t nchunks) {
    constexpr uint32_t NG = 8 / GX;                    // groups
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, j = bid >> 3;
    const uint32_t group = xcd / GX, member = xcd % GX;
    const uint32_t per = (nchunks + GX - 1) / GX;      // chunks per member per level
    const uint32_t li = j / per;                       // li-th level of the group
    const uint32_t chunk = (j % per) * GX + member;
    const uint32_t round = li, posn = (round & 1u) ? (NG - 1 - group) : group;
    const int level = (int)L - 1 - (int)(round * NG + posn);
    if (level < 0 || chunk >= nchunks) return;
    const uint32_t b = chunk * 256 + threadIdx.x;
    if (b >= B) return;
    float2* o = out + (layout == 0 ? (size_t)level * B + b : (size_t)b * L + level);
    point_level<VAR, NT>(in, grid, o, tab, (uint32_t)level, b);
}

// persistent work-stealing: queue y holds levels {y, y+8} x chunks; a block drains its own XCD's queue, then the others'
template <int VAR, bool NT>
__global__ void __launch_bounds__(256) k_steal(const float* __restrict__ in, const float2* __restrict__ grid,
                                               float2* __restrict__ out, uint32_t B, Tab tab, int layout, uint32_t nchunks,
                                               uint32_t* __restrict__ ctr) {
    __shared__ uint32_t s_item;
    const uint32_t me = blockIdx.x & 7u;
    const uint32_t Q = 2 * nchunks;
    uint32_t victim = 0;
    while (true) {
        if (threadIdx.x == 0) {
            uint32_t item = 0xffffffffu;
            while (victim < 8) {
                const uint32_t y = (me + victim) & 7u;
                const uint32_t t = atomicAdd(&ctr[y], 1u);
                if (t < Q) { item = y * Q + t; break; }
                victim++;
            }
            s_item = item;
        }
        __syncthreads();
        const uint32_t item = s_item;
        __syncthreads();
        if (item == 0xffffffffu) break;
        if (threadIdx.x == 0 && victim >= 8) victim = 8;
        const uint32_t y = item / Q, t = item % Q;
        const uint32_t level = y + 8u * (t / nchunks), chunk = t % nchunks;
        const uint32_t b = chunk * 256 + threadIdx.x;
        if (b < B) {
            float2* o = out + (layout == 0 ? (size_t)level * B + b : (size_t)b * L + level);
            point_level<VAR, NT>(in, grid, o, tab, level, b);
        }
    }
}

static Tab make_tab(float desired, uint32_t log2T, std::vector<uint32_t>& offsets) {
    Tab t;
    const float S = log2f(exp2f(log2f(desired / 16.f) / (L - 1)));
    uint32_t off = 0;
    for (int l = 0; l < L; l++) {
        const float scale = exp2f(l * S) * 16.f - 1.0f;
        t.scale[l] = scale;
        t.res[l] = (uint32_t)ceil(scale) + 1;
        const uint32_t resn = (uint32_t)ceil(16.0 * pow((double)exp2f(S), l));
        uint64_t n = (uint64_t)(resn + 1) * (resn + 1) * (resn + 1);
        uint32_t sz = (uint32_t)(n < (1ull << log2T) ? n : (1ull << log2T));
        sz = (sz + 7) / 8 * 8;
        t.off[l] = off;
        off += sz;
    }
    t.off[L] = off;
    offsets.assign(t.off, t.off + L + 1);
    return t;
}

int main() {
    std::vector<uint32_t> offs;
    Tab tab = make_tab(2048.f * 3, 19, offs);
    const uint32_t rows = tab.off[L];
    printf("rows %u (%.1f MB)\n", rows, rows * 8 / 1e6);
    std::vector<float> hg((size_t)rows * 2);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : hg) v = rnd() * 2 - 1;
    float2* grid; CK(hipMalloc(&grid, hg.size() * 4 + 64)); CK(hipMemcpy(grid, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    uint32_t* ctr; CK(hipMalloc(&ctr, 64));

    struct Set { const char* name; std::vector<float> pts; };
    std::vector<Set> sets(2);
    sets[0].name = "uniform 2M";
    sets[0].pts.resize((size_t)2 * 1024 * 1024 * 3);
    for (auto& v : sets[0].pts) v = rnd();
    sets[1].name = "ray-ordered 139K";
    {   // 4096 rays x 34 consecutive samples, step 0.00096 (= sqrt(3)/1024/6 * 2 in the unit cube of a bound-3 scene)
        const int R = 4096, K = 34;
        sets[1].pts.resize((size_t)R * K * 3);
        for (int r = 0; r < R; r++) {
            float o[3] = {0.3f + 0.4f * rnd(), 0.3f + 0.4f * rnd(), 0.3f + 0.4f * rnd()};
            float d[3] = {rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f};
            const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1e-6f;
            for (int k = 0; k < K; k++)
                for (int c = 0; c < 3; c++) sets[1].pts[((size_t)r * K + k) * 3 + c] = o[c] + d[c] / n * 0.000564f * k;
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& st : sets) {
        const uint32_t B = (uint32_t)(st.pts.size() / 3);
        float* in; CK(hipMalloc(&in, st.pts.size() * 4)); CK(hipMemcpy(in, st.pts.data(), st.pts.size() * 4, hipMemcpyHostToDevice));
        float2 *out, *ref; CK(hipMalloc(&out, (size_t)B * L * 8)); CK(hipMalloc(&ref, (size_t)B * L * 8));
        const uint32_t nchunks = (B + 255) / 256, nblocks = 8 * nchunks * 2;
        std::vector<float> hr((size_t)B * L * 2), ho((size_t)B * L * 2);
        for (int layout = 0; layout < 2; layout++) {
            k_static<0, false><<<nblocks, 256>>>(in, grid, ref, B, tab, layout, nchunks);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
            for (int var = 0; var < 10; var++) {
                auto launch = [&]() {
                    switch (var) {
                        case 0: k_static<0, false><<<nblocks, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 1: k_static<0, true><<<nblocks, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 2: k_static<1, false><<<nblocks, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 3: k_static<1, true><<<nblocks, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 4: k_group<1, false, 2><<<8 * ((nchunks + 1) / 2) * 4, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 5: k_group<1, false, 4><<<8 * ((nchunks + 3) / 4) * 8, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 6: k_group<1, false, 8><<<8 * ((nchunks + 7) / 8) * 16, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 7: k_group<0, false, 2><<<8 * ((nchunks + 1) / 2) * 4, 256>>>(in, grid, out, B, tab, layout, nchunks); break;
                        case 8: { const uint32_t nc2 = (B + 511) / 512; k_group_multi<2, 2><<<8 * ((nc2 + 1) / 2) * 4, 256>>>(in, grid, out, B, tab, layout, nc2); } break;
                        case 9: { const uint32_t nc3 = (B + 1023) / 1024; k_group_multi<4, 2><<<8 * ((nc3 + 1) / 2) * 4, 256>>>(in, grid, out, B, tab, layout, nc3); } break;
                    }
                };
                CK(hipMemset(out, 0xff, (size_t)B * L * 8));
                launch();
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
                const bool same = memcmp(ho.data(), hr.data(), ho.size() * 4) == 0;
                for (int i = 0; i < 3; i++) launch();
                CK(hipEventRecord(e0));
                const int reps = 20;
                for (int i = 0; i < reps; i++) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                static const char* names[] = {"static 8x8B", "static 8x8B nt", "static paired", "static paired nt",
                                              "group2 paired", "group4 paired", "group8 paired", "group2 8x8B",
                                              "group2 paired x2pts", "group2 paired x4pts"};
                printf("[%s] layout %d %-22s %8.4f ms  %7.0f GB/s alg  %s\n", st.name, layout, names[var], ms,
                       B * 1164.0 / ms / 1e6, same ? "bit-exact" : "MISMATCH");
            }
        }
        hipFree(in); hipFree(out); hipFree(ref);
    }
    return 0;
}
