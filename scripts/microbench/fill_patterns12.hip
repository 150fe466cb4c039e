// Feasibility of a "pre-binned, one store per wave" heat-map writer (MI355X): single-store waves stream at 6.8-7.1 TB/s where
// the 8-store tile waves of splat_kernel plateau at 5.5-5.9.  A single-store wave cannot afford its own cull, so a
// pre-pass would leave one 128-bit object mask per 128 x 16 tile; the row wave (2 rows x 128 px, ONE 16-byte store per lane)
// reads the mask of its tile with a scalar load and splats only the listed objects.  This file measures whether such waves
// keep the single-store rate on the bench workload's geometry (64 x 1080 x 1920, N in [1,128], r in [2,60]).
// Build: hipcc -O3 --offload-arch=gfx950 fill_patterns12.hip -o fill_patterns12 ; run: ./fill_patterns12
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

constexpr int H = 1080, W = 1920, B = 64, NMAX = 128;
constexpr size_t N = (size_t)B * H * W;
constexpr int TX = W / 128, TY = (H + 15) / 16;   // mask tiles: 15 x 68 per plane (128 x 16)
constexpr int TXW = (W + 255) / 256, TYW = (H + 7) / 8;   // wide variant: 256 x 8 mask tiles

struct __attribute__((aligned(16))) Obj {
    int x, y;
    float c2;
    int r;
};

// MODE 0: pure store (no mask); 1: scalar mask load, store zeros regardless (latency only); 2: mask + splat of listed objects
// LAYOUT 0: one wave per block, grid (TX, H/2, B); 1: 4 waves = 4 adjacent column tiles; 2: 4 waves = 4 stacked row pairs
// LAYOUT 3: wave = 1 row x 256 px, 4 waves stacked (4 rows); 4: wave = 1 row x 256 px, 4 waves side by side (4 KB contiguous);
// 5: like 2 with 8 waves (512 threads) = one whole 128 x 16 mask tile per workgroup
template <int MODE, int LAYOUT, bool WT>
__global__ __launch_bounds__(LAYOUT == 0 ? 64 : (LAYOUT == 5 ? 512 : 256)) void k_rowwave(float* __restrict__ dst, const uint4* __restrict__ masks,
                                                                     const Obj* __restrict__ objs, float k)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tx, rp;
    if (LAYOUT == 3 || LAYOUT == 4) {
        int row;
        if (LAYOUT == 3) {
            tx = blockIdx.x;
            row = blockIdx.y * 4 + wave;
        } else {
            tx = blockIdx.x * 4 + wave;
            row = blockIdx.y;
        }
        const int col0 = tx * 256 + lane * 4;
        if (row >= H || tx >= TXW) return;
        const int plane = blockIdx.z;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE >= 1) {
            const uint4 m = masks[((size_t)plane * TYW + (row >> 3)) * TXW + tx];
            unsigned w[4] = {m.x, m.y, m.z, m.w};
            if (MODE == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned bits = __builtin_amdgcn_readfirstlane(w[q]);
                    while (bits) {
                        const int j = __builtin_ctz(bits) + 32 * q;
                        bits &= bits - 1;
                        const Obj o = objs[plane * NMAX + j];
                        const float dy = (float)(row - o.y);
                        const float ey = (fabsf(dy) <= (float)o.r) ? k * __builtin_amdgcn_exp2f(-(dy * dy) * o.c2) : __builtin_nanf("");
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float dx = (float)(col0 + c - o.x);
                            const float ex = (fabsf(dx) <= (float)o.r) ? __builtin_amdgcn_exp2f(-(dx * dx) * o.c2) : __builtin_nanf("");
                            float r;
                            asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(acc[c]), "v"(ex * ey));
                            acc[c] = r;
                        }
                    }
                }
            } else {
                acc[0] = (float)((w[0] | w[1] | w[2] | w[3]) & 1u) * 1e-30f;
            }
        }
        if (col0 >= W) return;
        const vfloat4 out = {acc[0], acc[1], acc[2], acc[3]};
        float* plane_ptr = dst + (size_t)plane * H * W;
        if (WT) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(plane_ptr, 0, (int)((size_t)H * W * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(out, rsrc, (int)(((size_t)row * W + col0) * 4), 0, 18);
        } else {
            *reinterpret_cast<vfloat4*>(plane_ptr + (size_t)row * W + col0) = out;
        }
        return;
    } else if (LAYOUT == 5) {
        tx = blockIdx.x;
        rp = blockIdx.y * 8 + wave;
        if (rp >= H / 2) return;
    } else if (LAYOUT == 0) {
        tx = blockIdx.x;
        rp = blockIdx.y;
    } else if (LAYOUT == 1) {
        tx = blockIdx.x * 4 + wave;
        rp = blockIdx.y;
        if (tx >= TX) return;
    } else {
        tx = blockIdx.x;
        rp = blockIdx.y * 4 + wave;
        if (rp >= H / 2) return;
    }
    const int plane = blockIdx.z;
    const int row = rp * 2 + (lane >> 5), col0 = tx * 128 + (lane & 31) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE >= 1) {
        const uint4 m = masks[((size_t)plane * TY + (rp >> 3)) * TX + tx];   // wave-uniform address: scalar load
        unsigned w[4] = {m.x, m.y, m.z, m.w};
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned bits = __builtin_amdgcn_readfirstlane(w[q]);
                while (bits) {
                    const int j = __builtin_ctz(bits) + 32 * q;
                    bits &= bits - 1;
                    const Obj o = objs[plane * NMAX + j];   // uniform: scalar load
                    const float dy = (float)(row - o.y);
                    const float ey = (fabsf(dy) <= (float)o.r) ? k * __builtin_amdgcn_exp2f(-(dy * dy) * o.c2) : __builtin_nanf("");
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float dx = (float)(col0 + c - o.x);
                        const float ex = (fabsf(dx) <= (float)o.r) ? __builtin_amdgcn_exp2f(-(dx * dx) * o.c2) : __builtin_nanf("");
                        float r;
                        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(acc[c]), "v"(ex * ey));
                        acc[c] = r;
                    }
                }
            }
        } else {
            acc[0] = (float)((w[0] | w[1] | w[2] | w[3]) & 1u) * 1e-30f;
        }
    }
    const vfloat4 out = {acc[0], acc[1], acc[2], acc[3]};
    float* plane_ptr = dst + (size_t)plane * H * W;
    if (WT) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(plane_ptr, 0, (int)((size_t)H * W * 4), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(out, rsrc, (int)(((size_t)row * W + col0) * 4), 0, 18);
    } else {
        *reinterpret_cast<vfloat4*>(plane_ptr + (size_t)row * W + col0) = out;
    }
}

// the pre-pass, objects of the plane staged in LDS (one coalesced load), one thread per mask tile
template <int TW_, int TH_>
__global__ __launch_bounds__(256) void k_bin_lds(const Obj* __restrict__ objs, const int* __restrict__ counts, uint4* __restrict__ masks,
                                                  int tiles_x, int tiles_y)
{
    __shared__ Obj s_obj[NMAX];
    const int plane = blockIdx.y;
    const int n = counts[plane];
    if ((int)threadIdx.x < n) s_obj[threadIdx.x] = objs[plane * NMAX + threadIdx.x];
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    const int tx = t % tiles_x, ty = t / tiles_x;
    const int x0 = tx * TW_, x1 = x0 + TW_, y0 = ty * TH_, y1 = min(y0 + TH_, H);
    unsigned w[4] = {0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
        const Obj o = s_obj[j];
        const bool hit = o.x - o.r < x1 && o.x + o.r >= x0 && o.y - o.r < y1 && o.y + o.r >= y0;
        w[j >> 5] |= hit ? (1u << (j & 31)) : 0u;
    }
    masks[(size_t)plane * tiles_y * tiles_x + t] = make_uint4(w[0], w[1], w[2], w[3]);
}

// the pre-pass: one thread per mask tile, objects of the plane through scalar loads
__global__ __launch_bounds__(256) void k_bin(const Obj* __restrict__ objs, const int* __restrict__ counts, uint4* __restrict__ masks)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int plane = blockIdx.y;
    if (t >= TX * TY) return;
    const int tx = t % TX, ty = t / TX;
    const int x0 = tx * 128, x1 = x0 + 128, y0 = ty * 16, y1 = min(y0 + 16, H);
    unsigned w[4] = {0, 0, 0, 0};
    const int n = counts[plane];
    for (int j = 0; j < n; ++j) {
        const Obj o = objs[plane * NMAX + j];
        const bool hit = o.x - o.r < x1 && o.x + o.r >= x0 && o.y - o.r < y1 && o.y + o.r >= y0;
        w[j >> 5] |= hit ? (1u << (j & 31)) : 0u;
    }
    masks[(size_t)plane * TY * TX + t] = make_uint4(w[0], w[1], w[2], w[3]);
}

template <class F>
float time_it(F launch, int iters = 30, int rounds = 5)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms / iters);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}
void report(const char* name, float ms) { printf("%-78s %8.4f ms  %8.1f GB/s\n", name, ms, N * 4 / ms / 1e6); }

int main()
{
    float* d;
    uint4* masks;
    Obj* objs;
    int* counts;
    CK(hipMalloc(&d, N * 4));
    CK(hipMalloc(&masks, (size_t)B * TX * TY * sizeof(uint4)));
    CK(hipMalloc(&objs, (size_t)B * NMAX * sizeof(Obj)));
    CK(hipMalloc(&counts, B * sizeof(int)));
    std::mt19937 rng(5);
    std::vector<Obj> ho((size_t)B * NMAX);
    std::vector<int> hc(B);
    for (int b = 0; b < B; ++b) {
        hc[b] = 1 + (int)(rng() % 128);
        for (int j = 0; j < NMAX; ++j) {
            const int r = 2 + (int)(rng() % 59);
            const float sigma = (2 * r + 1) / 6.0f;
            ho[(size_t)b * NMAX + j] = Obj{(int)(rng() % W), (int)(rng() % H), 1.4426950408889634f / (2 * sigma * sigma), r};
        }
    }
    CK(hipMemcpy(objs, ho.data(), ho.size() * sizeof(Obj), hipMemcpyHostToDevice));
    CK(hipMemcpy(counts, hc.data(), B * sizeof(int), hipMemcpyHostToDevice));
    const dim3 bin_grid((TX * TY + 255) / 256, B);
    hipLaunchKernelGGL(k_bin, bin_grid, dim3(256), 0, 0, objs, counts, masks);
    CK(hipDeviceSynchronize());
    {
        std::vector<uint4> hm((size_t)B * TX * TY);
        CK(hipMemcpy(hm.data(), masks, hm.size() * sizeof(uint4), hipMemcpyDeviceToHost));
        size_t nz = 0, bits = 0;
        for (auto& m : hm) {
            const int c = __builtin_popcount(m.x) + __builtin_popcount(m.y) + __builtin_popcount(m.z) + __builtin_popcount(m.w);
            nz += c > 0;
            bits += c;
        }
        printf("mask tiles: %zu, non-empty %.1f %%, objects per tile %.2f\n", hm.size(), 100.0 * nz / hm.size(), (double)bits / hm.size());
    }
    report("pre-pass k_bin alone", time_it([&] { hipLaunchKernelGGL(k_bin, bin_grid, dim3(256), 0, 0, objs, counts, masks); }));

#define RUN(MODE, LAYOUT, WT, GRID, name)                                                                                  \
    report(name, time_it([&] { hipLaunchKernelGGL((k_rowwave<MODE, LAYOUT, WT>), GRID, dim3(LAYOUT == 0 ? 64 : 256), 0, 0, d, \
                                                  masks, objs, 1.0f); }))
    const dim3 g0(TX, H / 2, B), g1((TX + 3) / 4, H / 2, B), g2(TX, (H / 2 + 3) / 4, B);
    const dim3 bin_grid_l((TX * TY + 255) / 256, B);
    report("pre-pass k_bin_lds (128 x 16 tiles) alone",
           time_it([&] { hipLaunchKernelGGL((k_bin_lds<128, 16>), bin_grid_l, dim3(256), 0, 0, objs, counts, masks, TX, TY); }));
    RUN(0, 0, false, g0, "row wave, pure store, 1 wave/WG");
    RUN(0, 1, false, g1, "row wave, pure store, 4 waves/WG side by side");
    RUN(0, 2, false, g2, "row wave, pure store, 4 waves/WG stacked");
    RUN(0, 0, true, g0, "row wave, pure store sc1 nt, 1 wave/WG");
    RUN(0, 2, true, g2, "row wave, pure store sc1 nt, 4 waves/WG stacked");
    RUN(1, 0, false, g0, "row wave, mask load + store, 1 wave/WG");
    RUN(1, 1, false, g1, "row wave, mask load + store, 4 waves/WG side by side");
    RUN(1, 2, false, g2, "row wave, mask load + store, 4 waves/WG stacked");
    RUN(1, 2, true, g2, "row wave, mask load + store sc1 nt, 4 waves/WG stacked");
    RUN(2, 0, false, g0, "row wave, mask + splat, 1 wave/WG");
    RUN(2, 1, false, g1, "row wave, mask + splat, 4 waves/WG side by side");
    RUN(2, 2, false, g2, "row wave, mask + splat, 4 waves/WG stacked");
    RUN(2, 0, true, g0, "row wave, mask + splat sc1 nt, 1 wave/WG");
    RUN(2, 2, true, g2, "row wave, mask + splat sc1 nt, 4 waves/WG stacked");
    const dim3 g5(TX, (H / 2 + 7) / 8, B);
    RUN(0, 5, true, g5, "row wave, pure store sc1 nt, 8 waves/WG stacked");
    RUN(2, 5, true, g5, "row wave, mask + splat sc1 nt, 8 waves/WG stacked");
    report("pre-pass(lds) + row wave (mask + splat sc1 nt, 4 stacked), two launches",
           time_it([&] {
               hipLaunchKernelGGL((k_bin_lds<128, 16>), bin_grid_l, dim3(256), 0, 0, objs, counts, masks, TX, TY);
               hipLaunchKernelGGL((k_rowwave<2, 2, true>), g2, dim3(256), 0, 0, d, masks, objs, 1.0f);
           }));
    // wide waves: masks re-binned on 256 x 8 tiles
    uint4* masks_w;
    CK(hipMalloc(&masks_w, (size_t)B * TXW * TYW * sizeof(uint4)));
    const dim3 bin_grid_w((TXW * TYW + 255) / 256, B);
    hipLaunchKernelGGL((k_bin_lds<256, 8>), bin_grid_w, dim3(256), 0, 0, objs, counts, masks_w, TXW, TYW);
    CK(hipDeviceSynchronize());
    report("pre-pass k_bin_lds (256 x 8 tiles) alone",
           time_it([&] { hipLaunchKernelGGL((k_bin_lds<256, 8>), bin_grid_w, dim3(256), 0, 0, objs, counts, masks_w, TXW, TYW); }));
#define RUNW(MODE, LAYOUT, WT, GRID, name)                                                                                \
    report(name, time_it([&] { hipLaunchKernelGGL((k_rowwave<MODE, LAYOUT, WT>), GRID, dim3(256), 0, 0, d, masks_w, objs, 1.0f); }))
    const dim3 g3(TXW, (H + 3) / 4, B), g4((TXW + 3) / 4, H, B);
    RUNW(0, 3, false, g3, "wide wave (1 row x 256 px), pure store, 4 stacked");
    RUNW(0, 3, true, g3, "wide wave, pure store sc1 nt, 4 stacked");
    RUNW(0, 4, false, g4, "wide wave, pure store, 4 side by side (4 KB contiguous)");
    RUNW(0, 4, true, g4, "wide wave, pure store sc1 nt, 4 side by side");
    RUNW(2, 3, false, g3, "wide wave, mask + splat, 4 stacked");
    RUNW(2, 3, true, g3, "wide wave, mask + splat sc1 nt, 4 stacked");
    RUNW(2, 4, false, g4, "wide wave, mask + splat, 4 side by side");
    RUNW(2, 4, true, g4, "wide wave, mask + splat sc1 nt, 4 side by side");
    report("pre-pass(lds) + wide wave (mask + splat sc1 nt, 4 side by side), two launches",
           time_it([&] {
               hipLaunchKernelGGL((k_bin_lds<256, 8>), bin_grid_w, dim3(256), 0, 0, objs, counts, masks_w, TXW, TYW);
               hipLaunchKernelGGL((k_rowwave<2, 4, true>), g4, dim3(256), 0, 0, d, masks_w, objs, 1.0f);
           }));
    report("pre-pass + row wave (mask + splat, stacked), two launches",
           time_it([&] {
               hipLaunchKernelGGL(k_bin, bin_grid, dim3(256), 0, 0, objs, counts, masks);
               hipLaunchKernelGGL((k_rowwave<2, 2, false>), g2, dim3(256), 0, 0, d, masks, objs, 1.0f);
           }));
    report("pre-pass + row wave (mask + splat, 1 wave/WG), two launches",
           time_it([&] {
               hipLaunchKernelGGL(k_bin, bin_grid, dim3(256), 0, 0, objs, counts, masks);
               hipLaunchKernelGGL((k_rowwave<2, 0, false>), g0, dim3(64), 0, 0, d, masks, objs, 1.0f);
           }));
    hipFree(d);
    return 0;
}
