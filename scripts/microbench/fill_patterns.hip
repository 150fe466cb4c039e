// Store-pattern microbenchmark for the heat-map writer (MI355X). Pure stores of a [64,1080,1920] fp32 batch
// with different thread->address mappings, to find which pattern reaches the streaming-write ceiling.
// Build: hipcc -O3 --offload-arch=gfx950 fill_patterns.hip -o fill_patterns ; run: ./fill_patterns
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

constexpr int H = 1080, W = 1920, B = 64;
constexpr size_t N = (size_t)B * H * W;

// P0: torch-like. block = 256 threads, thread writes U float4 at stride 256*16 B; block covers U*4 KB contiguous.
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_linear(vfloat4* dst, size_t n4, float v)
{
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const vfloat4 val = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + (size_t)u * 256;
        if (i < n4) {
            if (NT) __builtin_nontemporal_store(val, &dst[i]);
            else dst[i] = val;
        }
    }
}

// P1: splat-like tiles. wave tile = (32*4) cols x (2*R) rows, lane -> (sub=lane>>5, colgroup=lane&31), 4 waves per WG
// covering 4 adjacent column tiles. Non-persistent. 32-bit index math.
template <int R>
__global__ __launch_bounds__(256) void k_tile128(float* dst, int tiles_x, int tiles_y, int n_tiles, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, plane = t2 / tiles_y;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = ty * 2 * R + sub * R + i;
        if (row < H && col0 < W) *reinterpret_cast<vfloat4*>(p + (size_t)row * W + col0) = val;
    }
}

// P2: wave tile = 256 cols x R rows, one full 1 KB row segment per store instruction.
template <int R>
__global__ __launch_bounds__(256) void k_tile256(float* dst, int tiles_x, int tiles_y, int n_tiles, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, plane = t2 / tiles_y;
    const int col0 = tx * 256 + lane * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = ty * R + i;
        if (row < H && col0 < W) *reinterpret_cast<vfloat4*>(p + (size_t)row * W + col0) = val;
    }
}

// P3: row strips. A WG of 256 threads owns RS full rows of one plane (contiguous RS*7680 B); thread t writes
// float4 t, t+256, ... of the strip (fully linear within the strip).
template <int RS>
__global__ __launch_bounds__(256) void k_strip(float* dst, int strips_per_plane, int n_strips, float v)
{
    const int strip = blockIdx.x;
    if (strip >= n_strips) return;
    const int plane = strip / strips_per_plane, s = strip % strips_per_plane;
    const int row0 = s * RS;
    const int rows = min(RS, H - row0);
    vfloat4* p = reinterpret_cast<vfloat4*>(dst + (size_t)plane * H * W + (size_t)row0 * W);
    const int n4 = rows * (W / 4);
    const vfloat4 val = {v, v, v, v};
    for (int i = threadIdx.x; i < n4; i += 256) p[i] = val;
}

// P4: persistent variant of P1 (grid-stride over wave tiles).
template <int R>
__global__ __launch_bounds__(256) void k_tile128_persist(float* dst, int tiles_x, int tiles_y, int n_tiles, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const vfloat4 val = {v, v, v, v};
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, plane = t2 / tiles_y;
        const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
        float* p = dst + (size_t)plane * H * W;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int row = ty * 2 * R + sub * R + i;
            if (row < H && col0 < W) *reinterpret_cast<vfloat4*>(p + (size_t)row * W + col0) = val;
        }
    }
}

// P5: like P1 but the 4 waves of a WG stack VERTICALLY (same columns, consecutive row tiles).
template <int R>
__global__ __launch_bounds__(256) void k_tile128_vstack(float* dst, int tiles_x, int tiles_y4, int n_groups, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x;
    if (g >= n_groups) return;
    const int tx = g % tiles_x, t2 = g / tiles_x, tyg = t2 % tiles_y4, plane = t2 / tiles_y4;
    const int ty = tyg * 4 + wave;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = ty * 2 * R + sub * R + i;
        if (row < H && col0 < W) *reinterpret_cast<vfloat4*>(p + (size_t)row * W + col0) = val;
    }
}

template <typename F>
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

void report(const char* name, float ms) { printf("%-44s %8.4f ms  %8.1f GB/s\n", name, ms, N * 4.0 / ms / 1e6); }

int main()
{
    float* d;
    CK(hipMalloc(&d, N * 4));
    const size_t n4 = N / 4;
    report("hipMemsetAsync", time_it([&] { hipMemsetAsync(d, 0, N * 4, 0); }));
#define LIN(U, NT) report("linear U=" #U " nt=" #NT, time_it([&] { \
        hipLaunchKernelGGL((k_linear<U, NT>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, (vfloat4*)d, n4, 1.0f); }))
    LIN(1, false); LIN(2, false); LIN(4, false); LIN(8, false); LIN(16, false); LIN(4, true); LIN(8, true);
#define T128(R) { int tx = W / 128, ty = (H + 2 * R - 1) / (2 * R), nt = B * tx * ty; \
        report("tile128 R=" #R, time_it([&] { hipLaunchKernelGGL((k_tile128<R>), dim3((nt + 3) / 4), dim3(256), 0, 0, d, tx, ty, nt, 1.0f); })); \
        report("tile128 persistent(2048 WG) R=" #R, time_it([&] { hipLaunchKernelGGL((k_tile128_persist<R>), dim3(2048), dim3(256), 0, 0, d, tx, ty, nt, 1.0f); })); \
        report("tile128 persistent(1024 WG) R=" #R, time_it([&] { hipLaunchKernelGGL((k_tile128_persist<R>), dim3(1024), dim3(256), 0, 0, d, tx, ty, nt, 1.0f); })); \
        int ty4 = (ty + 3) / 4, ng = B * tx * ty4; \
        report("tile128 vstack R=" #R, time_it([&] { hipLaunchKernelGGL((k_tile128_vstack<R>), dim3(ng), dim3(256), 0, 0, d, tx, ty4, ng, 1.0f); })); }
    T128(4) T128(8) T128(16)
#define T256(R) { int tx = (W + 255) / 256, ty = (H + R - 1) / R, nt = B * tx * ty; \
        report("tile256 R=" #R, time_it([&] { hipLaunchKernelGGL((k_tile256<R>), dim3((nt + 3) / 4), dim3(256), 0, 0, d, tx, ty, nt, 1.0f); })); }
    T256(4) T256(8) T256(16)
#define STRIP(RS) { int spp = (H + RS - 1) / RS, ns = B * spp; \
        report("strip rows=" #RS, time_it([&] { hipLaunchKernelGGL((k_strip<RS>), dim3(ns), dim3(256), 0, 0, d, spp, ns, 1.0f); })); }
    STRIP(1) STRIP(2) STRIP(4) STRIP(8) STRIP(16)
    hipFree(d);
    return 0;
}
