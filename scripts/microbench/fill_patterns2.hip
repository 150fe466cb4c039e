// Second store-pattern sweep: why do waves with >1 store lose ~20% of write bandwidth on MI355X?
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
constexpr int H = 1080, W = 1920, B = 64;
constexpr size_t N = (size_t)B * H * W;

__device__ __forceinline__ float burn(float x, int iters)
{
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    return x;
}

// E1: one float4 per thread (as the 6.8 TB/s pattern) but after `work` dependent FMAs.
__global__ __launch_bounds__(256) void k_one_store_after_work(vfloat4* dst, size_t n4, float v, int work)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float x = burn(v, work);
    if (i < n4) dst[i] = vfloat4{x, x, x, x};
}

// E3: each wave writes U consecutive 1 KB pieces (wave-contiguous U KB), block covers 4*U KB contiguous.
template <int U>
__global__ __launch_bounds__(256) void k_wave_contig(vfloat4* dst, size_t n4, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t base = ((size_t)blockIdx.x * 4 + wave) * 64 * U + lane;
    const vfloat4 val = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + (size_t)u * 64;
        if (i < n4) dst[i] = val;
    }
}

// E5: single-wave workgroups, U consecutive 1 KB pieces.
template <int U>
__global__ __launch_bounds__(64) void k_wave_contig_wg64(vfloat4* dst, size_t n4, float v)
{
    size_t base = (size_t)blockIdx.x * 64 * U + threadIdx.x;
    const vfloat4 val = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + (size_t)u * 64;
        if (i < n4) dst[i] = val;
    }
}

// E6: occupancy-limited (dynamic LDS) version of the block-contiguous strided pattern (k_linear of sweep 1).
template <int U>
__global__ __launch_bounds__(256) void k_linear_lds(vfloat4* dst, size_t n4, float v)
{
    extern __shared__ float lds[];
    if (v == 12345.f) lds[threadIdx.x] = v;  // keep the allocation
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const vfloat4 val = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + (size_t)u * 256;
        if (i < n4) dst[i] = val;
    }
}

// E4: paced: wait for each store to retire before the next.
template <int U>
__global__ __launch_bounds__(256) void k_linear_paced(vfloat4* dst, size_t n4, float v)
{
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const vfloat4 val = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + (size_t)u * 256;
        if (i < n4) dst[i] = val;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// E7: persistent "one store per wave per iteration" with a dynamic counter? -> simple grid-stride U=1 per iter
__global__ __launch_bounds__(256) void k_gridstride(vfloat4* dst, size_t n4, float v)
{
    const vfloat4 val = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = val;
}

// E8: tile pattern (128 cols x 2R rows per wave) but with the stores of a wave spread over time by work
template <int R>
__global__ __launch_bounds__(256) void k_tile128_work(float* dst, int tiles_x, int tiles_y, int n_tiles, float v, int work)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, plane = t2 / tiles_y;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    float* p = dst + (size_t)plane * H * W;
    float x = burn(v, work);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = ty * 2 * R + sub * R + i;
        if (row < H && col0 < W) *reinterpret_cast<vfloat4*>(p + (size_t)row * W + col0) = vfloat4{x, x, x, x};
    }
}

// E9: dwordx2 / dword stores, one per thread (512 B / 256 B per wave instruction)
__global__ __launch_bounds__(256) void k_one_store_f2(float2* dst, size_t n2, float v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) dst[i] = make_float2(v, v);
}

template <typename F>
float time_it(F launch, int iters = 30, int rounds = 5)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 3; ++i) launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms / iters);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}
void report(const char* name, float ms) { printf("%-52s %8.4f ms  %8.1f GB/s\n", name, ms, N * 4.0 / ms / 1e6); }

int main()
{
    float* d;
    if (hipMalloc(&d, N * 4) != hipSuccess) return 1;
    const size_t n4 = N / 4;
    const unsigned g1 = (unsigned)((n4 + 255) / 256);
    char name[128];
    for (int work : {0, 100, 400, 1600, 6400}) {
        snprintf(name, sizeof name, "one store/wave after %d FMAs", work);
        report(name, time_it([&] { hipLaunchKernelGGL(k_one_store_after_work, dim3(g1), dim3(256), 0, 0, (vfloat4*)d, n4, 1.0f, work); }));
    }
#define WC(U) report("wave-contiguous U=" #U " (4 waves/WG)", time_it([&] { \
        hipLaunchKernelGGL((k_wave_contig<U>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, (vfloat4*)d, n4, 1.0f); })); \
    report("wave-contiguous U=" #U " (1 wave/WG)", time_it([&] { \
        hipLaunchKernelGGL((k_wave_contig_wg64<U>), dim3((unsigned)((n4 + 64 * U - 1) / (64 * U))), dim3(64), 0, 0, (vfloat4*)d, n4, 1.0f); }))
    WC(1); WC(2); WC(4); WC(8);
#define LDS(U, KB) report("block-strided U=" #U " LDS=" #KB "KB", time_it([&] { \
        hipLaunchKernelGGL((k_linear_lds<U>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), KB * 1024, 0, (vfloat4*)d, n4, 1.0f); }))
    LDS(8, 0); LDS(8, 20); LDS(8, 40); LDS(8, 80); LDS(8, 160); LDS(2, 40); LDS(2, 80); LDS(1, 40); LDS(1, 80);
#define PACED(U) report("block-strided paced U=" #U, time_it([&] { \
        hipLaunchKernelGGL((k_linear_paced<U>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, (vfloat4*)d, n4, 1.0f); }))
    PACED(2); PACED(4); PACED(8);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 32768}) {
        snprintf(name, sizeof name, "grid-stride 1 store/iter, %d WGs", blocks);
        report(name, time_it([&] { hipLaunchKernelGGL(k_gridstride, dim3(blocks), dim3(256), 0, 0, (vfloat4*)d, n4, 1.0f); }));
    }
    {
        int tx = W / 128, ty = (H + 15) / 16, nt = B * tx * ty;
        for (int work : {0, 400, 1600, 6400}) {
            snprintf(name, sizeof name, "tile128 R=8 after %d FMAs", work);
            report(name, time_it([&] { hipLaunchKernelGGL((k_tile128_work<8>), dim3((nt + 3) / 4), dim3(256), 0, 0, d, tx, ty, nt, 1.0f, work); }));
        }
        int ty1 = (H + 1) / 2, nt1 = B * tx * ty1;
        report("tile128 R=1 (2 rows x 128 cols, one store/wave)", time_it([&] { hipLaunchKernelGGL((k_tile128_work<1>), dim3((nt1 + 3) / 4), dim3(256), 0, 0, d, tx, ty1, nt1, 1.0f, 0); }));
        int ty2 = (H + 3) / 4, nt2 = B * tx * ty2;
        report("tile128 R=2", time_it([&] { hipLaunchKernelGGL((k_tile128_work<2>), dim3((nt2 + 3) / 4), dim3(256), 0, 0, d, tx, ty2, nt2, 1.0f, 0); }));
    }
    report("one float2 store/thread", time_it([&] { hipLaunchKernelGGL(k_one_store_f2, dim3((unsigned)((N / 2 + 255) / 256)), dim3(256), 0, 0, (float2*)d, N / 2, 1.0f); }));
    (void)hipFree(d);
    return 0;
}
