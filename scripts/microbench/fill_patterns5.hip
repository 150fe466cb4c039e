// Fifth sweep: completion-based pacing.  A wave of the tile kernel has 8 stores in flight at once; a single-store wave
// (the fastest pattern, linear U=1) has one.  Does limiting a tile wave to K outstanding stores (s_waitcnt vmcnt(K-1)
// after every store) bring the tile pattern closer to the single-store rate?  Also: occupancy needed by each variant.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
constexpr int H = 1080, W = 1920, B = 64;
constexpr size_t N = (size_t)B * H * W;
constexpr size_t PLANE_BYTES = (size_t)H * W * 4;

template <int AUX>
__device__ __forceinline__ void st(float* plane_base, size_t byte_off, vfloat4 v)
{
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(plane_base, 0, (int)PLANE_BYTES, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, 0, AUX);
}

template <int K>
__device__ __forceinline__ void wait_outstanding()
{
    if constexpr (K == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (K == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (K == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}

// K = max stores in flight per wave (0 = unlimited, the shipped behaviour)
template <int K, int AUX>
__global__ __launch_bounds__(64) void k_tile_limited(float* dst, float v)
{
    const int lane = threadIdx.x;
    const int tx = blockIdx.x, ty = blockIdx.y, plane = blockIdx.z;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = ty * 16 + sub * 8 + i;
        if (row < H) st<AUX>(p, ((size_t)row * W + col0) * 4, val);
        if (K > 0) wait_outstanding<K>();
    }
}

template <int AUX>
__global__ __launch_bounds__(256) void k_linear(float* dst, size_t n4, float v)
{
    const vfloat4 val = {v, v, v, v};
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const size_t byte = i * 16, plane = byte / PLANE_BYTES;
        st<AUX>(dst + plane * (PLANE_BYTES / 4), byte - plane * PLANE_BYTES, val);
    }
}

template <typename F>
float time_it(F launch, int iters = 30, int rounds = 5)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 100; ++i) launch();
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
    const dim3 grid(W / 128, (H + 15) / 16, B);
    for (int w = 0; w < 300; ++w) hipLaunchKernelGGL((k_tile_limited<0, 0>), grid, dim3(64), 0, 0, d, 1.0f);
    report("linear 1 store/wave aux=18", time_it([&] {
        hipLaunchKernelGGL((k_linear<18>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4, 1.0f); }));
#define T(K, A) report("tile128 R=8 max-in-flight=" #K " aux=" #A, time_it([&] { \
        hipLaunchKernelGGL((k_tile_limited<K, A>), grid, dim3(64), 0, 0, d, 1.0f); }))
    T(0, 18); T(4, 18); T(2, 18); T(1, 18);
    T(0, 0); T(4, 0); T(2, 0); T(1, 0);
    report("linear 1 store/wave aux=18 (again)", time_it([&] {
        hipLaunchKernelGGL((k_linear<18>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4, 1.0f); }));
    (void)hipFree(d);
    return 0;
}
