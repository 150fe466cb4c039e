// Fourth sweep: does spreading a wave's 8 stores over time (s_sleep between them) avoid the TA write-data FIFO jams
// that PMC counters show for multi-store waves?  Pattern: tile128 R=8, one wave per workgroup (the shipped geometry).
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

template <int SLEEP, int AUX, int PRE>
__global__ __launch_bounds__(64) void k_tile_paced(float* dst, float v)
{
    const int lane = threadIdx.x;
    const int tx = blockIdx.x, ty = blockIdx.y, plane = blockIdx.z;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
    if (PRE > 0) __builtin_amdgcn_s_sleep(PRE);  // emulate a compute phase of ~64*PRE cycles before the stores
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = ty * 16 + sub * 8 + i;
        if (row < H) st<AUX>(p, ((size_t)row * W + col0) * 4, val);
        if (SLEEP > 0 && i < 7) __builtin_amdgcn_s_sleep(SLEEP);
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
    const dim3 grid(W / 128, (H + 15) / 16, B);
    for (int w = 0; w < 300; ++w) hipLaunchKernelGGL((k_tile_paced<0, 0, 0>), grid, dim3(64), 0, 0, d, 1.0f);
#define T(S, A, P) report("tile128 R=8 sleep=" #S " aux=" #A " pre=" #P, time_it([&] { \
        hipLaunchKernelGGL((k_tile_paced<S, A, P>), grid, dim3(64), 0, 0, d, 1.0f); }))
    T(0, 0, 0); T(1, 0, 0); T(2, 0, 0); T(4, 0, 0); T(8, 0, 0); T(16, 0, 0); T(32, 0, 0);
    T(0, 18, 0); T(1, 18, 0); T(2, 18, 0); T(4, 18, 0); T(8, 18, 0); T(16, 18, 0); T(32, 18, 0);
    T(0, 18, 16); T(4, 18, 16); T(8, 18, 16); T(0, 18, 64); T(8, 18, 64);
    (void)hipFree(d);
    return 0;
}
