// Third sweep: the same store patterns under write-through / non-temporal cache policies (raw buffer stores, aux bits:
// 0 plain, 16 sc1, 18 sc1+nt, 2 nt).
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

// linear: block of 256 threads, U float4 per thread at stride 4 KB; addressing relative to the plane containing it
template <int U, int AUX>
__global__ __launch_bounds__(256) void k_linear(float* dst, size_t n4, float v)
{
    const vfloat4 val = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = (size_t)blockIdx.x * 256 * U + (size_t)u * 256 + threadIdx.x;
        if (i < n4) {
            size_t byte = i * 16, plane = byte / PLANE_BYTES;
            st<AUX>(dst + plane * (PLANE_BYTES / 4), byte - plane * PLANE_BYTES, val);
        }
    }
}

template <int R, int WPG, int AUX>
__global__ __launch_bounds__(WPG * 64) void k_tile128(float* dst, int tiles_x, int tiles_y, int n_tiles, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * WPG + wave;
    if (tile >= n_tiles) return;
    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, plane = t2 / tiles_y;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = ty * 2 * R + sub * R + i;
        if (row < H && col0 < W) st<AUX>(p, ((size_t)row * W + col0) * 4, val);
    }
}

// full-width band: WG = 15 waves, each 128 cols x 16 rows; all waves store row pair (i, i+8) at the same step
template <int AUX, bool SYNC>
__global__ __launch_bounds__(960) void k_band(float* dst, int bands_per_plane, int n_bands, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int band = blockIdx.x;
    if (band >= n_bands) return;
    const int plane = band / bands_per_plane, ty = band % bands_per_plane;
    const int sub = lane >> 5, col0 = wave * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = ty * 16 + 2 * i + sub;     // adjacent rows per step: the WG writes 15 KB contiguous per step
        if (row < H) st<AUX>(p, ((size_t)row * W + col0) * 4, val);
        if (SYNC) __syncthreads();
    }
}


// vstack: WG of NW waves owns ONE 128-column tile of 2*NW rows; wave w stores rows (2w, 2w+1) with ONE instruction
template <int NW, int AUX>
__global__ __launch_bounds__(NW * 64) void k_vstack(float* dst, int tiles_x, int tiles_y, int n_groups, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x;
    if (g >= n_groups) return;
    const int tx = g % tiles_x, t2 = g / tiles_x, ty = t2 % tiles_y, plane = t2 / tiles_y;
    const int row = ty * 2 * NW + 2 * wave + (lane >> 5), col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    if (row < H && col0 < W) st<AUX>(dst + (size_t)plane * H * W, ((size_t)row * W + col0) * 4, val);
}
// hstack256: wave = 256 cols x 1 row (one 1 KB store); WG of NW waves = NW consecutive rows of one 256-col tile
template <int NW, int AUX>
__global__ __launch_bounds__(NW * 64) void k_vstack256(float* dst, int tiles_x, int tiles_y, int n_groups, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x;
    if (g >= n_groups) return;
    const int tx = g % tiles_x, t2 = g / tiles_x, ty = t2 % tiles_y, plane = t2 / tiles_y;
    const int row = ty * NW + wave, col0 = tx * 256 + lane * 4;
    const vfloat4 val = {v, v, v, v};
    if (row < H && col0 < W) st<AUX>(dst + (size_t)plane * H * W, ((size_t)row * W + col0) * 4, val);
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
    for (int warm = 0; warm < 200; ++warm) hipLaunchKernelGGL((k_linear<1, 0>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4, 1.0f);
#define LIN(U, AUX) report("linear U=" #U " aux=" #AUX, time_it([&] { \
        hipLaunchKernelGGL((k_linear<U, AUX>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, d, n4, 1.0f); }))
    LIN(1, 0); LIN(1, 16); LIN(1, 18); LIN(1, 2);
    LIN(2, 0); LIN(2, 16); LIN(2, 18);
    LIN(8, 0); LIN(8, 16); LIN(8, 18);
#define T128(R, WPG, AUX) { int tx = W / 128, ty = (H + 2 * R - 1) / (2 * R), nt = B * tx * ty; \
        report("tile128 R=" #R " wpg=" #WPG " aux=" #AUX, time_it([&] { \
        hipLaunchKernelGGL((k_tile128<R, WPG, AUX>), dim3((nt + WPG - 1) / WPG), dim3(WPG * 64), 0, 0, d, tx, ty, nt, 1.0f); })); }
    T128(8, 4, 0) T128(8, 4, 16) T128(8, 4, 18)
    T128(8, 1, 0) T128(8, 1, 16) T128(8, 1, 18)
    T128(4, 1, 0) T128(4, 1, 18) T128(16, 1, 0) T128(16, 1, 18)
    T128(1, 4, 0) T128(1, 4, 18) T128(2, 4, 18)
#define BAND(AUX, SYNC) { int bpp = (H + 15) / 16, nb = B * bpp; \
        report("band 1920x16 aux=" #AUX " sync=" #SYNC, time_it([&] { \
        hipLaunchKernelGGL((k_band<AUX, SYNC>), dim3(nb), dim3(960), 0, 0, d, bpp, nb, 1.0f); })); }
    BAND(0, false) BAND(18, false) BAND(0, true) BAND(18, true) BAND(16, true)
#define VS(NW, AUX) { int tx = W / 128, ty = (H + 2 * NW - 1) / (2 * NW), ng = B * tx * ty; \
        report("vstack128 waves=" #NW " aux=" #AUX, time_it([&] { \
        hipLaunchKernelGGL((k_vstack<NW, AUX>), dim3(ng), dim3(NW * 64), 0, 0, d, tx, ty, ng, 1.0f); })); }
    VS(8, 0) VS(8, 16) VS(8, 18) VS(4, 18) VS(16, 18) VS(16, 0) VS(2, 18)
#define VS256(NW, AUX) { int tx = (W + 255) / 256, ty = (H + NW - 1) / NW, ng = B * tx * ty; \
        report("vstack256 waves=" #NW " aux=" #AUX, time_it([&] { \
        hipLaunchKernelGGL((k_vstack256<NW, AUX>), dim3(ng), dim3(NW * 64), 0, 0, d, tx, ty, ng, 1.0f); })); }
    VS256(8, 0) VS256(8, 18) VS256(16, 18) VS256(4, 18)
    (void)hipFree(d);
    return 0;
}
