// Tenth sweep (round 2): producer / consumer inside a workgroup.  fill_patterns6: with write-through (sc1 nt) stores the
// tile pattern runs at 6.56 TB/s when only TWO tile waves per CU are issuing stores, and falls to 5.5 TB/s with 8-16 —
// the fewer store streams a CU interleaves, the better.  A splat kernel needs 16+ waves per CU for its arithmetic, so
// split the roles: NP "compute" waves finish 128x32 tiles into LDS slots, NS persistent "store" waves per workgroup
// drain the slots with 16 sc1-nt stores each.  Persistent workgroups (one or two per CU), static tile striding.
//   work   dummy FMAs per tile in the producer (the splat kernel spends ~600-900 VALU instructions per tile)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
constexpr int H = 1080, W = 1920, B = 64;
constexpr size_t N = (size_t)B * H * W;
constexpr size_t PLANE_BYTES = (size_t)H * W * 4;
constexpr int TX = W / 128, TY = (H + 31) / 32;
constexpr int N_TILES = B * TX * TY;  // 32640

template <int AUX>
__device__ __forceinline__ void st(float* plane_base, size_t byte_off, vfloat4 v)
{
    if constexpr (AUX == 0) {
        *reinterpret_cast<vfloat4*>(reinterpret_cast<char*>(plane_base) + byte_off) = v;
    } else {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(plane_base, 0, (int)PLANE_BYTES, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, 0, AUX);
    }
}

template <int AUX>
__global__ __launch_bounds__(64) void k_tile_static(float* dst, float v)
{
    extern __shared__ int dyn_lds[];
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int tx = tile % TX, t2 = tile / TX, ty = t2 % TY, plane = t2 / TY;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = ty * 32 + sub * 16 + i;
        if (row < H) st<AUX>(p, ((size_t)row * W + col0) * 4, val);
    }
}

__device__ __forceinline__ float burn(float x, int n)
{
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    return x;
}

template <int NP, int NS, int AUX>
__global__ __launch_bounds__((NP + NS) * 64) void k_pc(float* dst, float v, int work)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    vfloat4* slots = reinterpret_cast<vfloat4*>(lds_raw);                 // [NP][16 rows][64 lanes]
    volatile int* flag = reinterpret_cast<volatile int*>(lds_raw + (size_t)NP * 16384);  // [NP] 0 empty / 1 full
    volatile int* tid = flag + NP;                                       // [NP] tile id, -1 = producer finished
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < NP) flag[threadIdx.x] = 0;
    __syncthreads();
    if (wave < NP) {
        // ---- producer: tiles q, q + Q, ... of the global producer numbering
        const int q = blockIdx.x * NP + wave, Q = gridDim.x * NP;
        vfloat4* my = slots + (size_t)wave * 1024;
        for (int tile = q;; tile += Q) {
            const bool last = tile >= N_TILES;
            float x = last ? v : burn(v, work);
            for (int spin = 0; flag[wave] != 0 && spin < (1 << 24); ++spin) __builtin_amdgcn_s_sleep(1);   // bounded
            if (!last) {
                const vfloat4 val = {x, x, x, x};
#pragma unroll
                for (int i = 0; i < 16; ++i) my[i * 64 + lane] = val;
            }
            if (lane == 0) tid[wave] = last ? -1 : tile;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag[wave] = 1;
            if (last) break;
        }
    } else {
        // ---- store wave s drains the slots of producers p = s, s + NS, ...
        const int s = wave - NP;
        int alive = 0;
        for (int p = s; p < NP; p += NS) ++alive;
        int p = s;
        for (int spin = 0; alive > 0 && spin < (1 << 26); ++spin) {   // bounded
            if (flag[p] == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int tile = tid[p];
                if (tile < 0) {
                    --alive;
                    flag[p] = 2;   // retired
                } else {
                    vfloat4 r[16];
                    const vfloat4* src = slots + (size_t)p * 1024;
#pragma unroll
                    for (int i = 0; i < 16; ++i) r[i] = src[i * 64 + lane];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // reads done before the slot is handed back
                    if (lane == 0) flag[p] = 0;
                    const int tx = tile % TX, t2 = tile / TX, ty = t2 % TY, plane = t2 / TY;
                    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
                    float* pl = dst + (size_t)plane * H * W;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = ty * 32 + sub * 16 + i;
                        if (row < H) st<AUX>(pl, ((size_t)row * W + col0) * 4, r[i]);
                    }
                }
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
            p += NS;
            if (p >= NP) p = s;
        }
    }
}

// reference for the same amount of dummy work: the plain one-wave-per-tile kernel with `work` FMAs in front of its stores
template <int AUX>
__global__ __launch_bounds__(64) void k_tile_work(float* dst, float v, int work)
{
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int tx = tile % TX, t2 = tile / TX, ty = t2 % TY, plane = t2 / TY;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const float x = burn(v, work);
    const vfloat4 val = {x, x, x, x};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = ty * 32 + sub * 16 + i;
        if (row < H) st<AUX>(p, ((size_t)row * W + col0) * 4, val);
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
void report(const char* name, float ms)
{
    printf("%-72s %8.4f ms  %8.1f GB/s\n", name, ms, N * 4.0 / ms / 1e6);
    fflush(stdout);
}

template <int NP, int NS, int AUX>
void run_pc(float* d, int wgs_per_cu)
{
    (void)hipFuncSetAttribute((const void*)k_pc<NP, NS, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int need = NP * 16384 + 2 * NP * 4 + 64;
    const int lds = std::max(need, (160 * 1024) / wgs_per_cu - 512);   // also pins the workgroups per CU
    if (lds > 160 * 1024) return;
    for (int work : {0, 400, 800}) {
        char name[160];
        snprintf(name, sizeof name, "producer/consumer: %2d compute + %d store waves, %d WG/CU, aux=%d, work %d", NP, NS, wgs_per_cu,
                 AUX, work);
        report(name, time_it([&] {
                   hipLaunchKernelGGL((k_pc<NP, NS, AUX>), dim3(256 * wgs_per_cu), dim3((NP + NS) * 64), lds, 0, d, 1.0f, work); }));
    }
}

int main()
{
    float* d;
    if (hipMalloc(&d, N * 4) != hipSuccess) return 1;
    for (int warm = 0; warm < 300; ++warm) hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipFuncSetAttribute((const void*)k_tile_static<18>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    report("static: 1 tile per WG, plain", time_it([&] { hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("static: 1 tile per WG, sc1+nt", time_it([&] { hipLaunchKernelGGL((k_tile_static<18>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("static: 1 tile per WG, sc1+nt, 2 waves/CU (LDS limited)", time_it([&] {
               hipLaunchKernelGGL((k_tile_static<18>), dim3(N_TILES), dim3(64), 80 * 1024 - 256, 0, d, 1.0f); }));
    for (int work : {0, 400, 800}) {
        char name[160];
        snprintf(name, sizeof name, "one wave per tile with %d FMAs before its stores, plain", work);
        report(name, time_it([&] { hipLaunchKernelGGL((k_tile_work<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f, work); }));
    }
    run_pc<8, 2, 18>(d, 1);
    run_pc<8, 1, 18>(d, 1);
    run_pc<8, 2, 0>(d, 1);
    run_pc<6, 2, 18>(d, 1);
    run_pc<4, 1, 18>(d, 2);
    run_pc<4, 1, 0>(d, 2);
    run_pc<4, 2, 18>(d, 2);
    run_pc<3, 1, 18>(d, 3);
    run_pc<2, 1, 18>(d, 4);
    // coverage
    (void)hipMemset(d, 0, N * 4);
    (void)hipFuncSetAttribute((const void*)k_pc<8, 2, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_pc<8, 2, 18>), dim3(256), dim3(640), 8 * 16384 + 128, 0, d, 1.0f, 0);
    (void)hipDeviceSynchronize();
    std::vector<float> host(N);
    (void)hipMemcpy(host.data(), d, N * 4, hipMemcpyDeviceToHost);
    size_t wrong = 0;
    for (size_t i = 0; i < N; ++i) wrong += host[i] != 1.0f;
    printf("producer/consumer coverage: %zu wrong elements\n", wrong);
    return 0;
}
