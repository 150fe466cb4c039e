// Seventh sweep (round 2).  fill_patterns6 showed with per-wave time stamps that the 8 XCDs do NOT write at the same
// rate: with the dispatcher's static round-robin (workgroup b -> XCD b % 8, equal work per XCD) the even XCCs finished
// their 4080 tiles after 73-79 us, the odd ones after 88-94 us, in the tile kernel and in the single-store kernel
// alike, so every launch ends with a tail in which half of the chip idles.  This sweep measures what DYNAMIC tile
// scheduling buys: persistent one-wave workgroups pull 128x32 tiles from device-scope atomic heads.
//   dyn1     one global head, CH consecutive tiles per dequeue
//   dyn8     eight heads, one per XCC (HW_REG_XCC_ID), each owning a contiguous eighth of the tiles; a wave whose shard
//            is empty steals from the next shards
//   dyn8i    the same with interleaved ownership (tile t belongs to shard t % 8)
// all with the last-wave-out self reset of the counters (no memset between launches).
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
constexpr int N_TILES = B * TX * TY;  // 32640 = 8 * 4080

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
__device__ __forceinline__ void store_tile(float* dst, int tile, int lane, float v)
{
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

template <int AUX>
__global__ __launch_bounds__(64) void k_tile_static(float* dst, float v)
{
    extern __shared__ int dyn_lds[];
    store_tile<AUX>(dst, blockIdx.x, threadIdx.x, v);
}

// counters: ctr[0] = done, ctr[32 * (1 + s)] = head of shard s (one 128-byte line each)
__device__ __forceinline__ void finish(unsigned* ctr, int n_heads)
{
    if (threadIdx.x == 0) {
        const unsigned d = atomicAdd(&ctr[0], 1u);
        if (d == gridDim.x - 1) {  // every other wave has made its last access to the heads
            for (int s = 0; s < n_heads; ++s) atomicExch(&ctr[32 * (1 + s)], 0u);
            atomicExch(&ctr[0], 0u);
        }
    }
}

template <int AUX, int CH>
__global__ __launch_bounds__(64) void k_dyn1(float* dst, float v, unsigned* ctr)
{
    extern __shared__ int dyn_lds[];
    for (;;) {
        unsigned t = 0;
        if (threadIdx.x == 0) t = atomicAdd(&ctr[32], (unsigned)CH);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= (unsigned)N_TILES) break;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (t + c < (unsigned)N_TILES) store_tile<AUX>(dst, (int)(t + c), threadIdx.x, v);
    }
    finish(ctr, 1);
}

// INTERLEAVED = false: shard s owns tiles [s * N/8, (s+1) * N/8); true: tiles t with t % 8 == s
template <int AUX, bool INTERLEAVED, int CH>
__global__ __launch_bounds__(64) void k_dyn8(float* dst, float v, unsigned* ctr)
{
    extern __shared__ int dyn_lds[];
    constexpr unsigned PER = N_TILES / 8;
    const int home = __builtin_amdgcn_s_getreg(6164) & 7;  // HW_REG_XCC_ID[3:0]
    int shard = home, tried = 0;
    while (tried < 8) {
        unsigned t = 0;
        if (threadIdx.x == 0) t = atomicAdd(&ctr[32 * (1 + shard)], (unsigned)CH);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= PER) {  // shard exhausted: steal from the next one
            shard = (shard + 1) & 7;
            ++tried;
            continue;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (t + c >= PER) break;
            const int tile = INTERLEAVED ? (int)((t + c) * 8 + shard) : (int)(shard * PER + t + c);
            store_tile<AUX>(dst, tile, threadIdx.x, v);
        }
    }
    finish(ctr, 8);
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
    printf("%-64s %8.4f ms  %8.1f GB/s\n", name, ms, N * 4.0 / ms / 1e6);
    fflush(stdout);
}

int main()
{
    float* d;
    unsigned* ctr;
    if (hipMalloc(&d, N * 4) != hipSuccess) return 1;
    if (hipMalloc(&ctr, 4096) != hipSuccess) return 1;
    (void)hipMemset(ctr, 0, 4096);
    for (int warm = 0; warm < 300; ++warm) hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f);
    (void)hipDeviceSynchronize();
    report("static: 1 tile per WG, plain", time_it([&] { hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("static: 1 tile per WG, sc1+nt", time_it([&] { hipLaunchKernelGGL((k_tile_static<18>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));

#define SETLDS(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define RUN(KERNEL, LABEL)                                                                               \
    {                                                                                                    \
        SETLDS((KERNEL));                                                                                \
        const int waves_per_cu[] = {2, 4, 8, 16, 32};                                                    \
        for (int wpc : waves_per_cu) {                                                                   \
            const int lds = wpc >= 32 ? 0 : (160 * 1024) / wpc - 256;                                    \
            char name[160];                                                                              \
            snprintf(name, sizeof name, "%s, %2d persistent waves/CU", LABEL, wpc);                      \
            report(name, time_it([&] {                                                                   \
                       hipLaunchKernelGGL((KERNEL), dim3(256 * wpc), dim3(64), lds, 0, d, 1.0f, ctr); })); \
        }                                                                                                \
    }
    RUN((k_dyn8<0, false, 1>), "dyn8 contiguous shards, plain")
    RUN((k_dyn8<0, true, 1>), "dyn8 interleaved shards, plain")
    RUN((k_dyn1<0, 1>), "dyn1 one head, 1 tile/dequeue, plain")
    RUN((k_dyn1<0, 4>), "dyn1 one head, 4 tiles/dequeue, plain")
    RUN((k_dyn8<18, false, 1>), "dyn8 contiguous shards, sc1+nt")
    RUN((k_dyn8<18, true, 1>), "dyn8 interleaved shards, sc1+nt")
    RUN((k_dyn8<0, false, 2>), "dyn8 contiguous shards, 2 tiles/dequeue, plain")
    // counters must be back at zero
    unsigned hc[1024];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hc, ctr, 4096, hipMemcpyDeviceToHost);
    unsigned bad = 0;
    for (int i = 0; i < 1024; ++i) bad |= hc[i];
    printf("counters after all launches: %s\n", bad ? "NOT ZERO" : "all zero (self reset works)");
    // coverage check of the dynamic kernels: fill with 2, then overwrite with 1 by dyn8, expect all 1
    (void)hipMemset(d, 0, N * 4);
    hipLaunchKernelGGL((k_dyn8<0, false, 1>), dim3(256 * 16), dim3(64), 0, 0, d, 1.0f, ctr);
    (void)hipDeviceSynchronize();
    std::vector<float> host(N);
    (void)hipMemcpy(host.data(), d, N * 4, hipMemcpyDeviceToHost);
    size_t wrong = 0;
    for (size_t i = 0; i < N; ++i) wrong += host[i] != 1.0f;
    printf("dyn8 coverage: %zu wrong elements\n", wrong);
    (void)hipFree(d);
    return 0;
}
