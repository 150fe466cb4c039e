// Eighth sweep (round 2): cheap dynamic balancing.  fill_patterns7 showed that one device-scope dequeue per tile is too
// expensive (a head word saturates at ~88 dequeues/us; a persistent wave that probes 8 empty shards at the end adds a
// storm of failed atomics; an atomic issued after the stores returns only when they have drained).  Here:
//   hybrid   persistent one-wave workgroups do a STATIC share of the tiles first (no atomics: wave w takes tiles
//            w, w + G, ...), and only the last (100 - S) % of the tiles are handed out dynamically from 8 per-XCC heads
//            (contiguous shards, stealing, CH tiles per dequeue).  Exhausted shards are detected with plain sc1 LOADS
//            of the heads (monotonic counters: a stale value can only cost one failed atomic), not with atomics.
//   stamps   per-XCC finish times of the best variant
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
    store_tile<AUX>(dst, blockIdx.x, threadIdx.x, v);
}

struct Stamp {
    unsigned long long t0, t2;
    unsigned xcc, tiles;
};

// counters: ctr[0] = done, ctr[32 * (1 + s)] = head of shard s (one 128-byte line each)
template <int AUX, int CH, bool STAMP>
__global__ __launch_bounds__(64) void k_hybrid(float* dst, float v, unsigned* ctr, int n_static, Stamp* stamps)
{
    extern __shared__ int dyn_lds[];
    unsigned long long t0 = 0;
    unsigned done_tiles = 0;
    if (STAMP) t0 = __builtin_amdgcn_s_memrealtime();
    const int home = __builtin_amdgcn_s_getreg(6164) & 7;
    // ---- static share
    for (int t = blockIdx.x; t < n_static; t += gridDim.x) {
        store_tile<AUX>(dst, t, threadIdx.x, v);
        ++done_tiles;
    }
    // ---- dynamic share: tiles [n_static, N_TILES) in 8 contiguous shards
    const unsigned n_dyn = (unsigned)(N_TILES - n_static);
    const unsigned per = (n_dyn + 7) / 8;
    const int lane = threadIdx.x;
    auto run_chunk = [&](int shard, unsigned t) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const unsigned idx = shard * per + t + c;
            if (t + c < per && idx < n_dyn) {
                store_tile<AUX>(dst, n_static + (int)idx, lane, v);
                ++done_tiles;
            }
        }
    };
    for (int rounds = 0; n_dyn > 0 && rounds < (1 << 20); ++rounds) {   // bounded
        // ONE wave-instruction reads all 8 heads (lane s -> head s; device-coherent sc1 load of a monotonic counter)
        unsigned head = per;
        if (lane < 8) head = __hip_atomic_load(&ctr[32 * (1 + lane)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned avail = (unsigned)(__ballot(head < per) & 0xffull);
        if (!avail) break;
        // first shard with work, starting at the home shard
        const unsigned rot = ((avail >> home) | (avail << (8 - home))) & 0xffu;
        const int shard = (home + __builtin_ctz(rot)) & 7;
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&ctr[32 * (1 + shard)], (unsigned)CH);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t < per) run_chunk(shard, t);
    }
    if (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{t0, t2, (unsigned)home, done_tiles};
    }
    if (threadIdx.x == 0) {
        const unsigned d = atomicAdd(&ctr[0], 1u);
        if (d == gridDim.x - 1) {
            for (int s = 0; s < 8; ++s) atomicExch(&ctr[32 * (1 + s)], 0u);
            atomicExch(&ctr[0], 0u);
        }
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

int main()
{
    float* d;
    unsigned* ctr;
    Stamp* ds;
    if (hipMalloc(&d, N * 4) != hipSuccess) return 1;
    if (hipMalloc(&ctr, 4096) != hipSuccess) return 1;
    if (hipMalloc(&ds, sizeof(Stamp) * 16384) != hipSuccess) return 1;
    (void)hipMemset(ctr, 0, 4096);
    for (int warm = 0; warm < 300; ++warm) hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f);
    (void)hipDeviceSynchronize();
    report("static: 1 tile per WG, plain", time_it([&] { hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));

#define SETLDS(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define RUN(AUX, CH)                                                                                              \
    {                                                                                                             \
        SETLDS((k_hybrid<AUX, CH, false>));                                                                       \
        const int waves_per_cu[] = {8, 16, 32};                                                                   \
        const int static_pct[] = {100, 90, 80, 70, 50, 0};                                                        \
        for (int wpc : waves_per_cu)                                                                              \
            for (int pct : static_pct) {                                                                          \
                const int lds = wpc >= 32 ? 0 : (160 * 1024) / wpc - 256;                                         \
                const int n_static = (int)((long long)N_TILES * pct / 100);                                       \
                char name[160];                                                                                   \
                snprintf(name, sizeof name, "hybrid aux=%d: %2d waves/CU, %3d %% static, %d tile(s)/dequeue", AUX, wpc, pct, CH); \
                report(name, time_it([&] {                                                                        \
                           hipLaunchKernelGGL((k_hybrid<AUX, CH, false>), dim3(256 * wpc), dim3(64), lds, 0, d, 1.0f, ctr, n_static, ds); })); \
            }                                                                                                     \
    }
    RUN(0, 1)
    RUN(0, 2)
    RUN(18, 1)

    // per-XCC finish times: static persistent vs 80 % static
    for (int pct : {100, 80}) {
        const int n_static = (int)((long long)N_TILES * pct / 100);
        const int G = 256 * 16;
        SETLDS((k_hybrid<0, 1, true>));
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL((k_hybrid<0, 1, true>), dim3(G), dim3(64), (160 * 1024) / 16 - 256, 0, d, 1.0f, ctr, n_static, ds);
        (void)hipDeviceSynchronize();
        std::vector<Stamp> hs(G);
        (void)hipMemcpy(hs.data(), ds, sizeof(Stamp) * G, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (auto& s : hs) {
            tmin = std::min(tmin, s.t0);
            tmax = std::max(tmax, s.t2);
        }
        double xend[8] = {0};
        long xt[8] = {0}, xw[8] = {0};
        for (auto& s : hs) {
            xend[s.xcc & 7] = std::max(xend[s.xcc & 7], (s.t2 - tmin) * 0.01);
            xt[s.xcc & 7] += s.tiles;
            xw[s.xcc & 7]++;
        }
        printf("stamps %3d %% static, 16 waves/CU: span %.1f us; per XCC [waves tiles last-end-us]:", pct, (tmax - tmin) * 0.01);
        for (int i = 0; i < 8; ++i) printf(" [%ld %ld %.1f]", xw[i], xt[i], xend[i]);
        printf("\n");
    }
    unsigned hc[1024];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hc, ctr, 4096, hipMemcpyDeviceToHost);
    unsigned bad = 0;
    for (int i = 0; i < 1024; ++i) bad |= hc[i];
    printf("counters after all launches: %s\n", bad ? "NOT ZERO" : "all zero (self reset works)");
    (void)hipMemset(d, 0, N * 4);
    hipLaunchKernelGGL((k_hybrid<0, 1, false>), dim3(256 * 16), dim3(64), 0, 0, d, 1.0f, ctr, N_TILES * 7 / 10, ds);
    (void)hipDeviceSynchronize();
    std::vector<float> host(N);
    (void)hipMemcpy(host.data(), d, N * 4, hipMemcpyDeviceToHost);
    size_t wrong = 0;
    for (size_t i = 0; i < N; ++i) wrong += host[i] != 1.0f;
    printf("hybrid coverage: %zu wrong elements\n", wrong);
    return 0;
}
