// Sixth sweep (round 2): WHY do waves that issue several stores plateau at 5.4-5.7 TB/s when single-store waves reach
// 6.8-7.3?  Round 1 ruled out tile shape, cache policy, stores in flight per WAVE and pacing.  This sweep varies what
// was not varied then:
//   occ     stores in flight per CU: tile waves (16 stores each) with the number of resident waves per CU limited by a
//           dynamic-LDS request (1..16 waves/CU) x a per-wave in-flight cap
//   ticket  16-wave workgroups, one 128x32 tile per wave, the STORE PHASES serialised through an LDS ticket semaphore
//           with K tokens (a wave stores only while holding a token and returns it after vmcnt(0))
//   stride  16 x 1 KB stores per wave at stride S (1 KB .. 8 MB): which address distance between a wave's consecutive
//           stores the memory system likes (DRAM page / channel interleave)
//   rot     tile waves that start at a per-tile rotated row, so concurrently running waves do not write the same row
//           residue at the same time
//   xcd     tile order remapped so that each XCD owns a contiguous range of tiles
//   stamps  s_memrealtime per wave (start / stores issued / stores drained) + XCC id: lifetime, drain time, per-XCD finish
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
constexpr int H = 1080, W = 1920, B = 64;
constexpr size_t N = (size_t)B * H * W;
constexpr size_t PLANE_BYTES = (size_t)H * W * 4;
constexpr int TX = W / 128, TY = (H + 31) / 32;  // 128 x 32 tiles: 15 x 34 per plane
constexpr int N_TILES = B * TX * TY;

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

template <int CAP>
__device__ __forceinline__ void cap_wait()
{
    if constexpr (CAP == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (CAP == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (CAP == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (CAP == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
}

// 16 stores of one 128 x 32 tile (lane l: 4 pixels of rows {i, i+16}), optionally starting at a rotated row
template <int AUX, int CAP>
__device__ __forceinline__ void store_tile(float* dst, int tile, int lane, float v, int rot)
{
    const int tx = tile % TX, t2 = tile / TX, ty = t2 % TY, plane = t2 / TY;
    const int sub = lane >> 5, col0 = tx * 128 + (lane & 31) * 4;
    const vfloat4 val = {v, v, v, v};
    float* p = dst + (size_t)plane * H * W;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = ty * 32 + sub * 16 + ((i + rot) & 15);
        if (row < H) st<AUX>(p, ((size_t)row * W + col0) * 4, val);
        cap_wait<CAP>();
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

// MODE 0: tile = blockIdx; 1: rotated start row; 2: XCD-contiguous tile order
template <int AUX, int CAP, int MODE>
__global__ __launch_bounds__(64) void k_tile(float* dst, float v)
{
    extern __shared__ int dyn_lds[];  // only its SIZE matters: it limits the workgroups (= waves) per CU
    int tile = blockIdx.x, rot = 0;
    if (MODE == 1) rot = (int)((blockIdx.x * 2654435761u) >> 28);
    if (MODE == 2) tile = (blockIdx.x & 7) * (N_TILES / 8) + (blockIdx.x >> 3);
    store_tile<AUX, CAP>(dst, tile, threadIdx.x, v, rot);
}

// NW waves per workgroup, one tile per wave; store phases gated by a counting semaphore with `tokens` tokens in LDS
template <int NW, int AUX>
__global__ __launch_bounds__(NW * 64) void k_ticket(float* dst, float v, int tokens)
{
    extern __shared__ int dyn_lds[];
    __shared__ int s_next, s_done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        s_next = 0;
        s_done = 0;
    }
    __syncthreads();
    const int tile = blockIdx.x * NW + wave;
    int my = 0;
    if (lane == 0) my = atomicAdd(&s_next, 1);
    my = __builtin_amdgcn_readfirstlane(my);
    // bounded spin: every earlier ticket holder finishes without waiting for anyone
    for (int spin = 0; spin < (1 << 22); ++spin) {
        const int done = __hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (done + tokens > my) break;
        __builtin_amdgcn_s_sleep(4);
    }
    if (tile < N_TILES) store_tile<AUX, 0>(dst, tile, lane, v, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(&s_done, 1);
}

// 16 x 1 KB stores per wave at a stride of S KB: super-block = 16*S KB, wave `off` of a super-block writes KB
// {off, off + S, ..., off + 15 S} of it
template <int AUX>
__global__ __launch_bounds__(64) void k_stride(float* dst, long long total_kb, int S, float v)
{
    const long long w = blockIdx.x;
    const long long super = w / S, off = w % S;
    const vfloat4 val = {v, v, v, v};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long long kb = super * 16 * S + (long long)j * S + off;
        if (kb < total_kb) {
            const size_t byte = (size_t)kb * 1024 + threadIdx.x * 16, plane = byte / PLANE_BYTES;
            st<AUX>(dst + plane * (PLANE_BYTES / 4), byte - plane * PLANE_BYTES, val);
        }
    }
}

// ---- time stamps
struct Stamp {
    unsigned long long t0, t1, t2;
    unsigned xcc, hwid;
};
__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(64) void k_tile_stamped(float* dst, float v, Stamp* out)
{
    const unsigned long long t0 = rt();
    store_tile<0, 0>(dst, blockIdx.x, threadIdx.x, v, 0);
    const unsigned long long t1 = rt();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = rt();
    if (threadIdx.x == 0)
        out[blockIdx.x] = Stamp{t0, t1, t2, (unsigned)__builtin_amdgcn_s_getreg(6164), (unsigned)__builtin_amdgcn_s_getreg(63492)};
}
__global__ __launch_bounds__(256) void k_linear_stamped(float* dst, size_t n4, float v, Stamp* out)
{
    const unsigned long long t0 = rt();
    const vfloat4 val = {v, v, v, v};
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) reinterpret_cast<vfloat4*>(dst)[i] = val;
    const unsigned long long t1 = rt();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = rt();
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] =
            Stamp{t0, t1, t2, (unsigned)__builtin_amdgcn_s_getreg(6164), (unsigned)__builtin_amdgcn_s_getreg(63492)};
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
    printf("%-60s %8.4f ms  %8.1f GB/s\n", name, ms, N * 4.0 / ms / 1e6);
    fflush(stdout);
}

void stamp_report(const char* name, std::vector<Stamp>& s)
{
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto& x : s) {
        tmin = std::min(tmin, x.t0);
        tmax = std::max(tmax, x.t2);
    }
    std::vector<double> life, issue, drain;
    double xend[8] = {0}, xstart_last[8] = {0};
    long xcnt[8] = {0};
    for (auto& x : s) {
        life.push_back((x.t2 - x.t0) * 0.01);
        issue.push_back((x.t1 - x.t0) * 0.01);
        drain.push_back((x.t2 - x.t1) * 0.01);
        const int xc = x.xcc & 7;
        xend[xc] = std::max(xend[xc], (x.t2 - tmin) * 0.01);
        xstart_last[xc] = std::max(xstart_last[xc], (x.t0 - tmin) * 0.01);
        xcnt[xc]++;
    }
    auto pct = [](std::vector<double>& v, double p) {
        std::sort(v.begin(), v.end());
        return v[(size_t)(p * (v.size() - 1))];
    };
    printf("%s: span %.1f us; wave life p10/p50/p90/max %.2f/%.2f/%.2f/%.2f us; issue p50/p90 %.2f/%.2f; drain p50/p90 %.2f/%.2f\n",
           name, (tmax - tmin) * 0.01, pct(life, .1), pct(life, .5), pct(life, .9), pct(life, 1.0), pct(issue, .5),
           pct(issue, .9), pct(drain, .5), pct(drain, .9));
    printf("   per XCC: waves / last start / last end (us):");
    for (int i = 0; i < 8; ++i) printf(" [%ld %.1f %.1f]", xcnt[i], xstart_last[i], xend[i]);
    printf("\n");
    // concurrency profile: resident waves at 10 sample times
    const double span = (tmax - tmin) * 0.01;
    printf("   resident waves at 10%%..90%% of the span:");
    for (int q = 1; q <= 9; ++q) {
        const unsigned long long t = tmin + (unsigned long long)((span * q / 10.0) * 100.0);
        long c = 0;
        for (auto& x : s) c += (x.t0 <= t && x.t2 > t);
        printf(" %ld", c);
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    float* d;
    if (hipMalloc(&d, N * 4 + (64 << 20)) != hipSuccess) return 1;
    const size_t n4 = N / 4;
    (void)hipFuncSetAttribute((const void*)k_tile<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int warm = 0; warm < 300; ++warm)
        hipLaunchKernelGGL((k_linear<0>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4, 1.0f);
    (void)hipDeviceSynchronize();

    report("linear 1 store/wave plain", time_it([&] {
               hipLaunchKernelGGL((k_linear<0>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4, 1.0f); }));
    report("linear 1 store/wave sc1+nt", time_it([&] {
               hipLaunchKernelGGL((k_linear<18>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4, 1.0f); }));
    report("tile 128x32 plain", time_it([&] { hipLaunchKernelGGL((k_tile<0, 0, 0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("tile 128x32 sc1+nt", time_it([&] { hipLaunchKernelGGL((k_tile<18, 0, 0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("tile 128x32 plain, rotated start row", time_it([&] { hipLaunchKernelGGL((k_tile<0, 0, 1>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("tile 128x32 sc1+nt, rotated start row", time_it([&] { hipLaunchKernelGGL((k_tile<18, 0, 1>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("tile 128x32 plain, XCD-contiguous order", time_it([&] { hipLaunchKernelGGL((k_tile<0, 0, 2>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("tile 128x32 sc1+nt, XCD-contiguous order", time_it([&] { hipLaunchKernelGGL((k_tile<18, 0, 2>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));

    // ---- occupancy x per-wave cap: stores in flight per CU = waves/CU x cap KB
#define SETLDS(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define OCC(AUX, CAP)                                                                                          \
    {                                                                                                          \
        SETLDS((k_tile<AUX, CAP, 0>));                                                                         \
        const int waves_per_cu[] = {1, 2, 4, 8, 16};                                                           \
        for (int wpc : waves_per_cu) {                                                                         \
            const int lds = (160 * 1024) / wpc - (wpc > 1 ? 256 : 0);                                          \
            char name[128];                                                                                    \
            snprintf(name, sizeof name, "occ: tile aux=%d, %2d waves/CU, cap %2d stores/wave in flight", AUX, wpc, \
                     CAP ? CAP : 16);                                                                          \
            report(name, time_it([&] {                                                                         \
                       hipLaunchKernelGGL((k_tile<AUX, CAP, 0>), dim3(N_TILES), dim3(64), lds, 0, d, 1.0f); })); \
        }                                                                                                      \
    }
    OCC(0, 0) OCC(0, 4) OCC(0, 1)
    if (!quick) { OCC(18, 0) OCC(18, 4) OCC(18, 1) OCC(0, 8) OCC(0, 2) }

    // ---- ticketed store phases
#define TICKET(NW, AUX, WGS_PER_CU)                                                                            \
    {                                                                                                          \
        SETLDS((k_ticket<NW, AUX>));                                                                           \
        const int toks[] = {1, 2, 4, 8, NW};                                                                   \
        for (int tk : toks) {                                                                                  \
            if (tk > NW) continue;                                                                             \
            const int lds = (160 * 1024) / WGS_PER_CU - 512;                                                   \
            char name[128];                                                                                    \
            snprintf(name, sizeof name, "ticket: %2d-wave WGs aux=%d, %d WG/CU, %2d tokens", NW, AUX, WGS_PER_CU, tk); \
            report(name, time_it([&] {                                                                         \
                       hipLaunchKernelGGL((k_ticket<NW, AUX>), dim3((N_TILES + NW - 1) / NW), dim3(NW * 64), lds, 0, d, 1.0f, tk); })); \
        }                                                                                                      \
    }
    TICKET(16, 0, 1) TICKET(16, 0, 2) TICKET(8, 0, 2) TICKET(8, 0, 4) TICKET(4, 0, 4)
    if (!quick) { TICKET(16, 18, 1) TICKET(16, 18, 2) TICKET(4, 0, 8) }

    // ---- stride sweep
    {
        const long long total_kb = (long long)(N * 4 / 1024);
        for (int S = 1; S <= 8192; S *= 2) {
            const long long supers = (total_kb + 16ll * S - 1) / (16ll * S);
            char name[128];
            snprintf(name, sizeof name, "stride: 16 x 1 KB per wave, stride %5d KB plain", S);
            report(name, time_it([&] {
                       hipLaunchKernelGGL((k_stride<0>), dim3((unsigned)(supers * S)), dim3(64), 0, 0, d, total_kb, S, 1.0f); }));
        }
        if (!quick)
            for (int S = 1; S <= 8192; S *= 4) {
                const long long supers = (total_kb + 16ll * S - 1) / (16ll * S);
                char name[128];
                snprintf(name, sizeof name, "stride: 16 x 1 KB per wave, stride %5d KB sc1+nt", S);
                report(name, time_it([&] {
                           hipLaunchKernelGGL((k_stride<18>), dim3((unsigned)(supers * S)), dim3(64), 0, 0, d, total_kb, S, 1.0f); }));
            }
    }

    // ---- stamps
    {
        Stamp* ds;
        (void)hipMalloc(&ds, sizeof(Stamp) * (n4 / 64 + 4096));
        std::vector<Stamp> hs(N_TILES);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_tile_stamped, dim3(N_TILES), dim3(64), 0, 0, d, 1.0f, ds);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hs.data(), ds, sizeof(Stamp) * N_TILES, hipMemcpyDeviceToHost);
        stamp_report("stamps tile 128x32 plain (20th back-to-back launch)", hs);
        const unsigned blocks = (unsigned)((n4 + 255) / 256);
        std::vector<Stamp> hl((size_t)blocks * 4);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_linear_stamped, dim3(blocks), dim3(256), 0, 0, d, n4, 1.0f, ds);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hl.data(), ds, sizeof(Stamp) * hl.size(), hipMemcpyDeviceToHost);
        stamp_report("stamps linear 1 store/wave plain (20th back-to-back launch)", hl);
        (void)hipFree(ds);
    }
    report("linear 1 store/wave plain (again)", time_it([&] {
               hipLaunchKernelGGL((k_linear<0>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, d, n4, 1.0f); }));
    (void)hipFree(d);
    return 0;
}
