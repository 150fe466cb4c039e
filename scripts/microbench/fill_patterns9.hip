// Ninth sweep (round 2): XCD-WEIGHTED static scheduling.  The XCDs of an MI355X do not write at the same rate (fill_patterns6/8:
// the even XCC ids finish an equal share 15-20 % earlier than the odd ones, also when the others have gone idle), and
// dequeuing tiles with device-scope atomics under a saturated store stream is far too slow (fill_patterns7/8).  Here each
// persistent wave registers ONCE (one returning atomic on its XCC's counter, before it has stores in flight) to learn its
// rank among the waves of its XCC, and shard x (a contiguous range of tiles proportional to weight[x]) is split statically
// over the ranks.  The host closes the loop over a few launches: weight[x] *= (mean finish / finish[x]).
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

struct Params {
    int lo[9];        // shard x owns tiles [lo[x], lo[x+1])
    int per_xcc;      // expected waves per XCC (grid / 8)
};
// ws: [0] done, [32 * (1 + x)] registration counter of XCC x, [32 * 9 + 2 * x] (u64) last finish of XCC x, [32*9+16] (u64) first start
template <int AUX, bool INTERLEAVE>
__global__ __launch_bounds__(64) void k_weighted(float* dst, float v, unsigned* ws, Params p)
{
    extern __shared__ int dyn_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int x = __builtin_amdgcn_s_getreg(6164) & 7;
    unsigned r = 0;
    if (threadIdx.x == 0) r = atomicAdd(&ws[32 * (1 + x)], 1u);
    r = __builtin_amdgcn_readfirstlane(r);
    r %= (unsigned)p.per_xcc;   // surplus waves repeat a series (idempotent); missing ranks are repaired by the last wave
    const int lo = p.lo[x], hi = p.lo[x + 1];
    if (INTERLEAVE) {
        for (int t = lo + (int)r; t < hi; t += p.per_xcc) store_tile<AUX>(dst, t, threadIdx.x, v);
    } else {  // contiguous block per wave
        const int n = hi - lo, c = (n + p.per_xcc - 1) / p.per_xcc;
        for (int t = lo + (int)r * c; t < min(hi, lo + ((int)r + 1) * c); ++t) store_tile<AUX>(dst, t, threadIdx.x, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned long long* tw = reinterpret_cast<unsigned long long*>(ws + 32 * 9);
        atomicMax(&tw[x], t2);          // no-return
        atomicMin(&tw[8], t0);
        const unsigned d = atomicAdd(&ws[0], 1u);
        if (d == gridDim.x - 1) {
            // (the product kernel repairs missing ranks here; the microbenchmark only reports them)
            for (int s = 0; s < 8; ++s) {
                const unsigned c = atomicExch(&ws[32 * (1 + s)], 0u);
                ws[32 * 12 + s] = c;   // registration counts for the host
            }
            atomicExch(&ws[0], 0u);
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

Params make_params(const double* w, int grid)
{
    Params p;
    double sum = 0;
    for (int i = 0; i < 8; ++i) sum += w[i];
    double acc = 0;
    p.lo[0] = 0;
    for (int i = 0; i < 8; ++i) {
        acc += w[i];
        p.lo[i + 1] = (int)(N_TILES * (acc / sum) + 0.5);
    }
    p.lo[8] = N_TILES;
    p.per_xcc = grid / 8;
    return p;
}

template <int AUX, bool INTERLEAVE>
void closed_loop(float* d, unsigned* ws, int wpc, const char* label)
{
    const int grid = 256 * wpc;
    const int lds = wpc >= 32 ? 0 : (160 * 1024) / wpc - 256;
    (void)hipFuncSetAttribute((const void*)k_weighted<AUX, INTERLEAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    double w[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < 6; ++it) {
        Params p = make_params(w, grid);
        char name[200];
        snprintf(name, sizeof name, "%s %2d waves/CU iter %d  w = %.3f %.3f %.3f %.3f %.3f %.3f %.3f %.3f", label, wpc, it, w[0], w[1],
                 w[2], w[3], w[4], w[5], w[6], w[7]);
        report(name, time_it([&] { hipLaunchKernelGGL((k_weighted<AUX, INTERLEAVE>), dim3(grid), dim3(64), lds, 0, d, 1.0f, ws, p); }));
        // one stamped launch in steady state: reset the time words, 20 launches, read back
        (void)hipDeviceSynchronize();
        for (int i = 0; i < 20; ++i) {
            if (i == 19) {
                (void)hipDeviceSynchronize();
                unsigned long long init[9];
                for (int k = 0; k < 8; ++k) init[k] = 0;
                init[8] = ~0ull;
                (void)hipMemcpy(ws + 32 * 9, init, sizeof init, hipMemcpyHostToDevice);
            }
            hipLaunchKernelGGL((k_weighted<AUX, INTERLEAVE>), dim3(grid), dim3(64), lds, 0, d, 1.0f, ws, p);
        }
        (void)hipDeviceSynchronize();
        unsigned long long tw[9];
        unsigned counts[8];
        (void)hipMemcpy(tw, ws + 32 * 9, sizeof tw, hipMemcpyDeviceToHost);
        (void)hipMemcpy(counts, ws + 32 * 12, sizeof counts, hipMemcpyDeviceToHost);
        double fin[8], mean = 0;
        for (int k = 0; k < 8; ++k) {
            fin[k] = (tw[k] - tw[8]) * 0.01;
            mean += fin[k] / 8;
        }
        printf("      finish per XCC (us, isolated launch):");
        for (int k = 0; k < 8; ++k) printf(" %.1f", fin[k]);
        printf("   registered waves:");
        for (int k = 0; k < 8; ++k) printf(" %u", counts[k]);
        printf("\n");
        for (int k = 0; k < 8; ++k) w[k] *= mean / fin[k];
    }
}

int main()
{
    float* d;
    unsigned* ws;
    if (hipMalloc(&d, N * 4) != hipSuccess) return 1;
    if (hipMalloc(&ws, 4096) != hipSuccess) return 1;
    (void)hipMemset(ws, 0, 4096);
    for (int warm = 0; warm < 300; ++warm) hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f);
    (void)hipDeviceSynchronize();
    report("static: 1 tile per WG, plain", time_it([&] { hipLaunchKernelGGL((k_tile_static<0>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    report("static: 1 tile per WG, sc1+nt", time_it([&] { hipLaunchKernelGGL((k_tile_static<18>), dim3(N_TILES), dim3(64), 0, 0, d, 1.0f); }));
    closed_loop<0, true>(d, ws, 16, "weighted interleaved plain");
    closed_loop<0, false>(d, ws, 16, "weighted contiguous  plain");
    closed_loop<0, true>(d, ws, 8, "weighted interleaved plain");
    closed_loop<18, true>(d, ws, 16, "weighted interleaved sc1+nt");
    closed_loop<18, true>(d, ws, 8, "weighted interleaved sc1+nt");
    closed_loop<0, true>(d, ws, 32, "weighted interleaved plain");
    // coverage
    (void)hipMemset(d, 0, N * 4);
    double w[8] = {1.3, 0.7, 1.1, 0.9, 1.2, 0.8, 1.0, 1.0};
    Params p = make_params(w, 256 * 16);
    hipLaunchKernelGGL((k_weighted<0, true>), dim3(256 * 16), dim3(64), 0, 0, d, 1.0f, ws, p);
    (void)hipDeviceSynchronize();
    std::vector<float> host(N);
    (void)hipMemcpy(host.data(), d, N * 4, hipMemcpyDeviceToHost);
    size_t wrong = 0;
    for (size_t i = 0; i < N; ++i) wrong += host[i] != 1.0f;
    printf("weighted coverage: %zu wrong elements\n", wrong);
    return 0;
}
