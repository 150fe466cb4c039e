// lane_helpers — batched polyline arc-length interpolation and lengths for gfx950.
//
// Replaces packages/lane_helpers/ext_impl/polyline/include/polyline_kernels.cuh:390-455 (four kernels) and the host
// launch logic of src/polyline.cu:59-129; edge semantics follow include/polyline_common.cuh:58-163:
//   * accum[i] = distance from the first point to point i; query d (x total length when `relative`);
//   * d before the first point -> first point, d beyond the last -> last point;
//   * otherwise the segment [i, i+1] with accum[i] <= d is interpolated linearly; a segment shorter than
//     epsilon(accumulation type) returns its lower point;
//   * a polyline with zero points yields NaN for every query; lengths: 0 points -> NaN, 1 point -> 0.
// Design (not a translation): ONE workgroup of 256 threads per polyline; segment lengths are written to LDS by
// contiguous per-thread chunks, the 256 chunk totals are scanned by one 64-lane wave with __shfl_up (the reference
// hard-codes 32-lane shuffles, polyline_kernels.cuh:26-35,58-70), then every thread binary-searches its queries in
// LDS.  Accumulation is fp32 for f32/f16/bf16 storage and fp64 for f64 (the reference GPU path accumulates in the
// storage dtype, polyline_common.cuh:56-58).  Polylines longer than the LDS budget use a caller-provided global
// scratch row instead (same code path, pointer swap).
#include <hip/hip_runtime.h>

#include <climits>

#include <cmath>
#include <limits>
#include <thread>
#include <vector>

#include <algorithm>
#include <cstdint>
#include <limits>

#include "accv_common.h"
#include "polyline_arith.h"

namespace {

constexpr int kThreads = 256;
constexpr int kLdsBudgetBytes = 48 * 1024;

enum PolyType { kPF32 = 0, kPF64 = 1, kPF16 = 2, kPBF16 = 3 };

template <int TY>
struct Storage;
template <>
struct Storage<kPF32> {
    using T = float;
    using Acc = float;
    static __device__ __forceinline__ Acc load(const T* p) { return *p; }
    static __device__ __forceinline__ void store(T* p, Acc v) { *p = v; }
    static __device__ __forceinline__ void copy(T* d, const T* s) { *d = *s; }
};
template <>
struct Storage<kPF64> {
    using T = double;
    using Acc = double;
    static __device__ __forceinline__ Acc load(const T* p) { return *p; }
    static __device__ __forceinline__ void store(T* p, Acc v) { *p = v; }
    static __device__ __forceinline__ void copy(T* d, const T* s) { *d = *s; }
};
template <>
struct Storage<kPF16> {
    using T = uint16_t;
    using Acc = float;
    static __device__ __forceinline__ Acc load(const T* p) { return (float)(*reinterpret_cast<const _Float16*>(p)); }
    static __device__ __forceinline__ void store(T* p, Acc v)
    {
        _Float16 h = (_Float16)v;
        *p = *reinterpret_cast<uint16_t*>(&h);
    }
    static __device__ __forceinline__ void copy(T* d, const T* s) { *d = *s; }
};
template <>
struct Storage<kPBF16> {
    using T = uint16_t;
    using Acc = float;
    static __device__ __forceinline__ Acc load(const T* p) { return __uint_as_float((uint32_t)(*p) << 16); }
    static __device__ __forceinline__ void store(T* p, Acc v)
    {
        uint32_t u = __float_as_uint(v);
        *p = ((u & 0x7fffffffu) > 0x7f800000u) ? (uint16_t)((u >> 16) | 0x40u)
                                               : (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    static __device__ __forceinline__ void copy(T* d, const T* s) { *d = *s; }
};

struct PolyParams {
    const void* points;     // [B, P, D]
    const void* distances;  // [B, Q] or null (lengths only)
    const void* point_counts;  // null = all P valid
    const void* dist_counts;   // null = all Q valid
    void* out_points;       // [B, Q, D] or null
    void* out_lengths;      // [B] or null
    void* scratch;          // [B, P] accumulation-type elements, used when P does not fit LDS
    float* out_boxes;       // optional (f32, D == 2): [B, ceil(Q/64), 4] bounding box of every 64 consecutive samples
    long long batch;
    int P, Q, D;
    int counts_i64, relative, use_scratch;
    int q_chunk;            // queries per workgroup: blockIdx.y walks the chunks of one polyline (multiple of kThreads)
};

__device__ __forceinline__ long long load_count(const void* p, long long i, int is64)
{
    return is64 ? static_cast<const long long*>(p)[i] : (long long)static_cast<const int*>(p)[i];
}

// THREADS = workgroup size: 256, or 1024 for long polylines (the scan over the points is the serial part of a workgroup)
// DC = number of coordinates per point when it is 2 or 3 (compile-time), 0 = p.D at run time.  With a run-time D every loop
// over the coordinates is a real loop with its loads and a wait inside: a segment's length cost D dependent round trips to
// memory, and the four segments a thread takes per trip cost 4 D of them one after the other (5000 2-D points on 1024 threads:
// 16 round trips in front of the scan; ISA of round 2's kernel) — with DC the loops unroll and the loads of a trip are in flight
// together, which is what the four-segments-per-trip layout was written for
template <int TY, int THREADS, int DC>
__global__ __launch_bounds__(THREADS) void polyline_kernel(const PolyParams p)
{
    const int D = DC > 0 ? DC : p.D;
    constexpr int kPerLane = THREADS / 64;   // chunk totals per lane in the scan by wave 0
    using S = Storage<TY>;
    using T = typename S::T;
    using Acc = typename S::Acc;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ Acc s_part[THREADS];

    const long long b = blockIdx.x;
    const int t = threadIdx.x;
    int n = p.P, q = p.Q;
    if (p.point_counts) n = (int)max(0ll, min((long long)p.P, load_count(p.point_counts, b, p.counts_i64)));
    if (p.dist_counts) q = (int)max(0ll, min((long long)p.Q, load_count(p.dist_counts, b, p.counts_i64)));
    const T* pts = static_cast<const T*>(p.points) + (size_t)b * p.P * D;
    T* out = p.out_points ? static_cast<T*>(p.out_points) + (size_t)b * p.Q * D : nullptr;
    const Acc nan = std::numeric_limits<Acc>::quiet_NaN();

    // this workgroup's share of the queries: many queries of few polylines are spread over several workgroups (each repeats
    // the prefix scan, which is cheap next to thousands of binary searches by one workgroup while most of the chip idles)
    const int q_begin = (int)blockIdx.y * p.q_chunk;
    const int q_stop = min(q, q_begin + p.q_chunk), q_cap = min(p.Q, q_begin + p.q_chunk);
    const bool first_chunk = blockIdx.y == 0;

    if (n == 0) {  // undefined polyline: NaN everywhere (polyline_kernels.cuh:216-225)
        if (out)
            for (int i = q_begin * D + t; i < q_stop * D; i += THREADS) S::store(out + i, nan);
        if (p.out_lengths && t == 0 && first_chunk) S::store(static_cast<T*>(p.out_lengths) + b, nan);
        if (p.out_boxes) {
            const int groups = (p.Q + 63) / 64;
            const float inf = __builtin_inff();
            for (int g = (q_begin >> 6) + t; g < min(groups, (q_cap + 63) >> 6); g += THREADS)
                reinterpret_cast<float4*>(p.out_boxes)[b * groups + g] = make_float4(inf, inf, -inf, -inf);
        }
        return;
    }

    Acc* accum = p.use_scratch ? static_cast<Acc*>(p.scratch) + (size_t)b * p.P : reinterpret_cast<Acc*>(lds_raw);
    // the thread's first query distance does not depend on the scan: requested here, ahead of the four barriers, instead of as a
    // round trip of its own behind them
    const T* dist = static_cast<const T*>(p.distances) + (size_t)b * p.Q;
    Acc first_d = 0;
    if (out && q_begin + t < q_stop) first_d = S::load(dist + q_begin + t);

    // ---- segment lengths, chunked: thread t owns segments [lo, hi), writes the chunk-local inclusive prefix
    const int n_seg = n - 1;
    const int per = (n_seg + THREADS - 1) / THREADS;
    const int lo = min(t * per, n_seg), hi = min(lo + per, n_seg);
    // (lengths first, with consecutive threads on consecutive segments — coalesced reads of the points; a thread walking its
    // own chunk of the points made every wave load touch 64 different cache lines: 5000 points took 15 us)
    // four segments per thread and trip, their loads issued together: the loop is a chain of memory round trips otherwise
    // (20 trips of ~0.5 us for 5000 points)
    constexpr int kBatch = 4;
    for (int s0 = t; s0 < n_seg; s0 += kBatch * THREADS) {
        Acc sq[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int s = min(s0 + u * THREADS, n_seg - 1);      // clamped: loads stay in range, surplus results are dropped
            Acc acc2 = 0;
            for (int d = 0; d < D; ++d) {
                const Acc diff = S::load(pts + (size_t)s * D + d) - S::load(pts + (size_t)(s + 1) * D + d);
                if constexpr (TY == kPF32)
                    acc2 = accv_poly::seg_length2_step(acc2, diff);   // (pinned: the fused lane raster repeats it, polyline_arith.h)
                else
                    acc2 += diff * diff;
            }
            sq[u] = acc2;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int s = s0 + u * THREADS;
            if (s < n_seg) accum[s + 1] = sqrt(sq[u]);
        }
    }
    if (p.use_scratch) __threadfence_block();
    __syncthreads();
    Acc run = 0;
    for (int s = lo; s < hi; ++s) {     // same summation order as before: chunk-local prefix, then the chunk offsets
        run += accum[s + 1];
        accum[s + 1] = run;
    }
    if (t == 0) accum[0] = 0;
    s_part[t] = run;
    __syncthreads();
    // ---- exclusive scan of the THREADS chunk totals by wave 0 (kPerLane per lane + 64-lane shuffle scan)
    if (t < 64) {
        Acc v[kPerLane];
        Acc lane_total = 0;
#pragma unroll
        for (int u = 0; u < kPerLane; ++u) {
            v[u] = s_part[kPerLane * t + u];
            lane_total += v[u];
        }
        Acc incl = lane_total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const Acc up = __shfl_up(incl, off, 64);
            if (t >= off) incl += up;
        }
        Acc run2 = incl - lane_total;
#pragma unroll
        for (int u = 0; u < kPerLane; ++u) {
            s_part[kPerLane * t + u] = run2;
            run2 += v[u];
        }
    }
    __syncthreads();
    const Acc base = s_part[t];
    if (base != 0)
        for (int s = lo; s < hi; ++s) accum[s + 1] += base;
    if (p.use_scratch) __threadfence_block();
    __syncthreads();

    const Acc total = accum[n - 1];
    if (p.out_lengths && t == 0 && first_chunk) S::store(static_cast<T*>(p.out_lengths) + b, total);
    if (!out) return;

    // ---- queries
    const Acc eps = std::numeric_limits<Acc>::epsilon();
    // (whole waves walk the loop together: when group boxes are wanted, the 64 lanes of a wave hold 64 consecutive samples
    // and reduce their bounding box with shuffles)
    const int q_span = p.out_boxes ? q_begin + ((q_cap - q_begin + THREADS - 1) / THREADS) * THREADS : q_stop;
    const int groups = (p.Q + 63) / 64;
    for (int i = q_begin + t; i < q_span; i += THREADS) {
        float bx = __builtin_nanf(""), by = bx;  // sample coordinates for the group box (D == 2, f32 instantiation)
        if (i < q_stop) {
            Acc d = i == q_begin + t ? first_d : S::load(dist + i);
            if (p.relative) {
                if constexpr (TY == kPF32) d = accv_poly::scale_query(d, total);
                else d *= total;
            }
            T* res = out + (size_t)i * D;
            // last index whose accumulated distance is <= d (polyline_common.cuh:89-116)
            int idx;
            if (accum[0] > d) {
                idx = -1;
            } else if (accum[n - 1] < d) {
                idx = n - 1;
            } else {
                int mn = 0, mx = n - 1;
                while (mx - mn > 1) {
                    const int c = (mx + mn) >> 1;
                    const Acc v = accum[c];
                    if (v < d) mn = c;
                    else if (v > d) mx = c;
                    else mn = mx = c;
                }
                idx = mn;
            }
            if constexpr (DC > 0) {
                // every coordinate is READ before the first one is written: as far as the compiler knows the output may alias the
                // points, so a write between two reads made every coordinate a round trip of its own
                const bool inside = idx >= 0 && idx < n - 1;
                const int ia = inside ? idx : (idx == -1 ? 0 : n - 1);
                const T* a = pts + (size_t)ia * D;
                const T* c = inside ? a + D : a;
                T ra[DC], rc[DC];
#pragma unroll
                for (int k = 0; k < DC; ++k) {
                    ra[k] = a[k];
                    rc[k] = c[k];
                }
                Acc d0 = 0, d1 = 0;
                if (inside) {
                    d0 = accum[idx];
                    d1 = accum[idx + 1];
                }
                const Acc len = d1 - d0;
                if (inside && len >= eps) {
                    Acc w0, w1;
                    if constexpr (TY == kPF32) {
                        accv_poly::lerp_weights(d, d0, d1, len, w0, w1);
                    } else {
                        w1 = (d - d0) / len;
                        w0 = (d1 - d) / len;
                    }
#pragma unroll
                    for (int k = 0; k < DC; ++k) {
                        if constexpr (TY == kPF32) {
                            ra[k] = accv_poly::lerp_coord(ra[k], w0, rc[k], w1);
                            res[k] = ra[k];
                        } else {
                            S::store(res + k, S::load(&ra[k]) * w0 + S::load(&rc[k]) * w1);
                        }
                    }
                } else {   // before the first / beyond the last point, or a segment shorter than epsilon: its lower point
#pragma unroll
                    for (int k = 0; k < DC; ++k) S::copy(res + k, &ra[k]);
                }
                if constexpr (TY == kPF32 && DC == 2) {   // what this thread wrote (group boxes: f32, 2-D only)
                    bx = ra[0];
                    by = ra[1];
                }
            } else {
            if (idx >= 0 && idx < n - 1) {
                const Acc d0 = accum[idx], d1 = accum[idx + 1], len = d1 - d0;
                const T* a = pts + (size_t)idx * D;
                const T* c = a + D;
                if (len >= eps) {
                    if constexpr (TY == kPF32) {
                        float w0, w1;
                        accv_poly::lerp_weights(d, d0, d1, len, w0, w1);
                        for (int k = 0; k < D; ++k) res[k] = accv_poly::lerp_coord(a[k], w0, c[k], w1);
                    } else {
                        const Acc w1 = (d - d0) / len, w0 = (d1 - d) / len;
                        for (int k = 0; k < D; ++k) S::store(res + k, S::load(a + k) * w0 + S::load(c + k) * w1);
                    }
                } else {
                    for (int k = 0; k < D; ++k) S::copy(res + k, a + k);
                }
            } else if (idx == -1) {
                for (int k = 0; k < D; ++k) S::copy(res + k, pts + k);
            } else {
                for (int k = 0; k < D; ++k) S::copy(res + k, pts + (size_t)(n - 1) * D + k);
            }
            }
        }
        if (p.out_boxes) {
            const float inf = __builtin_inff();
            const bool ok = bx == bx && by == by;
            float x0 = ok ? bx : inf, y0 = ok ? by : inf, x1 = ok ? bx : -inf, y1 = ok ? by : -inf;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                x0 = fminf(x0, __shfl_xor(x0, sft));
                y0 = fminf(y0, __shfl_xor(y0, sft));
                x1 = fmaxf(x1, __shfl_xor(x1, sft));
                y1 = fmaxf(y1, __shfl_xor(y1, sft));
            }
            const int g = i >> 6;
            if ((t & 63) == 0 && g < groups) reinterpret_cast<float4*>(p.out_boxes)[b * groups + g] = make_float4(x0, y0, x1, y1);
        }
    }
}

inline size_t acc_size(int ty) { return ty == kPF64 ? 8 : 4; }

}  // namespace

extern "C" {

size_t accv_polyline_scratch_bytes(long long batch, int max_points, int dtype)
{
    if (batch <= 0 || max_points <= 0) return 0;
    const size_t need = (size_t)max_points * acc_size(dtype);
    return need <= (size_t)kLdsBudgetBytes ? 0 : need * (size_t)batch;
}

int accv_polyline_sample_boxes(const void* points, const void* distances, const void* point_counts, const void* dist_counts,
                               void* out_points, void* out_lengths, float* out_group_boxes, long long batch, int max_points,
                               int max_distances, int num_dims, int dtype, int counts_i64, int relative, void* scratch,
                               size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (out_group_boxes && (dtype != kPF32 || num_dims != 2 || !out_points))
        return accv::fail(ACCV_EINVAL, "polyline: group boxes need float32 samples of 2-D points");
    if (out_group_boxes && (reinterpret_cast<uintptr_t>(out_group_boxes) & 15u))
        return accv::fail(ACCV_EINVAL, "polyline: group boxes need 16-byte alignment");
    if (batch < 0 || max_points < 0 || max_distances < 0 || num_dims < 0)
        return accv::fail(ACCV_EINVAL, "polyline: negative extent");
    if (dtype < kPF32 || dtype > kPBF16) return accv::fail(ACCV_EINVAL, "polyline: unsupported dtype code %d", dtype);
    if (batch == 0) return ACCV_OK;
    if (!out_points && !out_lengths) return ACCV_OK;
    if (out_points && (max_distances == 0 || num_dims == 0) && !out_lengths) return ACCV_OK;
    if (max_points > 0 && num_dims > 0 && !points) return accv::fail(ACCV_EINVAL, "polyline: null points");
    if (out_points && max_distances > 0 && !distances) return accv::fail(ACCV_EINVAL, "polyline: null distances");
    PolyParams p{};
    p.points = points;
    p.distances = distances;
    p.point_counts = point_counts;
    p.dist_counts = dist_counts;
    p.out_points = (max_distances > 0 && num_dims > 0) ? out_points : nullptr;
    p.out_lengths = out_lengths;
    p.out_boxes = out_group_boxes;
    p.batch = batch;
    p.P = max_points;
    p.Q = max_distances;
    p.D = num_dims;
    p.counts_i64 = counts_i64;
    p.relative = relative;
    const size_t need = accv_polyline_scratch_bytes(batch, max_points, dtype);
    size_t lds = (size_t)std::max(1, max_points) * acc_size(dtype);
    if (need > 0) {
        if (!scratch || scratch_bytes < need)
            return accv::fail(ACCV_EWORKSPACE, "polyline: %d points per polyline need %zu bytes of scratch", max_points, need);
        p.scratch = scratch;
        p.use_scratch = 1;
        lds = 16;
    }
    if (!p.out_points && !p.out_lengths) return ACCV_OK;
    // queries per workgroup: everything, unless few polylines carry thousands of queries each — then the queries of a polyline
    // are cut into chunks (multiples of the workgroup size, so that groups of 64 samples never straddle a chunk), as many as
    // keep the launch at <= ~4096 workgroups (batch 64, 5000 x 5000: 50 us with one workgroup per polyline; profiles/r02_bench_published.jsonl)
    // long polylines take 1024-thread workgroups: the scan over the points is the serial part of a workgroup (5000 points:
    // 20 segments per thread with 256 threads, 5 with 1024)
    const bool wide = max_points >= accv::tune_get("poly_wide", 2048);
    const long long threads = wide ? 1024 : kThreads;
    long long chunks = 1;
    // (every chunk repeats the scan over max_points, so a chunk should hold at least ~max_points / 4 queries)
    long long min_chunk = std::max<long long>(threads, ((max_points / 4 + threads - 1) / threads) * threads);
    // few polylines: the chip is idle anyway, so every thread takes ONE query and the queries of a polyline spread over as
    // many workgroups as that needs (each repeats the scan) — a second query per thread is a second serial pass of loads,
    // binary search and interpolation (batch 1, 2000 x 2000: 9.1 -> 7.7 us, 5000 x 5000: 12.3 -> 10.4 us;
    // profiles/r03_tails_probe_sampler_launch_shapes.log)
    if (batch * ((max_distances + threads - 1) / threads) <= accv::tune_get("poly_spread", 512)) min_chunk = threads;
    if (p.out_points && !p.use_scratch && max_distances >= 2 * min_chunk && batch < 2048)
        chunks = std::max<long long>(1, std::min<long long>((max_distances + min_chunk - 1) / min_chunk, 4096 / batch));
    const long long per_chunk = (max_distances + chunks - 1) / chunks;
    p.q_chunk = (int)std::min<long long>(((per_chunk + threads - 1) / threads) * threads, (long long)INT_MAX - threads);
    if (p.q_chunk < threads) p.q_chunk = (int)threads;
    chunks = std::max<long long>(1, ((long long)max_distances + p.q_chunk - 1) / p.q_chunk);
    const dim3 grid((unsigned)batch, (unsigned)chunks), block((unsigned)threads);
#define ACCV_LAUNCH_POLY_D(TYV, DCV)                                                                        \
    do {                                                                                                    \
        if (wide)                                                                                           \
            hipLaunchKernelGGL((polyline_kernel<TYV, 1024, DCV>), grid, block, lds, stream, p);             \
        else                                                                                                \
            hipLaunchKernelGGL((polyline_kernel<TYV, kThreads, DCV>), grid, block, lds, stream, p);         \
    } while (0)
#define ACCV_LAUNCH_POLY(TYV)                                                                               \
    do {                                                                                                    \
        if (num_dims == 2)                                                                                  \
            ACCV_LAUNCH_POLY_D(TYV, 2);                                                                     \
        else if (num_dims == 3)                                                                             \
            ACCV_LAUNCH_POLY_D(TYV, 3);                                                                     \
        else                                                                                                \
            ACCV_LAUNCH_POLY_D(TYV, 0);                                                                     \
    } while (0)
    switch (dtype) {
        case kPF32: ACCV_LAUNCH_POLY(kPF32); break;
        case kPF64: ACCV_LAUNCH_POLY(kPF64); break;
        case kPF16: ACCV_LAUNCH_POLY(kPF16); break;
        default: ACCV_LAUNCH_POLY(kPBF16); break;
    }
#undef ACCV_LAUNCH_POLY
#undef ACCV_LAUNCH_POLY_D
    return accv::check_launch("polyline");
}

int accv_polyline_sample(const void* points, const void* distances, const void* point_counts, const void* dist_counts,
                         void* out_points, void* out_lengths, long long batch, int max_points, int max_distances,
                         int num_dims, int dtype, int counts_i64, int relative, void* scratch, size_t scratch_bytes,
                         void* stream)
{
    return accv_polyline_sample_boxes(points, distances, point_counts, dist_counts, out_points, out_lengths, nullptr, batch,
                                      max_points, max_distances, num_dims, dtype, counts_i64, relative, scratch,
                                      scratch_bytes, stream);
}
}

// ---------------------------------------------------------------- host path (CPU tensors)
// The reference's CPU implementation (packages/lane_helpers/ext_impl/polyline/src/polyline_cpu.cpp:28-132) accumulates in
// at::acc_type<dtype, false> = double for float and double inputs and runs at::parallel_for over the polylines; this is its
// counterpart for host memory: double accumulation, the same search / clamping / zero-length-segment / empty-polyline
// rules as the kernel above (polyline_common.cuh:58-163), polylines split over std::threads when there is enough work.
namespace {

template <typename T>
void sample_one_host(const T* pts, const T* dist, int n, int q, int dims, bool relative, T* out, T* out_len,
                     std::vector<double>& acc)
{
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (out_len) *out_len = (T)(n == 0 ? nan : 0.0);
    if (n == 0) {
        if (out)
            for (int i = 0; i < q * dims; ++i) out[i] = (T)nan;
        return;
    }
    acc.resize((size_t)n);
    acc[0] = 0.0;
    for (int i = 1; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < dims; ++k) {
            const double d = (double)pts[(size_t)(i - 1) * dims + k] - (double)pts[(size_t)i * dims + k];
            s += d * d;
        }
        acc[(size_t)i] = acc[(size_t)i - 1] + std::sqrt(s);
    }
    const double total = acc[(size_t)n - 1];
    if (out_len) *out_len = (T)total;
    if (!out) return;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int j = 0; j < q; ++j) {
        double d = (double)dist[j];
        if (relative) d *= total;
        // index of the last accumulated distance <= d (-1: before the start, n-1: at / beyond the end)
        int idx;
        if (acc[0] > d) {
            idx = -1;
        } else if (acc[(size_t)n - 1] < d) {
            idx = n - 1;
        } else {
            int lo = 0, hi = n - 1;
            while (hi - lo > 1) {
                const int c = (lo + hi) >> 1;
                if (acc[(size_t)c] < d)
                    lo = c;
                else if (acc[(size_t)c] > d)
                    hi = c;
                else
                    lo = hi = c;
            }
            idx = lo;
        }
        T* o = out + (size_t)j * dims;
        if (idx >= 0 && idx < n - 1) {
            const double seg = acc[(size_t)idx + 1] - acc[(size_t)idx];
            if (seg >= eps) {
                const double w0 = (acc[(size_t)idx + 1] - d) / seg, w1 = (d - acc[(size_t)idx]) / seg;
                for (int k = 0; k < dims; ++k)
                    o[k] = (T)((double)pts[(size_t)idx * dims + k] * w0 + (double)pts[(size_t)(idx + 1) * dims + k] * w1);
            } else {
                for (int k = 0; k < dims; ++k) o[k] = pts[(size_t)idx * dims + k];   // zero-length segment: its lower point
            }
        } else {
            const int e = idx < 0 ? 0 : n - 1;                                        // clamped to an end point
            for (int k = 0; k < dims; ++k) o[k] = pts[(size_t)e * dims + k];
        }
    }
}

template <typename T>
void sample_host(const void* points, const void* distances, const void* pc, const void* dc, void* out_points, void* out_lengths,
                 long long batch, int P, int Q, int D, int counts_i64, int relative, int threads)
{
    auto count_of = [&](const void* c, long long i, int cap) {
        if (!c) return cap;
        const long long v = counts_i64 ? static_cast<const long long*>(c)[i] : (long long)static_cast<const int*>(c)[i];
        return (int)std::max(0ll, std::min(v, (long long)cap));
    };
    auto run = [&](long long lo, long long hi) {
        std::vector<double> acc;
        for (long long i = lo; i < hi; ++i)
            sample_one_host<T>(static_cast<const T*>(points) + (size_t)i * P * D,
                               distances ? static_cast<const T*>(distances) + (size_t)i * Q : nullptr, count_of(pc, i, P),
                               out_points ? count_of(dc, i, Q) : 0, D, relative != 0,
                               out_points ? static_cast<T*>(out_points) + (size_t)i * Q * D : nullptr,
                               out_lengths ? static_cast<T*>(out_lengths) + i : nullptr, acc);
    };
    const long long work = batch * ((long long)P + Q) * std::max(1, D);
    int t = (int)std::min<long long>(std::max(1, threads), std::min<long long>(batch, work / 65536 + 1));
    if (t <= 1) {
        run(0, batch);
        return;
    }
    std::vector<std::thread> pool;
    for (int k = 0; k < t; ++k) pool.emplace_back(run, batch * k / t, batch * (k + 1) / t);
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" int accv_polyline_sample_host(const void* points, const void* distances, const void* point_counts,
                                         const void* dist_counts, void* out_points, void* out_lengths, long long batch,
                                         int max_points, int max_distances, int num_dims, int dtype, int counts_i64,
                                         int relative, int threads)
{
    if (batch < 0 || max_points < 0 || max_distances < 0 || num_dims < 0)
        return accv::fail(ACCV_EINVAL, "polyline (host): negative extent");
    if (dtype != 0 && dtype != 1) return accv::fail(ACCV_EINVAL, "polyline (host): float32 / float64 only, got dtype code %d", dtype);
    if (batch == 0 || (!out_points && !out_lengths)) return ACCV_OK;
    if (max_points > 0 && num_dims > 0 && !points) return accv::fail(ACCV_EINVAL, "polyline (host): null points");
    if (out_points && max_distances > 0 && !distances) return accv::fail(ACCV_EINVAL, "polyline (host): null distances");
    if (threads <= 0) threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (dtype == 0)
        sample_host<float>(points, distances, point_counts, dist_counts, out_points, out_lengths, batch, max_points,
                           max_distances, num_dims, counts_i64, relative, threads);
    else
        sample_host<double>(points, distances, point_counts, dist_counts, out_points, out_lengths, batch, max_points,
                            max_distances, num_dims, counts_i64, relative, threads);
    return ACCV_OK;
}
