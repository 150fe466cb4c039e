// F3 — the loss-side caller pattern of batching_helpers as ONE kernel per direction:
//
//   out[i] = sum_{j < counts[i]}  w(i,j) * sum_k  l( a[i, idx_a[i,j], k],  b[i, idx_b[i,j], k] )
//
// i.e. the matched gathers (batched_indexing_access on both sides of a match), an element-wise per-object loss and the
// masked per-sample reduction (sum_over_targets) that packages/batching_helpers/example/loss_computation.py:37-43,85-86
// spells as five gathers + element-wise torch code + two masked sums.  The gathered rows are never materialised.
// One workgroup per sample: 256 threads stride over (pair, element), wave shuffle + LDS tree reduction in a FIXED order,
// so the result is deterministic (no atomics in the forward).  The backward scatters
//   d a[i, idx_a[i,j], k] += g[i] * w * dl/da,   d b[i, idx_b[i,j], k] += g[i] * w * dl/db
// with float atomics into zero-initialised gradients (matches are one-to-one in practice, duplicates stay correct).
// Memory / launch bound (a few hundred KB per call): no MFMA, no roofline claim beyond "one launch instead of ~12".
#include <hip/hip_runtime.h>

#include <cstdint>

#include "accv_common.h"

namespace {

enum Kind { kL1 = 0, kL2 = 1, kSmoothL1 = 2 };

__device__ __forceinline__ long long load_int(const void* p, long long i, int is64)
{
    return is64 ? static_cast<const long long*>(p)[i] : (long long)static_cast<const int*>(p)[i];
}
__device__ __forceinline__ long long wrap_index(long long j, long long width)
{
    if (j < 0) j += width;   // negative indices wrap once, as in the gather kernels (cu:75-77)
    return (j < 0 || j >= width) ? -1 : j;
}

template <int KIND>
__device__ __forceinline__ float loss_of(float d, float beta)
{
    const float ad = fabsf(d);
    if (KIND == kL1) return ad;
    if (KIND == kL2) return d * d;
    return ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta;   // torch.nn.functional.smooth_l1_loss
}
template <int KIND>
__device__ __forceinline__ float dloss_of(float d, float beta)
{
    if (KIND == kL1) return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    if (KIND == kL2) return 2.0f * d;
    const float ad = fabsf(d);
    return ad < beta ? d / beta : (d > 0.0f ? 1.0f : -1.0f);
}

struct MatchedDesc {
    const float* a;        // [B, w_a, row]
    const float* b;        // [B, w_b, row]
    const void* idx_a;     // [B, idx_stride]
    const void* idx_b;     // [B, idx_stride]
    const void* counts;    // [B]
    const float* weights;  // [B, w_a] (indexed like a) or null
    long long w_a, w_b, w_idx, idx_stride, row;
    int idx_i64, counts_i64;
    float beta;
};

template <int KIND>
__global__ __launch_bounds__(256) void matched_reduce_kernel(const MatchedDesc d, float* __restrict__ out)
{
    __shared__ float s_part[4];
    const long long i = blockIdx.x;
    const long long n = max(0ll, min(load_int(d.counts, i, d.counts_i64), d.w_idx));
    const long long total = n * d.row;
    float acc = 0.0f;
    for (long long t = threadIdx.x; t < total; t += 256) {
        const long long j = t / d.row, k = t - j * d.row;
        const long long ga = wrap_index(load_int(d.idx_a, i * d.idx_stride + j, d.idx_i64), d.w_a);
        const long long gb = wrap_index(load_int(d.idx_b, i * d.idx_stride + j, d.idx_i64), d.w_b);
        if (ga < 0 || gb < 0) continue;
        const float w = d.weights ? d.weights[i * d.w_a + ga] : 1.0f;
        const float diff = d.a[(i * d.w_a + ga) * d.row + k] - d.b[(i * d.w_b + gb) * d.row + k];
        acc += w * loss_of<KIND>(diff, d.beta);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

template <int KIND>
__global__ __launch_bounds__(256) void matched_reduce_bwd_kernel(const MatchedDesc d, const float* __restrict__ grad_out,
                                                                 float* __restrict__ grad_a, float* __restrict__ grad_b,
                                                                 float* __restrict__ grad_w)
{
    const long long i = blockIdx.x;
    const long long n = max(0ll, min(load_int(d.counts, i, d.counts_i64), d.w_idx));
    const long long total = n * d.row;
    const float g = grad_out[i];
    for (long long t = threadIdx.x; t < total; t += 256) {
        const long long j = t / d.row, k = t - j * d.row;
        const long long ga = wrap_index(load_int(d.idx_a, i * d.idx_stride + j, d.idx_i64), d.w_a);
        const long long gb = wrap_index(load_int(d.idx_b, i * d.idx_stride + j, d.idx_i64), d.w_b);
        if (ga < 0 || gb < 0) continue;
        const float w = d.weights ? d.weights[i * d.w_a + ga] : 1.0f;
        const long long ea = (i * d.w_a + ga) * d.row + k, eb = (i * d.w_b + gb) * d.row + k;
        const float diff = d.a[ea] - d.b[eb];
        const float dl = g * w * dloss_of<KIND>(diff, d.beta);
        if (grad_a) atomicAdd(grad_a + ea, dl);
        if (grad_b) atomicAdd(grad_b + eb, -dl);
        if (grad_w) atomicAdd(grad_w + i * d.w_a + ga, g * loss_of<KIND>(diff, d.beta));
    }
}

int check(const char* who, const MatchedDesc& d, long long batch, int kind)
{
    if (batch < 0 || d.w_a < 0 || d.w_b < 0 || d.w_idx < 0 || d.row < 0 || d.idx_stride < d.w_idx)
        return accv::fail(ACCV_EINVAL, "%s: invalid extents", who);
    if (kind < 0 || kind > kSmoothL1) return accv::fail(ACCV_EINVAL, "%s: unknown loss kind %d", who, kind);
    if (kind == kSmoothL1 && !(d.beta > 0.0f)) return accv::fail(ACCV_EINVAL, "%s: smooth-L1 needs beta > 0", who);
    if (batch > 0x7fffffffll) return accv::fail(ACCV_EINVAL, "%s: batch exceeds the grid limit", who);
    return ACCV_OK;
}

}  // namespace

extern "C" {

int accv_matched_pair_reduce_f32(const float* a, const float* b, const void* idx_a, const void* idx_b, const void* counts,
                                 const float* weights_or_null, long long batch, long long w_a, long long w_b,
                                 long long w_idx, long long idx_stride, long long row_elems, int kind, float beta,
                                 int idx_i64, int counts_i64, float* out, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MatchedDesc d{a, b, idx_a, idx_b, counts, weights_or_null, w_a, w_b, w_idx, idx_stride, row_elems, idx_i64, counts_i64, beta};
    if (int rc = check("matched_pair_reduce", d, batch, kind)) return rc;
    if (batch == 0) return ACCV_OK;
    if (!out || !counts) return accv::fail(ACCV_EINVAL, "matched_pair_reduce: null output / counts pointer");
    if (w_idx * row_elems > 0 && (!a || !b || !idx_a || !idx_b))
        return accv::fail(ACCV_EINVAL, "matched_pair_reduce: null data / index pointer");
    const dim3 grid((unsigned)batch), block(256);
    switch (kind) {
        case kL1: hipLaunchKernelGGL((matched_reduce_kernel<kL1>), grid, block, 0, stream, d, out); break;
        case kL2: hipLaunchKernelGGL((matched_reduce_kernel<kL2>), grid, block, 0, stream, d, out); break;
        default: hipLaunchKernelGGL((matched_reduce_kernel<kSmoothL1>), grid, block, 0, stream, d, out); break;
    }
    return accv::check_launch("matched_pair_reduce");
}

int accv_matched_pair_reduce_bwd_f32(const float* a, const float* b, const void* idx_a, const void* idx_b,
                                     const void* counts, const float* weights_or_null, const float* grad_out,
                                     long long batch, long long w_a, long long w_b, long long w_idx,
                                     long long idx_stride, long long row_elems, int kind, float beta, int idx_i64,
                                     int counts_i64, float* grad_a_or_null, float* grad_b_or_null,
                                     float* grad_w_or_null, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MatchedDesc d{a, b, idx_a, idx_b, counts, weights_or_null, w_a, w_b, w_idx, idx_stride, row_elems, idx_i64, counts_i64, beta};
    if (int rc = check("matched_pair_reduce_bwd", d, batch, kind)) return rc;
    if (batch == 0 || w_idx * row_elems == 0) return ACCV_OK;
    if (!a || !b || !idx_a || !idx_b || !counts || !grad_out)
        return accv::fail(ACCV_EINVAL, "matched_pair_reduce_bwd: null pointer");
    if (grad_w_or_null && !weights_or_null)
        return accv::fail(ACCV_EINVAL, "matched_pair_reduce_bwd: a weight gradient needs weights");
    const dim3 grid((unsigned)batch), block(256);
#define BWD(K) hipLaunchKernelGGL((matched_reduce_bwd_kernel<K>), grid, block, 0, stream, d, grad_out, grad_a_or_null, grad_b_or_null, grad_w_or_null)
    switch (kind) {
        case kL1: BWD(kL1); break;
        case kL2: BWD(kL2); break;
        default: BWD(kSmoothL1); break;
    }
#undef BWD
    return accv::check_launch("matched_pair_reduce_bwd");
}
}
