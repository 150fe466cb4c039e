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
#include <hip/hip_fp16.h>

#include <cstdint>

#include "accv_common.h"

namespace {

// kinds 0-2: element-wise on d = a - b, summed over the row.  Round 3 adds the two per-object losses of the reference
// example itself (packages/batching_helpers/example/loss_computation.py):
//   kIoUxyxy   (:240-274, _per_object_bbox_overlap_loss) rows are boxes (x0, y0, x1, y1): 1 - intersection / max(union, eps)
//   kOneHotL1  (:37-43 class branch + :225-238) a holds integer class labels [B, w_a], b the scores [B, w_b, C]:
//              sum_c |onehot(label)[c] - score[c]|
enum Kind { kL1 = 0, kL2 = 1, kSmoothL1 = 2, kIoUxyxy = 3, kOneHotL1 = 4 };
enum DType { kF32 = 0, kF16 = 1, kBF16 = 2, kF64 = 3 };
struct bf16_raw {
    uint16_t v;
};

// arithmetic / accumulation type: float for f32, f16 and bf16 data (the gathers this kernel replaces accept them,
// batched_indexing_access_cuda_impl.cu:269-286), double for f64
template <class T>
struct AccOf {
    using type = float;
};
template <>
struct AccOf<double> {
    using type = double;
};
__device__ __forceinline__ float to_acc(float v) { return v; }
__device__ __forceinline__ double to_acc(double v) { return v; }
__device__ __forceinline__ float to_acc(__half v) { return __half2float(v); }
__device__ __forceinline__ float to_acc(bf16_raw v) { return __uint_as_float((unsigned)v.v << 16); }

__device__ __forceinline__ long long load_int(const void* p, long long i, int is64)
{
    return is64 ? static_cast<const long long*>(p)[i] : (long long)static_cast<const int*>(p)[i];
}
__device__ __forceinline__ void load_int_pair(const void* p, const void* q, long long i, int is64, long long& a, long long& b)
{
    if (is64) {
        a = static_cast<const long long*>(p)[i];
        b = static_cast<const long long*>(q)[i];
    } else {
        a = static_cast<const int*>(p)[i];
        b = static_cast<const int*>(q)[i];
    }
}
__device__ __forceinline__ long long wrap_index(long long j, long long width)
{
    if (j < 0) j += width;   // negative indices wrap once, as in the gather kernels (cu:75-77)
    return (j < 0 || j >= width) ? -1 : j;
}

template <int KIND, class A>
__device__ __forceinline__ A loss_of(A d, A beta)
{
    const A ad = d < A(0) ? -d : d;
    if (KIND == kL1 || KIND == kOneHotL1) return ad;
    if (KIND == kL2) return d * d;
    return ad < beta ? A(0.5) * d * d / beta : ad - A(0.5) * beta;   // torch.nn.functional.smooth_l1_loss
}
template <int KIND, class A>
__device__ __forceinline__ A dloss_of(A d, A beta)
{
    if (KIND == kL1 || KIND == kOneHotL1) return d > A(0) ? A(1) : (d < A(0) ? A(-1) : A(0));
    if (KIND == kL2) return A(2) * d;
    const A ad = d < A(0) ? -d : d;
    return ad < beta ? d / beta : (d > A(0) ? A(1) : A(-1));
}

// 1 - IoU of two (x0, y0, x1, y1) boxes with the reference's clamps (negative intersection extents -> 0, union < eps -> eps);
// optionally the gradient w.r.t. the eight coordinates as torch's autograd gives it for that code: the masked assignments
// pass no gradient where they fired, torch.max / torch.min split the gradient evenly on ties
template <class A, bool GRAD>
__device__ __forceinline__ A iou_loss(const A (&a)[4], const A (&b)[4], A eps, A (&da)[4], A (&db)[4])
{
    const A wa = a[2] - a[0], ha = a[3] - a[1], wb = b[2] - b[0], hb = b[3] - b[1];
    const A area_a = wa * ha, area_b = wb * hb;
    const A ulx = a[0] > b[0] ? a[0] : b[0], uly = a[1] > b[1] ? a[1] : b[1];
    const A lrx = a[2] < b[2] ? a[2] : b[2], lry = a[3] < b[3] ? a[3] : b[3];
    const A iw_raw = lrx - ulx, ih_raw = lry - uly;
    const A iw = iw_raw < A(0) ? A(0) : iw_raw, ih = ih_raw < A(0) ? A(0) : ih_raw;
    const A inter = iw * ih;
    const A uni = area_a + area_b - inter;
    const A uni_c = uni < eps ? eps : uni;
    if constexpr (GRAD) {
        const A g_uni = uni < eps ? A(0) : inter / (uni_c * uni_c);   // d loss / d union (through the clamp)
        const A g_inter = -A(1) / uni_c - g_uni;                       // inter appears in the ratio and in the union
        const A g_iw = iw_raw < A(0) ? A(0) : g_inter * ih, g_ih = ih_raw < A(0) ? A(0) : g_inter * iw;
        // share of a in max(a, b) / min(a, b): 1, 0 or 1/2 on a tie
        auto share_max = [](A x, A y) { return x > y ? A(1) : (x < y ? A(0) : A(0.5)); };
        auto share_min = [](A x, A y) { return x < y ? A(1) : (x > y ? A(0) : A(0.5)); };
        const A s0 = share_max(a[0], b[0]), s1 = share_max(a[1], b[1]), s2 = share_min(a[2], b[2]), s3 = share_min(a[3], b[3]);
        da[0] = -g_iw * s0 - g_uni * ha;
        da[1] = -g_ih * s1 - g_uni * wa;
        da[2] = g_iw * s2 + g_uni * ha;
        da[3] = g_ih * s3 + g_uni * wa;
        db[0] = -g_iw * (A(1) - s0) - g_uni * hb;
        db[1] = -g_ih * (A(1) - s1) - g_uni * wb;
        db[2] = g_iw * (A(1) - s2) + g_uni * hb;
        db[3] = g_ih * (A(1) - s3) + g_uni * wb;
    }
    return A(1) - inter / uni_c;
}

struct MatchedDesc {
    const void* a;         // [B, w_a, row] data dtype; kOneHotL1: integer labels [B, w_a]
    const void* b;         // [B, w_b, row]
    const void* idx_a;     // [B, idx_stride]
    const void* idx_b;     // [B, idx_stride]
    const void* counts;    // [B]
    const void* weights;   // [B, w_a] data dtype (indexed like a) or null
    long long w_a, w_b, w_idx, idx_stride, row;
    int idx_i64, counts_i64, labels_i64;
    float beta, eps;
};

// wave shuffle + LDS tree in a FIXED order: deterministic
template <class A>
__device__ __forceinline__ void block_sum_store(A acc, A* __restrict__ out, long long i)
{
    __shared__ A s_part[4];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

template <int KIND, class T>
__global__ __launch_bounds__(256) void matched_reduce_kernel(const MatchedDesc d, typename AccOf<T>::type* __restrict__ out)
{
    using A = typename AccOf<T>::type;
    const T* __restrict__ da = static_cast<const T*>(d.a);
    const T* __restrict__ db = static_cast<const T*>(d.b);
    const T* __restrict__ dw = static_cast<const T*>(d.weights);
    const long long i = blockIdx.x;
    const long long n = max(0ll, min(load_int(d.counts, i, d.counts_i64), d.w_idx));
    const long long work = KIND == kIoUxyxy ? 1 : d.row;   // work items per pair
    const long long total = n * work;
    A acc = A(0);
    for (long long t = threadIdx.x; t < total; t += 256) {
        const long long j = t / work, k = t - j * work;
        long long ia, ib;   // both match indices in ONE branch on their dtype: two load_int() calls are two branch diamonds, and
                            // hipcc waits for the first index before it requests the second (a dependent round trip for nothing)
        load_int_pair(d.idx_a, d.idx_b, i * d.idx_stride + j, d.idx_i64, ia, ib);
        const long long ga = wrap_index(ia, d.w_a), gb = wrap_index(ib, d.w_b);
        if (ga < 0 || gb < 0) continue;
        const A w = dw ? to_acc(dw[i * d.w_a + ga]) : A(1);
        if constexpr (KIND == kIoUxyxy) {
            A ba[4], bb[4], ga_[4], gb_[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ba[c] = to_acc(da[(i * d.w_a + ga) * 4 + c]);
                bb[c] = to_acc(db[(i * d.w_b + gb) * 4 + c]);
            }
            acc += w * iou_loss<A, false>(ba, bb, (A)d.eps, ga_, gb_);
        } else if constexpr (KIND == kOneHotL1) {
            const long long label = load_int(d.a, i * d.w_a + ga, d.labels_i64);
            const A target = (k == label) ? A(1) : A(0);
            acc += w * loss_of<KIND, A>(target - to_acc(db[(i * d.w_b + gb) * d.row + k]), A(0));
        } else {
            const A diff = to_acc(da[(i * d.w_a + ga) * d.row + k]) - to_acc(db[(i * d.w_b + gb) * d.row + k]);
            acc += w * loss_of<KIND, A>(diff, (A)d.beta);
        }
    }
    block_sum_store<A>(acc, out, i);
}

template <int KIND, class T>
__global__ __launch_bounds__(256) void matched_reduce_bwd_kernel(const MatchedDesc d,
                                                                 const typename AccOf<T>::type* __restrict__ grad_out,
                                                                 typename AccOf<T>::type* __restrict__ grad_a,
                                                                 typename AccOf<T>::type* __restrict__ grad_b,
                                                                 typename AccOf<T>::type* __restrict__ grad_w)
{
    using A = typename AccOf<T>::type;
    const T* __restrict__ da = static_cast<const T*>(d.a);
    const T* __restrict__ db = static_cast<const T*>(d.b);
    const T* __restrict__ dw = static_cast<const T*>(d.weights);
    const long long i = blockIdx.x;
    const long long n = max(0ll, min(load_int(d.counts, i, d.counts_i64), d.w_idx));
    const long long work = KIND == kIoUxyxy ? 1 : d.row;
    const long long total = n * work;
    const A g = grad_out[i];
    for (long long t = threadIdx.x; t < total; t += 256) {
        const long long j = t / work, k = t - j * work;
        long long ia, ib;   // both match indices in ONE branch on their dtype: two load_int() calls are two branch diamonds, and
                            // hipcc waits for the first index before it requests the second (a dependent round trip for nothing)
        load_int_pair(d.idx_a, d.idx_b, i * d.idx_stride + j, d.idx_i64, ia, ib);
        const long long ga = wrap_index(ia, d.w_a), gb = wrap_index(ib, d.w_b);
        if (ga < 0 || gb < 0) continue;
        const A w = dw ? to_acc(dw[i * d.w_a + ga]) : A(1);
        if constexpr (KIND == kIoUxyxy) {
            A ba[4], bb[4], ga_[4], gb_[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ba[c] = to_acc(da[(i * d.w_a + ga) * 4 + c]);
                bb[c] = to_acc(db[(i * d.w_b + gb) * 4 + c]);
            }
            const A l = iou_loss<A, true>(ba, bb, (A)d.eps, ga_, gb_);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (grad_a) atomicAdd(grad_a + (i * d.w_a + ga) * 4 + c, g * w * ga_[c]);
                if (grad_b) atomicAdd(grad_b + (i * d.w_b + gb) * 4 + c, g * w * gb_[c]);
            }
            if (grad_w) atomicAdd(grad_w + i * d.w_a + ga, g * l);
        } else if constexpr (KIND == kOneHotL1) {
            const long long label = load_int(d.a, i * d.w_a + ga, d.labels_i64);
            const long long eb = (i * d.w_b + gb) * d.row + k;
            const A diff = ((k == label) ? A(1) : A(0)) - to_acc(db[eb]);
            if (grad_b) atomicAdd(grad_b + eb, -g * w * dloss_of<KIND, A>(diff, A(0)));
            if (grad_w) atomicAdd(grad_w + i * d.w_a + ga, g * loss_of<KIND, A>(diff, A(0)));
        } else {
            const long long ea = (i * d.w_a + ga) * d.row + k, eb = (i * d.w_b + gb) * d.row + k;
            const A diff = to_acc(da[ea]) - to_acc(db[eb]);
            const A dl = g * w * dloss_of<KIND, A>(diff, (A)d.beta);
            if (grad_a) atomicAdd(grad_a + ea, dl);
            if (grad_b) atomicAdd(grad_b + eb, -dl);
            if (grad_w) atomicAdd(grad_w + i * d.w_a + ga, g * loss_of<KIND, A>(diff, (A)d.beta));
        }
    }
}

int check(const char* who, const MatchedDesc& d, long long batch, int kind, int dtype)
{
    if (batch < 0 || d.w_a < 0 || d.w_b < 0 || d.w_idx < 0 || d.row < 0 || d.idx_stride < d.w_idx)
        return accv::fail(ACCV_EINVAL, "%s: invalid extents", who);
    if (kind < 0 || kind > kOneHotL1) return accv::fail(ACCV_EINVAL, "%s: unknown loss kind %d", who, kind);
    if (dtype < 0 || dtype > kF64) return accv::fail(ACCV_EINVAL, "%s: unknown data type code %d", who, dtype);
    if (kind == kSmoothL1 && !(d.beta > 0.0f)) return accv::fail(ACCV_EINVAL, "%s: smooth-L1 needs beta > 0", who);
    if (kind == kIoUxyxy && d.row != 4) return accv::fail(ACCV_EINVAL, "%s: the IoU loss needs rows of 4 (x0, y0, x1, y1)", who);
    if (batch > 0x7fffffffll) return accv::fail(ACCV_EINVAL, "%s: batch exceeds the grid limit", who);
    return ACCV_OK;
}

template <class T>
void launch_fwd(int kind, const dim3& grid, hipStream_t stream, const MatchedDesc& d, void* out_)
{
    auto* out = static_cast<typename AccOf<T>::type*>(out_);
    const dim3 block(256);
    switch (kind) {
        case kL1: hipLaunchKernelGGL((matched_reduce_kernel<kL1, T>), grid, block, 0, stream, d, out); break;
        case kL2: hipLaunchKernelGGL((matched_reduce_kernel<kL2, T>), grid, block, 0, stream, d, out); break;
        case kSmoothL1: hipLaunchKernelGGL((matched_reduce_kernel<kSmoothL1, T>), grid, block, 0, stream, d, out); break;
        case kIoUxyxy: hipLaunchKernelGGL((matched_reduce_kernel<kIoUxyxy, T>), grid, block, 0, stream, d, out); break;
        default: hipLaunchKernelGGL((matched_reduce_kernel<kOneHotL1, T>), grid, block, 0, stream, d, out); break;
    }
}
template <class T>
void launch_bwd(int kind, const dim3& grid, hipStream_t stream, const MatchedDesc& d, const void* go_, void* ga_, void* gb_,
                void* gw_)
{
    using A = typename AccOf<T>::type;
    const A* go = static_cast<const A*>(go_);
    A *ga = static_cast<A*>(ga_), *gb = static_cast<A*>(gb_), *gw = static_cast<A*>(gw_);
    const dim3 block(256);
#define BWD(K) hipLaunchKernelGGL((matched_reduce_bwd_kernel<K, T>), grid, block, 0, stream, d, go, ga, gb, gw)
    switch (kind) {
        case kL1: BWD(kL1); break;
        case kL2: BWD(kL2); break;
        case kSmoothL1: BWD(kSmoothL1); break;
        case kIoUxyxy: BWD(kIoUxyxy); break;
        default: BWD(kOneHotL1); break;
    }
#undef BWD
}

}  // namespace

extern "C" {

int accv_matched_pair_reduce(const void* a, const void* b, const void* idx_a, const void* idx_b, const void* counts,
                             const void* weights_or_null, long long batch, long long w_a, long long w_b, long long w_idx,
                             long long idx_stride, long long row_elems, int kind, int dtype, float beta, float eps,
                             unsigned flags, void* out, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MatchedDesc d{a, b, idx_a, idx_b, counts, weights_or_null, w_a, w_b, w_idx, idx_stride, row_elems,
                  (flags & ACCV_MP_IDX_I64) ? 1 : 0, (flags & ACCV_MP_COUNTS_I64) ? 1 : 0, (flags & ACCV_MP_LABELS_I64) ? 1 : 0,
                  beta, eps};
    if (int rc = check("matched_pair_reduce", d, batch, kind, dtype)) return rc;
    if (batch == 0) return ACCV_OK;
    if (!out || !counts) return accv::fail(ACCV_EINVAL, "matched_pair_reduce: null output / counts pointer");
    if (w_idx * row_elems > 0 && (!a || !b || !idx_a || !idx_b))
        return accv::fail(ACCV_EINVAL, "matched_pair_reduce: null data / index pointer");
    const dim3 grid((unsigned)batch);
    switch (dtype) {
        case kF32: launch_fwd<float>(kind, grid, stream, d, out); break;
        case kF16: launch_fwd<__half>(kind, grid, stream, d, out); break;
        case kBF16: launch_fwd<bf16_raw>(kind, grid, stream, d, out); break;
        default: launch_fwd<double>(kind, grid, stream, d, out); break;
    }
    return accv::check_launch("matched_pair_reduce");
}

int accv_matched_pair_reduce_bwd(const void* a, const void* b, const void* idx_a, const void* idx_b, const void* counts,
                                 const void* weights_or_null, const void* grad_out, long long batch, long long w_a,
                                 long long w_b, long long w_idx, long long idx_stride, long long row_elems, int kind,
                                 int dtype, float beta, float eps, unsigned flags, void* grad_a_or_null,
                                 void* grad_b_or_null, void* grad_w_or_null, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MatchedDesc d{a, b, idx_a, idx_b, counts, weights_or_null, w_a, w_b, w_idx, idx_stride, row_elems,
                  (flags & ACCV_MP_IDX_I64) ? 1 : 0, (flags & ACCV_MP_COUNTS_I64) ? 1 : 0, (flags & ACCV_MP_LABELS_I64) ? 1 : 0,
                  beta, eps};
    if (int rc = check("matched_pair_reduce_bwd", d, batch, kind, dtype)) return rc;
    if (batch == 0 || w_idx * row_elems == 0) return ACCV_OK;
    if (!a || !b || !idx_a || !idx_b || !counts || !grad_out)
        return accv::fail(ACCV_EINVAL, "matched_pair_reduce_bwd: null pointer");
    if (grad_w_or_null && !weights_or_null)
        return accv::fail(ACCV_EINVAL, "matched_pair_reduce_bwd: a weight gradient needs weights");
    if (kind == kOneHotL1 && grad_a_or_null)
        return accv::fail(ACCV_EINVAL, "matched_pair_reduce_bwd: integer labels have no gradient");
    const dim3 grid((unsigned)batch);
    switch (dtype) {
        case kF32: launch_bwd<float>(kind, grid, stream, d, grad_out, grad_a_or_null, grad_b_or_null, grad_w_or_null); break;
        case kF16: launch_bwd<__half>(kind, grid, stream, d, grad_out, grad_a_or_null, grad_b_or_null, grad_w_or_null); break;
        case kBF16: launch_bwd<bf16_raw>(kind, grid, stream, d, grad_out, grad_a_or_null, grad_b_or_null, grad_w_or_null); break;
        default: launch_bwd<double>(kind, grid, stream, d, grad_out, grad_a_or_null, grad_b_or_null, grad_w_or_null); break;
    }
    return accv::check_launch("matched_pair_reduce_bwd");
}

/* the float32 entry points of round 2 (kinds 0-2), kept for ABI stability */
int accv_matched_pair_reduce_f32(const float* a, const float* b, const void* idx_a, const void* idx_b, const void* counts,
                                 const float* weights_or_null, long long batch, long long w_a, long long w_b,
                                 long long w_idx, long long idx_stride, long long row_elems, int kind, float beta,
                                 int idx_i64, int counts_i64, float* out, void* stream_)
{
    if (kind < 0 || kind > kSmoothL1) return accv::fail(ACCV_EINVAL, "matched_pair_reduce: unknown loss kind %d", kind);
    return accv_matched_pair_reduce(a, b, idx_a, idx_b, counts, weights_or_null, batch, w_a, w_b, w_idx, idx_stride, row_elems,
                                    kind, kF32, beta, 0.0f, (idx_i64 ? ACCV_MP_IDX_I64 : 0u) | (counts_i64 ? ACCV_MP_COUNTS_I64 : 0u),
                                    out, stream_);
}

int accv_matched_pair_reduce_bwd_f32(const float* a, const float* b, const void* idx_a, const void* idx_b,
                                     const void* counts, const float* weights_or_null, const float* grad_out,
                                     long long batch, long long w_a, long long w_b, long long w_idx,
                                     long long idx_stride, long long row_elems, int kind, float beta, int idx_i64,
                                     int counts_i64, float* grad_a_or_null, float* grad_b_or_null,
                                     float* grad_w_or_null, void* stream_)
{
    if (kind < 0 || kind > kSmoothL1) return accv::fail(ACCV_EINVAL, "matched_pair_reduce_bwd: unknown loss kind %d", kind);
    return accv_matched_pair_reduce_bwd(a, b, idx_a, idx_b, counts, weights_or_null, grad_out, batch, w_a, w_b, w_idx, idx_stride,
                                        row_elems, kind, kF32, beta, 0.0f,
                                        (idx_i64 ? ACCV_MP_IDX_I64 : 0u) | (counts_i64 ? ACCV_MP_COUNTS_I64 : 0u), grad_a_or_null,
                                        grad_b_or_null, grad_w_or_null, stream_);
}
}
