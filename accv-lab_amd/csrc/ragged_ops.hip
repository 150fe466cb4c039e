// H2 — ragged batch kernels for gfx950 (MI355X): gather / scatter / index-pair mapping / constant insert /
// pad fill / mask -> compact indices (wave ballot + popcount prefix) / flat -> padded pack.
//
// Replaces the device layer of the reference's batching_helpers extension:
//   packages/batching_helpers/accvlab/batching_helpers/cpp_impl/batched_indexing_access_cuda_impl.cu
//   :52-113 (indexing_kernel), :115-160 (map_values_by_index_pairs_kernel), :162-194
//   (insert_const_at_indices_kernel), :196-213 (set_ragged_batch_padded_to_filler_value_kernel)
// Design differences (not a translation):
//   * all COPY ops are dtype-agnostic byte movers over `row_bytes` per index (bit-exact by construction),
//     vectorised to the widest of 16/8/4/2/1 bytes that divides the row and matches the base alignment,
//     consecutive lanes on consecutive vectors of one row (coalesced);
//   * ACCUMULATE ops use hardware atomics (f32/f64/i32/i64) or a 32-bit CAS on the containing word
//     (f16/bf16) instead of the reference's per-element spin lock (:30-50), which cannot make progress when
//     two lanes of one lock-step wave contend; "set first, then add" is realised as clear-touched-slots
//     followed by atomic adds;
//   * mask compaction (the reference uses torch boolean indexing, batched_bool_indexing.py:195-221) is a
//     single pass per row with __ballot/__popcll running offsets: order preserving, no atomics;
//   * out-of-range indices are skipped and counted in an optional error counter instead of a device assert.
#include <hip/hip_runtime.h>


#include <cstdint>

#include "accv_common.h"

namespace {

template <int VB>
struct VecOf;
template <>
struct VecOf<16> {
    using type = uint4;
};
template <>
struct VecOf<8> {
    using type = uint2;
};
template <>
struct VecOf<4> {
    using type = uint32_t;
};
template <>
struct VecOf<2> {
    using type = uint16_t;
};
template <>
struct VecOf<1> {
    using type = uint8_t;
};

struct RaggedDesc {
    const void* idx_a;   // primary index tensor [B, idx_stride]
    const void* idx_b;   // second index tensor (pair mapping) or null
    const void* counts;  // [B]
    long long batch, w_idx, idx_stride;
    long long w_src, w_dst;  // extent of the indexed dimension on the gathered / scattered side
    long long row_vecs;      // vectors per indexed row
    int idx_i64, counts_i64;
    int* err;  // optional out-of-range counter
};

__device__ __forceinline__ long long load_int(const void* p, long long i, int is64)
{
    return is64 ? static_cast<const long long*>(p)[i] : (long long)static_cast<const int*>(p)[i];
}

// negative indices wrap once (reference :75-77); returns -1 when out of range
__device__ __forceinline__ long long wrap_index(long long j, long long width, int* err)
{
    if (j < 0) j += width;
    if (j < 0 || j >= width) {
        if (err) atomicAdd(err, 1);
        return -1;
    }
    return j;
}

enum Mode { kGather = 0, kScatter = 1, kMapPairs = 2 };

// Launch geometry shared by the row kernels: NO per-thread division.  A row = one (sample i, slot j) pair of `row_vecs`
// vectors.  2^lg lanes work on a row (consecutive lanes = consecutive vectors: coalesced), a 256-thread workgroup
// covers 256 >> lg consecutive slots of ONE sample: blockIdx.x = slot group, blockIdx.y = sample (looped when the
// batch exceeds the grid's y extent).  For the row sizes of this path (<= 4 KB) every thread moves exactly one
// vector, i.e. every wave issues ONE load and ONE store — the pattern that streams fastest on this chip.
struct RowGeom {
    int lg;  // log2(lanes per row), 0..8
};
__device__ __forceinline__ bool row_of_thread(const RowGeom g, long long w_idx, long long& j, int& v0, int& lanes)
{
    lanes = 1 << g.lg;
    v0 = (int)threadIdx.x & (lanes - 1);
    j = (long long)blockIdx.x * (256 >> g.lg) + ((int)threadIdx.x >> g.lg);
    return j < w_idx;
}

template <int VB, int MODE>
__global__ __launch_bounds__(256) void copy_rows_kernel(const RaggedDesc d, const RowGeom g, const void* __restrict__ src_,
                                                        void* __restrict__ dst_)
{
    using V = typename VecOf<VB>::type;
    const V* src = static_cast<const V*>(src_);
    V* dst = static_cast<V*>(dst_);
    long long j;
    int v0, lanes;
    if (!row_of_thread(g, d.w_idx, j, v0, lanes)) return;
    for (long long i = blockIdx.y; i < d.batch; i += gridDim.y) {
        // the count (uniform: scalar load) and the index of this slot are fetched together; slots behind the count hold
        // anything (the reference fills them with out-of-range junk) and are never interpreted
        const long long slot = i * d.idx_stride + j;
        const long long a = load_int(d.idx_a, slot, d.idx_i64);
        const long long b = MODE == kMapPairs ? load_int(d.idx_b, slot, d.idx_i64) : 0;
        if (j >= load_int(d.counts, i, d.counts_i64)) continue;
        long long from, to;
        if (MODE == kGather) {
            const long long s = wrap_index(a, d.w_src, d.err);
            if (s < 0) continue;
            from = (i * d.w_src + s) * d.row_vecs;
            to = (i * d.w_idx + j) * d.row_vecs;
        } else if (MODE == kScatter) {
            const long long o = wrap_index(a, d.w_dst, d.err);
            if (o < 0) continue;
            from = (i * d.w_idx + j) * d.row_vecs;
            to = (i * d.w_dst + o) * d.row_vecs;
        } else {
            const long long s = wrap_index(a, d.w_src, d.err);
            const long long o = wrap_index(b, d.w_dst, d.err);
            if (s < 0 || o < 0) continue;
            from = (i * d.w_src + s) * d.row_vecs;
            to = (i * d.w_dst + o) * d.row_vecs;
        }
        for (long long v = v0; v < d.row_vecs; v += lanes) dst[to + v] = src[from + v];
    }
}

// gather that also writes the filler: dst[i, j, :] = src[i, idx[i,j], :] for valid (i, j), `pattern` elsewhere, so the
// caller hands over an UNINITIALISED result (one launch instead of torch::full + gather, cpp:82-85)
template <int VB>
__global__ __launch_bounds__(256) void gather_fill_kernel(const RaggedDesc d, const RowGeom g, const void* __restrict__ src_,
                                                          void* __restrict__ dst_, typename VecOf<VB>::type pattern)
{
    using V = typename VecOf<VB>::type;
    const V* src = static_cast<const V*>(src_);
    V* dst = static_cast<V*>(dst_);
    long long j;
    int v0, lanes;
    if (!row_of_thread(g, d.w_idx, j, v0, lanes)) return;
    for (long long i = blockIdx.y; i < d.batch; i += gridDim.y) {
        const long long a = load_int(d.idx_a, i * d.idx_stride + j, d.idx_i64);
        long long s = -1;
        if (j < load_int(d.counts, i, d.counts_i64)) s = wrap_index(a, d.w_src, d.err);
        const long long from = (i * d.w_src + max(s, 0ll)) * d.row_vecs, to = (i * d.w_idx + j) * d.row_vecs;
        for (long long v = v0; v < d.row_vecs; v += lanes) dst[to + v] = s >= 0 ? src[from + v] : pattern;
    }
}

// dst[i, idx[i,j], :] = pattern
template <int VB>
__global__ __launch_bounds__(256) void insert_const_kernel(const RaggedDesc d, const RowGeom g, void* __restrict__ dst_,
                                                           typename VecOf<VB>::type pattern)
{
    using V = typename VecOf<VB>::type;
    V* dst = static_cast<V*>(dst_);
    long long j;
    int v0, lanes;
    if (!row_of_thread(g, d.w_idx, j, v0, lanes)) return;
    for (long long i = blockIdx.y; i < d.batch; i += gridDim.y) {
        const long long a = load_int(d.idx_a, i * d.idx_stride + j, d.idx_i64);
        if (j >= load_int(d.counts, i, d.counts_i64)) continue;
        const long long o = wrap_index(a, d.w_dst, d.err);
        if (o < 0) continue;
        const long long to = (i * d.w_dst + o) * d.row_vecs;
        for (long long v = v0; v < d.row_vecs; v += lanes) dst[to + v] = pattern;
    }
}

// data[i, j, :] = pattern for j >= counts[i].  The padding of a sample is ONE contiguous span of vectors
// [(i*width + count) * row_vecs, (i+1)*width * row_vecs): blockIdx.x strides over that span from its START, so no
// workgroup is launched only to find its slots filled (the per-slot grid spent most of its time on those: 3.7 TB/s of
// the padded bytes at 1 GB), and the host caps the workgroups per sample so that a large fill loops instead.
template <int VB>
__global__ __launch_bounds__(256) void pad_fill_kernel(void* __restrict__ data_, const void* __restrict__ counts,
                                                       int counts_i64, long long batch, long long width,
                                                       long long row_vecs, typename VecOf<VB>::type pattern)
{
    using V = typename VecOf<VB>::type;
    V* data = static_cast<V*>(data_);
    const long long step = (long long)gridDim.x * 256;
    for (long long i = blockIdx.y; i < batch; i += gridDim.y) {
        const long long c = min(max(load_int(counts, i, counts_i64), 0ll), width);
        const long long end = (i + 1) * width * row_vecs;
        for (long long v = (i * width + c) * row_vecs + (long long)blockIdx.x * 256 + threadIdx.x; v < end; v += step)
            data[v] = pattern;
    }
}

// ---- accumulate: dst[i, out(i,j), k] += src[i, in(i,j), k]
enum AccType { kF32 = 0, kF64 = 1, kI32 = 2, kI64 = 3, kF16 = 4, kBF16 = 5 };

__device__ __forceinline__ float half_bits_to_float(uint16_t h) { return (float)(*reinterpret_cast<const _Float16*>(&h)); }
__device__ __forceinline__ uint16_t float_to_half_bits(float f)
{
    _Float16 h = (_Float16)f;
    return *reinterpret_cast<uint16_t*>(&h);
}
__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // keep NaN a NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool BF>
__device__ __forceinline__ void atomic_add_16(uint16_t* addr, uint16_t val_bits)
{
    uint32_t* word = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(addr) & ~(uintptr_t)3);
    const bool hi = (reinterpret_cast<uintptr_t>(addr) & 2) != 0;
    const float add = BF ? bf16_bits_to_float(val_bits) : half_bits_to_float(val_bits);
    uint32_t old = *word, assumed;
    do {
        assumed = old;
        const uint16_t cur = hi ? (uint16_t)(assumed >> 16) : (uint16_t)(assumed & 0xffffu);
        const float sum = (BF ? bf16_bits_to_float(cur) : half_bits_to_float(cur)) + add;
        const uint16_t nb = BF ? float_to_bf16_bits(sum) : float_to_half_bits(sum);
        const uint32_t repl = hi ? ((assumed & 0x0000ffffu) | ((uint32_t)nb << 16)) : ((assumed & 0xffff0000u) | nb);
        old = atomicCAS(word, assumed, repl);
    } while (old != assumed);
}

template <int ACC>
__global__ __launch_bounds__(256) void accumulate_rows_kernel(const RaggedDesc d, const RowGeom g, const void* __restrict__ src_,
                                                              void* __restrict__ dst_, int pairs)
{
    long long j;
    int v0, lanes;
    if (!row_of_thread(g, d.w_idx, j, v0, lanes)) return;   // row_vecs == elements per row here
    for (long long i = blockIdx.y; i < d.batch; i += gridDim.y) {
        const long long slot = i * d.idx_stride + j;
        const long long a = load_int(d.idx_a, slot, d.idx_i64);
        const long long b = pairs ? load_int(d.idx_b, slot, d.idx_i64) : 0;
        if (j >= load_int(d.counts, i, d.counts_i64)) continue;
        long long s_row, o;
        if (pairs) {
            const long long s = wrap_index(a, d.w_src, d.err);
            o = wrap_index(b, d.w_dst, d.err);
            if (s < 0 || o < 0) continue;
            s_row = (i * d.w_src + s) * d.row_vecs;
        } else {
            o = wrap_index(a, d.w_dst, d.err);
            if (o < 0) continue;
            s_row = (i * d.w_idx + j) * d.row_vecs;
        }
        const long long o_row = (i * d.w_dst + o) * d.row_vecs;
        for (long long k = v0; k < d.row_vecs; k += lanes) {
            const long long s_elem = s_row + k, o_elem = o_row + k;
            if (ACC == kF32)
                atomicAdd(static_cast<float*>(dst_) + o_elem, static_cast<const float*>(src_)[s_elem]);
            else if (ACC == kF64)
                atomicAdd(static_cast<double*>(dst_) + o_elem, static_cast<const double*>(src_)[s_elem]);
            else if (ACC == kI32)
                atomicAdd(static_cast<int*>(dst_) + o_elem, static_cast<const int*>(src_)[s_elem]);
            else if (ACC == kI64)
                atomicAdd(static_cast<unsigned long long*>(dst_) + o_elem,
                          static_cast<const unsigned long long*>(src_)[s_elem]);
            else if (ACC == kF16)
                atomic_add_16<false>(static_cast<uint16_t*>(dst_) + o_elem, static_cast<const uint16_t*>(src_)[s_elem]);
            else
                atomic_add_16<true>(static_cast<uint16_t*>(dst_) + o_elem, static_cast<const uint16_t*>(src_)[s_elem]);
        }
    }
}

// ---- mask -> compact indices: one wave per row, ballot + popcount running offset (order preserving)
__global__ __launch_bounds__(256) void mask_to_indices_kernel(const uint8_t* __restrict__ mask, const void* __restrict__ valid,
                                                              int valid_i64, long long batch, long long width,
                                                              long long* __restrict__ out_idx, long long* __restrict__ out_sizes)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= batch) return;
    long long limit = width;
    if (valid) limit = max(0ll, min(width, load_int(valid, row, valid_i64)));
    const uint8_t* m = mask + row * width;
    long long* o = out_idx + row * width;
    long long offset = 0;
    for (long long base = 0; base < limit; base += 64) {
        const long long j = base + lane;
        const bool on = j < limit && m[j] != 0;
        const unsigned long long b = __ballot(on);
        if (on) o[offset + __popcll(b & ((1ull << lane) - 1ull))] = j;
        offset += __popcll(b);
    }
    for (long long j = offset + lane; j < width; j += 64) o[j] = 0;  // filler, as torch.full(..., 0) in the reference
    if (lane == 0) out_sizes[row] = offset;
}

// ---- few, very wide rows (a dense anchor mask of a small batch): a row is cut into segments of kSeg mask bytes, one
// workgroup per (segment, row) — 8 rows of 65 536 keep 128 workgroups busy instead of 8.  Pass 1 counts the hits per
// segment into a small workspace; pass 2 turns the counts of its row into its base offset (a wave reduction) and writes
// its indices and its share of the zero tail.  Every thread owns 16 consecutive mask bytes (one 16-byte load when the row
// is 16-byte aligned): count, exclusive scan over the 256 threads, then it writes its <= 16 indices in order.
constexpr int kSeg = 4096;

__device__ __forceinline__ unsigned nonzero_bytes16(const uint8_t* m, long long j0, long long limit, bool vec)
{
    // bit k set <=> byte j0 + k is a hit (non-zero and below `limit`)
    unsigned bits = 0;
    if (vec && j0 + 16 <= limit) {
        const uint4 v = *reinterpret_cast<const uint4*>(m + j0);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) bits |= (((w[q] >> (8 * b)) & 0xffu) ? 1u : 0u) << (4 * q + b);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (j0 + k < limit && m[j0 + k] != 0) bits |= 1u << k;
    }
    return bits;
}

__global__ __launch_bounds__(256) void mask_seg_count_kernel(const uint8_t* __restrict__ mask, const void* __restrict__ valid,
                                                             int valid_i64, long long width, int segs, int vec,
                                                             int* __restrict__ seg_counts)
{
    __shared__ int s_w[4];
    const long long row = blockIdx.y;
    long long limit = width;
    if (valid) limit = max(0ll, min(width, load_int(valid, row, valid_i64)));
    const long long j0 = (long long)blockIdx.x * kSeg + (long long)threadIdx.x * 16;
    int c = j0 < limit ? __popc(nonzero_bytes16(mask + row * width, j0, limit, vec != 0)) : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_counts[row * segs + blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void mask_seg_write_kernel(const uint8_t* __restrict__ mask, const void* __restrict__ valid,
                                                             int valid_i64, long long width, int segs, int vec,
                                                             const int* __restrict__ seg_counts, long long* __restrict__ out_idx,
                                                             long long* __restrict__ out_sizes)
{
    __shared__ long long s_base, s_total;
    __shared__ int s_w[4];
    const long long row = blockIdx.y;
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long limit = width;
    if (valid) limit = max(0ll, min(width, load_int(valid, row, valid_i64)));
    if (wave == 0) {  // offsets of this row's segments: hits before this one, and the row total
        long long before = 0, total = 0;
        for (int s0 = 0; s0 < segs; s0 += 64) {
            const int sidx = s0 + lane;
            const long long c = sidx < segs ? seg_counts[row * segs + sidx] : 0;
            long long b = sidx < seg ? c : 0, t = c;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                b += __shfl_xor(b, d);
                t += __shfl_xor(t, d);
            }
            before += b;
            total += t;
        }
        if (lane == 0) {
            s_base = before;
            s_total = total;
        }
    }
    const long long j0 = (long long)seg * kSeg + (long long)tid * 16;
    const unsigned bits = j0 < limit ? nonzero_bytes16(mask + row * width, j0, limit, vec != 0) : 0u;
    const int c = __popc(bits);
    int incl = c;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int wave_before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) wave_before += s_w[w];
    long long* o = out_idx + row * width;
    long long pos = s_base + wave_before + (incl - c);
    unsigned b = bits;
    while (b) {  // this thread's hits, in order
        const int k = __builtin_ctz(b);
        b &= b - 1;
        o[pos++] = j0 + k;
    }
    // zero tail of the row: positions [total, width) that fall into this segment's position range
    const long long total = s_total;
    const long long z0 = max(total, (long long)seg * kSeg), z1 = min(width, (long long)(seg + 1) * kSeg);
    for (long long j = z0 + tid; j < z1; j += 256) o[j] = 0;
    if (seg == 0 && tid == 0) out_sizes[row] = total;
}

// ---- the same in ONE launch (round 3) for rows of up to kOnePassSegs segments: every segment workgroup counts the hits
// of its WHOLE row itself — up to three extra 4 KB mask reads that hit L2 — which yields both the hits in front of its
// segment and the row total (the zero tail needs it) with no second launch and no workspace: 1 x 8192 5.7 -> 3.4 us.  Longer
// rows keep the count launch + write launch: kernel-level they already take 6.2-6.7 us for 8 x 65 536 / 2 x 131 072 (two
// tiny back-to-back launches overlap their launch cost), the redundant counting 8.5 / 12.7 us, and segment workgroups that
// exchange their counts through tagged workspace words in one launch 14.3 us — a cross-workgroup round trip costs as much
// as the launch it saves (profiles/r03_tails_probe_*.log).  What the 12.5 us of round 2 measured was the python operator.
constexpr int kOnePassSegs = 4;

__global__ __launch_bounds__(256) void mask_seg_onepass_kernel(const uint8_t* __restrict__ mask, const void* __restrict__ valid,
                                                               int valid_i64, long long width, int segs, int vec,
                                                               long long* __restrict__ out_idx, long long* __restrict__ out_sizes)
{
    __shared__ int s_w[4], s_before[4], s_total[4];
    const long long row = blockIdx.y;
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long limit = width;
    if (valid) limit = max(0ll, min(width, load_int(valid, row, valid_i64)));
    const uint8_t* mrow = mask + row * width;
    unsigned bits = 0;
    int before = 0, total = 0;
#pragma unroll 4
    for (int s2 = 0; s2 < segs; ++s2) {
        const long long j = (long long)s2 * kSeg + (long long)tid * 16;
        const unsigned b2 = j < limit ? nonzero_bytes16(mrow, j, limit, vec != 0) : 0u;
        const int c2 = __popc(b2);
        total += c2;
        if (s2 < seg) before += c2;
        if (s2 == seg) bits = b2;
    }
    const int c = __popc(bits);
    int incl = c;  // inclusive scan of this segment's hits inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        before += __shfl_xor(before, d);
        total += __shfl_xor(total, d);
    }
    if (lane == 63) s_w[wave] = incl;
    if (lane == 0) {
        s_before[wave] = before;
        s_total[wave] = total;
    }
    __syncthreads();
    int wave_before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) wave_before += s_w[w];
    const long long base = (long long)s_before[0] + s_before[1] + s_before[2] + s_before[3];
    const long long row_total = (long long)s_total[0] + s_total[1] + s_total[2] + s_total[3];
    long long* o = out_idx + row * width;
    const long long j0 = (long long)seg * kSeg + (long long)tid * 16;
    long long pos = base + wave_before + (incl - c);
    while (bits) {  // this thread's hits, in order
        const int k = __builtin_ctz(bits);
        bits &= bits - 1;
        o[pos++] = j0 + k;
    }
    // zero tail of the row: positions [total, width) that fall into this segment's position range
    const long long z0 = max(row_total, (long long)seg * kSeg), z1 = min(width, (long long)(seg + 1) * kSeg);
    for (long long j = z0 + tid; j < z1; j += 256) o[j] = 0;
    if (seg == 0 && tid == 0) out_sizes[row] = row_total;
}

// ---- flat -> padded pack: dst[i, j, :] = flat[offsets[i] + j, :] (j < sizes[i]); padding gets `pattern`
// wide rows: one WORKGROUP of NW waves per row, NW*64 mask bytes per step; the waves' hit counts meet in LDS (double
// buffered: one barrier per step) so that the order-preserving offsets stay exact
template <int NW>
__global__ __launch_bounds__(NW * 64) void mask_to_indices_block_kernel(const uint8_t* __restrict__ mask,
                                                                        const void* __restrict__ valid, int valid_i64,
                                                                        long long width, long long* __restrict__ out_idx,
                                                                        long long* __restrict__ out_sizes)
{
    __shared__ int s_cnt[2][NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row = blockIdx.x;
    long long limit = width;
    if (valid) limit = max(0ll, min(width, load_int(valid, row, valid_i64)));
    const uint8_t* m = mask + row * width;
    long long* o = out_idx + row * width;
    long long offset = 0;
    int step = 0;
    for (long long base = 0; base < limit; base += NW * 64, ++step) {
        const long long j = base + tid;
        const bool on = j < limit && m[j] != 0;
        const unsigned long long b = __ballot(on);
        if (lane == 0) s_cnt[step & 1][wave] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int v = s_cnt[step & 1][w];
            total += v;
            if (w < wave) before += v;
        }
        if (on) o[offset + before + __popcll(b & ((1ull << lane) - 1ull))] = j;
        offset += total;
    }
    for (long long j = offset + tid; j < width; j += NW * 64) o[j] = 0;
    if (tid == 0) out_sizes[row] = offset;
}

template <int VB>
__global__ __launch_bounds__(256) void pack_rows_kernel(const void* __restrict__ flat_, void* __restrict__ dst_,
                                                        const long long* __restrict__ offsets,
                                                        const long long* __restrict__ sizes, long long batch,
                                                        long long width, long long row_vecs, const RowGeom g, int unpack)
{
    using V = typename VecOf<VB>::type;
    long long j;
    int v0, lanes;
    if (!row_of_thread(g, width, j, v0, lanes)) return;
    for (long long i = blockIdx.y; i < batch; i += gridDim.y) {
        const bool valid = j < sizes[i];
        const long long padded = (i * width + j) * row_vecs, flat = (offsets[i] + (valid ? j : 0)) * row_vecs;
        if (!unpack) {
            for (long long v = v0; v < row_vecs; v += lanes)
                static_cast<V*>(dst_)[padded + v] = valid ? static_cast<const V*>(flat_)[flat + v] : V{};
        } else if (valid) {
            for (long long v = v0; v < row_vecs; v += lanes)
                static_cast<V*>(dst_)[flat + v] = static_cast<const V*>(flat_)[padded + v];
        }
    }
}

// lanes per row = smallest power of two >= row_vecs (at most 256); grid = (slot groups, samples)
struct RowLaunch {
    RowGeom geom;
    dim3 grid;
    bool ok;
};
inline RowLaunch row_launch(long long batch, long long width, long long row_vecs)
{
    RowLaunch r{};
    int lg = 0;
    while (lg < 8 && (1ll << lg) < row_vecs) ++lg;
    r.geom.lg = lg;
    const long long rows_per_block = 256 >> lg;
    const long long gx = (width + rows_per_block - 1) / rows_per_block;
    r.ok = gx <= 0x7fffffffll;
    r.grid = dim3((unsigned)(gx > 0 ? gx : 1), (unsigned)(batch < 65535 ? (batch > 0 ? batch : 1) : 65535));
    return r;
}

inline int pick_vec(long long row_bytes, std::initializer_list<const void*> ptrs)
{
    uintptr_t bits = (uintptr_t)row_bytes;
    for (const void* p : ptrs) bits |= reinterpret_cast<uintptr_t>(p);
    if ((bits & 15) == 0) return 16;
    if ((bits & 7) == 0) return 8;
    if ((bits & 3) == 0) return 4;
    if ((bits & 1) == 0) return 2;
    return 1;
}

template <int VB>
typename VecOf<VB>::type make_pattern(uint64_t elem_bits, int elem_size)
{
    alignas(16) unsigned char buf[16];
    for (int b = 0; b < 16; ++b) buf[b] = (unsigned char)((elem_bits >> (8 * (b % elem_size))) & 0xff);
    typename VecOf<VB>::type v;
    __builtin_memcpy(&v, buf, VB);
    return v;
}

int check_desc(const char* who, long long batch, long long w_idx, long long idx_stride, long long row_bytes,
               const void* idx, const void* counts)
{
    if (batch < 0 || w_idx < 0 || row_bytes < 0 || idx_stride < w_idx)
        return accv::fail(ACCV_EINVAL, "%s: invalid extents (batch %lld, slots %lld, stride %lld, row bytes %lld)", who,
                          batch, w_idx, idx_stride, row_bytes);
    if (batch * w_idx * row_bytes > 0 && (!idx || !counts)) return accv::fail(ACCV_EINVAL, "%s: null index/count pointer", who);
    return ACCV_OK;
}

#define DISPATCH_VB(vb, CALL)                   \
    switch (vb) {                               \
        case 16: { constexpr int VB = 16; CALL; } break; \
        case 8: { constexpr int VB = 8; CALL; } break;   \
        case 4: { constexpr int VB = 4; CALL; } break;   \
        case 2: { constexpr int VB = 2; CALL; } break;   \
        default: { constexpr int VB = 1; CALL; } break;  \
    }

int run_copy(int mode, const void* src, void* dst, const void* idx_a, const void* idx_b, const void* counts,
             long long batch, long long w_idx, long long idx_stride, long long w_src, long long w_dst,
             long long row_bytes, int idx_i64, int counts_i64, int* err, hipStream_t stream, const char* who)
{
    if (int rc = check_desc(who, batch, w_idx, idx_stride, row_bytes, idx_a, counts)) return rc;
    if (batch * w_idx * row_bytes == 0) return ACCV_OK;
    if (!src || !dst) return accv::fail(ACCV_EINVAL, "%s: null data pointer", who);
    const int vb = pick_vec(row_bytes, {src, dst});
    RaggedDesc d{idx_a, idx_b, counts, batch, w_idx, idx_stride, w_src, w_dst, row_bytes / vb, idx_i64, counts_i64, err};
    const RowLaunch rl = row_launch(batch, w_idx, d.row_vecs);
    if (!rl.ok) return accv::fail(ACCV_EINVAL, "%s: %lld index slots per sample exceed the grid limit", who, w_idx);
    DISPATCH_VB(vb, {
        if (mode == kGather)
            hipLaunchKernelGGL((copy_rows_kernel<VB, kGather>), rl.grid, dim3(256), 0, stream, d, rl.geom, src, dst);
        else if (mode == kScatter)
            hipLaunchKernelGGL((copy_rows_kernel<VB, kScatter>), rl.grid, dim3(256), 0, stream, d, rl.geom, src, dst);
        else
            hipLaunchKernelGGL((copy_rows_kernel<VB, kMapPairs>), rl.grid, dim3(256), 0, stream, d, rl.geom, src, dst);
    });
    return accv::check_launch(who);
}

}  // namespace

extern "C" {

int accv_ragged_gather(const void* src, void* dst, const void* indices, const void* counts, long long batch,
                       long long w_src, long long w_idx, long long idx_stride, long long row_bytes, int idx_i64,
                       int counts_i64, int* err_counter, void* stream)
{
    return run_copy(kGather, src, dst, indices, nullptr, counts, batch, w_idx, idx_stride, w_src, w_idx, row_bytes,
                    idx_i64, counts_i64, err_counter, static_cast<hipStream_t>(stream), "ragged_gather");
}

int accv_ragged_gather_fill(const void* src, void* dst, const void* indices, const void* counts, long long batch,
                            long long w_src, long long w_idx, long long idx_stride, long long row_bytes,
                            uint64_t fill_bits, int elem_size, int idx_i64, int counts_i64, int* err_counter,
                            void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_desc("ragged_gather_fill", batch, w_idx, idx_stride, row_bytes, indices, counts)) return rc;
    if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8)
        return accv::fail(ACCV_EINVAL, "ragged_gather_fill: unsupported element size %d", elem_size);
    if (batch * w_idx * row_bytes == 0) return ACCV_OK;
    if (!src || !dst) return accv::fail(ACCV_EINVAL, "ragged_gather_fill: null data pointer");
    const int vb = pick_vec(row_bytes, {src, dst});
    if (vb < elem_size) return accv::fail(ACCV_EINVAL, "ragged_gather_fill: data not aligned to its element size");
    RaggedDesc d{indices, nullptr, counts, batch, w_idx, idx_stride, w_src, w_idx, row_bytes / vb, idx_i64, counts_i64,
                 err_counter};
    const RowLaunch rl = row_launch(batch, w_idx, d.row_vecs);
    if (!rl.ok) return accv::fail(ACCV_EINVAL, "ragged_gather_fill: %lld index slots per sample exceed the grid limit", w_idx);
    DISPATCH_VB(vb, hipLaunchKernelGGL((gather_fill_kernel<VB>), rl.grid, dim3(256), 0, stream, d, rl.geom, src, dst,
                                       make_pattern<VB>(fill_bits, elem_size)));
    return accv::check_launch("ragged_gather_fill");
}

int accv_ragged_scatter(const void* src, void* dst, const void* indices, const void* counts, long long batch,
                        long long w_idx, long long idx_stride, long long w_dst, long long row_bytes, int idx_i64,
                        int counts_i64, int* err_counter, void* stream)
{
    return run_copy(kScatter, src, dst, indices, nullptr, counts, batch, w_idx, idx_stride, w_idx, w_dst, row_bytes,
                    idx_i64, counts_i64, err_counter, static_cast<hipStream_t>(stream), "ragged_scatter");
}

int accv_ragged_map_pairs(const void* src, void* dst, const void* src_indices, const void* dst_indices,
                          const void* counts, long long batch, long long w_src, long long w_idx, long long idx_stride,
                          long long w_dst, long long row_bytes, int idx_i64, int counts_i64, int* err_counter,
                          void* stream)
{
    if (!dst_indices && batch * w_idx * row_bytes > 0) return accv::fail(ACCV_EINVAL, "ragged_map_pairs: null target indices");
    return run_copy(kMapPairs, src, dst, src_indices, dst_indices, counts, batch, w_idx, idx_stride, w_src, w_dst,
                    row_bytes, idx_i64, counts_i64, err_counter, static_cast<hipStream_t>(stream), "ragged_map_pairs");
}

int accv_ragged_insert_const(void* dst, const void* indices, const void* counts, long long batch, long long w_idx,
                             long long idx_stride, long long w_dst, long long row_bytes, uint64_t elem_bits,
                             int elem_size, int idx_i64, int counts_i64, int* err_counter, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_desc("ragged_insert_const", batch, w_idx, idx_stride, row_bytes, indices, counts)) return rc;
    if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8)
        return accv::fail(ACCV_EINVAL, "ragged_insert_const: unsupported element size %d", elem_size);
    if (batch * w_idx * row_bytes == 0) return ACCV_OK;
    if (!dst) return accv::fail(ACCV_EINVAL, "ragged_insert_const: null data pointer");
    int vb = pick_vec(row_bytes, {dst});
    if (vb < elem_size) return accv::fail(ACCV_EINVAL, "ragged_insert_const: data not aligned to its element size");
    RaggedDesc d{indices, nullptr, counts, batch, w_idx, idx_stride, 0, w_dst, row_bytes / vb, idx_i64, counts_i64, err_counter};
    const RowLaunch rl = row_launch(batch, w_idx, d.row_vecs);
    if (!rl.ok) return accv::fail(ACCV_EINVAL, "ragged_insert_const: %lld index slots per sample exceed the grid limit", w_idx);
    DISPATCH_VB(vb, hipLaunchKernelGGL((insert_const_kernel<VB>), rl.grid, dim3(256), 0, stream, d, rl.geom, dst,
                                       make_pattern<VB>(elem_bits, elem_size)));
    return accv::check_launch("ragged_insert_const");
}

int accv_ragged_pad_fill(void* data, const void* counts, long long batch, long long width, long long row_bytes,
                         uint64_t elem_bits, int elem_size, int counts_i64, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (batch < 0 || width < 0 || row_bytes < 0) return accv::fail(ACCV_EINVAL, "ragged_pad_fill: negative extent");
    if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8)
        return accv::fail(ACCV_EINVAL, "ragged_pad_fill: unsupported element size %d", elem_size);
    if (batch * width * row_bytes == 0) return ACCV_OK;
    if (!data || !counts) return accv::fail(ACCV_EINVAL, "ragged_pad_fill: null pointer");
    int vb = pick_vec(row_bytes, {data});
    if (vb < elem_size) return accv::fail(ACCV_EINVAL, "ragged_pad_fill: data not aligned to its element size");
    const long long row_vecs = row_bytes / vb;
    // one store per wave while that fills the chip; beyond ~16k workgroups each one strides over its sample's span
    const long long gy = batch < 65535 ? batch : 65535;
    const long long span_blocks = (width * row_vecs + 255) / 256;
    long long gx = 16384 / gy;
    gx = gx < 1 ? 1 : gx;
    gx = span_blocks < gx ? span_blocks : gx;
    DISPATCH_VB(vb, hipLaunchKernelGGL((pad_fill_kernel<VB>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, data,
                                       counts, counts_i64, batch, width, row_vecs, make_pattern<VB>(elem_bits, elem_size)));
    return accv::check_launch("ragged_pad_fill");
}

int accv_ragged_accumulate(const void* src, void* dst, const void* src_indices_or_null, const void* dst_indices,
                           const void* counts, long long batch, long long w_src, long long w_idx, long long idx_stride,
                           long long w_dst, long long row_elems, int acc_dtype, int idx_i64, int counts_i64,
                           int* err_counter, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_desc("ragged_accumulate", batch, w_idx, idx_stride, row_elems, dst_indices, counts)) return rc;
    if (acc_dtype < 0 || acc_dtype > kBF16) return accv::fail(ACCV_EINVAL, "ragged_accumulate: unsupported dtype code %d", acc_dtype);
    if (batch * w_idx * row_elems == 0) return ACCV_OK;
    if (!src || !dst) return accv::fail(ACCV_EINVAL, "ragged_accumulate: null data pointer");
    const int pairs = src_indices_or_null != nullptr;
    RaggedDesc d{pairs ? src_indices_or_null : dst_indices, pairs ? dst_indices : nullptr, counts, batch, w_idx, idx_stride,
                 w_src, w_dst, row_elems, idx_i64, counts_i64, err_counter};
    const RowLaunch rl = row_launch(batch, w_idx, row_elems);
    if (!rl.ok) return accv::fail(ACCV_EINVAL, "ragged_accumulate: %lld index slots per sample exceed the grid limit", w_idx);
#define ACC_CASE(A) case A: hipLaunchKernelGGL((accumulate_rows_kernel<A>), rl.grid, dim3(256), 0, stream, d, rl.geom, src, dst, pairs); break;
    switch (acc_dtype) {
        ACC_CASE(kF32) ACC_CASE(kF64) ACC_CASE(kI32) ACC_CASE(kI64) ACC_CASE(kF16) ACC_CASE(kBF16)
    }
    return accv::check_launch("ragged_accumulate");
}

size_t accv_ragged_mask_to_indices_workspace_bytes(long long batch, long long width)
{
    // the segmented two-pass path pays for few, very wide rows; everything else needs no workspace
    if (batch <= 0 || width < 2 * kSeg || batch > 1024) return 0;
    const long long segs = (width + kSeg - 1) / kSeg;
    if (segs > 65535 || batch > 65535) return 0;
    return (size_t)batch * (size_t)segs * sizeof(int);
}

int accv_ragged_mask_to_indices_ws(const void* mask_u8, const void* valid_counts_or_null, int valid_i64, long long batch,
                                   long long width, long long* out_indices, long long* out_sizes, void* workspace,
                                   size_t workspace_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (batch < 0 || width < 0) return accv::fail(ACCV_EINVAL, "ragged_mask_to_indices: negative extent");
    if (batch == 0) return ACCV_OK;
    if (!out_sizes || (width > 0 && (!mask_u8 || !out_indices)))
        return accv::fail(ACCV_EINVAL, "ragged_mask_to_indices: null pointer");
    const uint8_t* m = static_cast<const uint8_t*>(mask_u8);
    const size_t need = accv_ragged_mask_to_indices_workspace_bytes(batch, width);
    if (need > 0 && (width + kSeg - 1) / kSeg <= kOnePassSegs) {   // few segments per row: one launch, no workspace
        const int segs = (int)((width + kSeg - 1) / kSeg);
        const int vec = ((reinterpret_cast<uintptr_t>(m) & 15u) == 0 && width % 16 == 0) ? 1 : 0;
        hipLaunchKernelGGL(mask_seg_onepass_kernel, dim3((unsigned)segs, (unsigned)batch), dim3(256), 0, stream, m,
                           valid_counts_or_null, valid_i64, width, segs, vec, out_indices, out_sizes);
        return accv::check_launch("ragged_mask_to_indices (segmented, one pass)");
    }
    if (need > 0 && workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 3u) == 0) {
        const int segs = (int)((width + kSeg - 1) / kSeg);
        const int vec = ((reinterpret_cast<uintptr_t>(m) & 15u) == 0 && width % 16 == 0) ? 1 : 0;
        const dim3 grid((unsigned)segs, (unsigned)batch), block(256);
        int* counts = static_cast<int*>(workspace);
        hipLaunchKernelGGL(mask_seg_count_kernel, grid, block, 0, stream, m, valid_counts_or_null, valid_i64, width, segs, vec, counts);
        hipLaunchKernelGGL(mask_seg_write_kernel, grid, block, 0, stream, m, valid_counts_or_null, valid_i64, width, segs, vec,
                           counts, out_indices, out_sizes);
        return accv::check_launch("ragged_mask_to_indices (segmented)");
    }
    if (width > 512 && batch <= 0x7fffffff) {  // wide rows: a workgroup per row (16x / 4x the lanes of a wave per row)
        if (width > 4096)
            hipLaunchKernelGGL((mask_to_indices_block_kernel<16>), dim3((unsigned)batch), dim3(1024), 0, stream, m,
                               valid_counts_or_null, valid_i64, width, out_indices, out_sizes);
        else
            hipLaunchKernelGGL((mask_to_indices_block_kernel<4>), dim3((unsigned)batch), dim3(256), 0, stream, m,
                               valid_counts_or_null, valid_i64, width, out_indices, out_sizes);
        return accv::check_launch("ragged_mask_to_indices");
    }
    const unsigned grid = (unsigned)((batch + 3) / 4);
    hipLaunchKernelGGL(mask_to_indices_kernel, dim3(grid), dim3(256), 0, stream, m, valid_counts_or_null, valid_i64, batch,
                       width, out_indices, out_sizes);
    return accv::check_launch("ragged_mask_to_indices");
}

int accv_ragged_mask_to_indices(const void* mask_u8, const void* valid_counts_or_null, int valid_i64, long long batch,
                                long long width, long long* out_indices, long long* out_sizes, void* stream_)
{
    return accv_ragged_mask_to_indices_ws(mask_u8, valid_counts_or_null, valid_i64, batch, width, out_indices, out_sizes, nullptr,
                                          0, stream_);
}

int accv_ragged_pack(const void* flat, void* padded, const long long* offsets, const long long* sizes, long long batch,
                     long long width, long long row_bytes, int unpack, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (batch < 0 || width < 0 || row_bytes < 0) return accv::fail(ACCV_EINVAL, "ragged_pack: negative extent");
    if (batch * width * row_bytes == 0) return ACCV_OK;
    if (!flat || !padded || !offsets || !sizes) return accv::fail(ACCV_EINVAL, "ragged_pack: null pointer");
    const int vb = pick_vec(row_bytes, {flat, padded});
    const long long row_vecs = row_bytes / vb;
    const RowLaunch rl = row_launch(batch, width, row_vecs);
    if (!rl.ok) return accv::fail(ACCV_EINVAL, "ragged_pack: width %lld exceeds the grid limit", width);
    const void* in = unpack ? padded : flat;
    void* out = unpack ? const_cast<void*>(flat) : padded;
    DISPATCH_VB(vb, hipLaunchKernelGGL((pack_rows_kernel<VB>), rl.grid, dim3(256), 0, stream, in, out, offsets, sizes,
                                       batch, width, row_vecs, rl.geom, unpack));
    return accv::check_launch("ragged_pack");
}
}
