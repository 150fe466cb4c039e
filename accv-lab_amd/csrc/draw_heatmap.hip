// H1 — Gaussian heat-map rasteriser for gfx950 (MI355X), written as a GATHER-BY-TILE:
//
//   * the frame is cut into wave tiles of (32*PX) columns x (2*R) rows; one 64-lane wavefront owns one
//     tile, lane l holds PX consecutive pixels of rows {i, i+R} (half-wave each) in registers;
//   * the wave culls the plane's objects against its tile with one __ballot per 64 candidates, compacts the
//     hits into LDS with a popcount prefix (no atomics, no pre-pass, no workspace for the batched API);
//   * exp(-(dx^2+dy^2)/var) is evaluated separably: a per-tile LDS table of k*exp(-dy^2/var) per (hit,row)
//     and PX in-register column factors per hit, so the inner loop is one multiply + one max per pixel;
//     "outside the object's clipped box" is encoded as NaN in either factor — fmaxf(acc, NaN) == acc —
//     which reproduces the reference's write extent exactly for any sign of k and any base value;
//   * max is order independent, so the result is deterministic and equals the reference's atomicMax result;
//   * each pixel is written exactly once with 16-byte stores (fused-clear mode) or read-max-written once
//     (in-place mode, only tiles that are touched).
//
// Kernels in this file (all share locate_tile / cull_round / make_hit):
//   splat_kernel        the tile kernel above = splat_body<PX,R,CLEAR,SM,WPG,SRC=0> (headline: <4,8,true,0,1>, in place <4,8,false,5,1>)
//   splat_multi_kernel  the same body over the tiles of up to four scales in one launch, objects given as float
//                       centres / boxes and converted per scale inside the cull (SRC=1)
//   splat_small_kernel  point-like objects (ACCV_HM_SMALL_RADII): tile in LDS, lanes walk each hit's box, ds_max_f32
//   splat_points_multi_kernel   lane raster: sampled polyline points of all scales, two-level cull (group boxes), same LDS tile
//   lane_raster_multi_kernel    lane raster of sparse lane sets in ONE launch: the tile waves sample the polylines themselves
//   splat_multi_sampler_kernel  splat_multi_kernel with the polyline sampler riding in the launch (draw_targets_multiscale)
//   bin_* kernels       flat API: counting sort of the objects by plane into plane-sorted copies
//   targets_from_*      float boxes / sampled polyline points -> integer centre + radius
//
// Replaces: packages/draw_heatmap/accvlab/draw_heatmap/include/draw_heatmap_cuda_kernel.cuh:26-108 and
// csrc/draw_heatmap_cuda.cu:29-165 of the reference (one thread per object, serial atomicMax splat).
// Plane offsets are 64-bit (the reference's are int and overflow for class-wise full-HD batches).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstring>

#include "accv_common.h"
#include "polyline_arith.h"

#ifndef ACCV_BOX_TILE_R
#define ACCV_BOX_TILE_R 8   // rows per half-wave of a box-map tile (multi-scale launches): 128 x 16 pixel tiles.  4 = 128 x 8 tiles
                            // (54 VGPRs, 8 waves per SIMD, twice the waves): box maps of config 3 17.1 -> 18.4 us, measured
#endif

namespace {

constexpr int kWavesPerGroup = 1;  // 1 wave per workgroup measured 6.5 % faster than 4 (profiles/r01_h1_variants_wpg.log)
constexpr int kCand = 64;  // candidates per cull round = one per lane
constexpr float kLog2e = 1.4426950408889634f;

struct SplatParams {
    float* hm;
    const int32_t* centers;
    const int32_t* radii;
    const int32_t* labels;     // class-wise batched: labels; otherwise any readable int32 array shaped like radii
    const void* counts;        // batched: i32[B] or i64[B]
    const int32_t* plane_off;  // flat: [P+1] offsets into centers/radii (which are then the plane-sorted copies)
    int H, W;
    int n_max;      // batched: padded objects per sample
    int n_classes;  // class-wise: C, else 0
    int tiles_x, tiles_y;
    long long n_tiles;
    float factor, k;
    int counts_i64;
    int grid3d;           // tile index comes from a 3-D grid instead of a linear block index
    float dense_area;     // SM == 5: a plane whose objects cover at least this many pixels (sum (2r+1)^2) stores write-through
    // multi-scale front end (SRC == 1): objects are float centres / boxes in source pixels, converted per scale
    const float* centers_f;  // [B, n_max, 2] (x, y)
    const float* boxes_f;    // [B, n_max, 4] (x0, y0, x1, y1)
    float stride;
    // point splats (SRC == 2): centers_f = sampled points [B, n_max, 2]; boxes_f = bounding boxes of every 64 consecutive
    // points [B, n_groups, 4] (xmin, ymin, xmax, ymax; source pixels); every point gets the same radius
    int radius, n_groups;
};

constexpr int kMaxScales = 4;
struct MultiParams {
    SplatParams scale[kMaxScales];
    long long tile_begin[kMaxScales + 1];  // linear workgroup index where each scale's tiles start
    int n_scales;
};

// one culled hit, read back as a single ds_read_b128 broadcast.  The clipped box is stored relative to the tile and
// clamped to it (each bound fits a byte: tiles are at most 128 x 32), so a wave needs 1 KB for the list instead of 2
// and 5 KB of LDS in total -> 32 single-wave workgroups (8 waves per SIMD) fit a CU's 160 KB
struct __attribute__((aligned(16))) Hit {
    int x, y;
    float c2;      // log2(e) / var
    unsigned box;  // xlo | xhi << 8 | ylo << 16 | yhi << 24 : columns [xlo,xhi), rows [ylo,yhi) of the tile
};

template <int PX>
struct Vec;
typedef float vfloat4 __attribute__((ext_vector_type(4)));
template <>
struct Vec<4> {
    using type = vfloat4;
};
template <>
struct Vec<1> {
    using type = float;
};

__device__ __forceinline__ float raw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max that treats a quiet NaN as "no data" in ONE instruction.  fmaxf() has the same meaning, but hipcc puts a
// canonicalising v_max_f32 x, x, x in front of every call whose operand it cannot prove to be a quiet value (the
// loop-carried accumulator), i.e. 3 VALU ops per pixel instead of 2.  All masks in this file are the quiet NaN
// 0x7fc00000 and products of a quiet NaN stay quiet, so the raw instruction is exact here.
__device__ __forceinline__ float max_skip_nan(float acc, float v)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(acc), "v"(v));
    return r;
}

// ---------------------------------------------------------------- pieces shared by the tile kernels
struct TileCtx {
    int tx0, ty0, tx1, ty1;  // pixel bounds of the tile, clipped to the frame
    long long plane;
    const int2* centers2;    // objects of this plane: [0, n)
    const int32_t* radii;
    const int32_t* labels;
    const float2* centers_f;  // SRC == 1: float objects of this plane
    const float4* boxes_f;
    float4 box0;             // SRC == 2 (SCALAR_COUNT): group box min(lane, n_groups - 1) of the plane, requested ahead of the count
    float stride;
    int radius;              // SRC == 2: the radius of every point
    int n, cls;              // cls < 0: no class filter
};

template <bool SCALAR_COUNT = false>
__device__ __forceinline__ void plane_objects(const SplatParams& p, TileCtx& t);

// tile coordinates from the launch geometry + the object range that feeds this plane; false = wave has no tile
template <int TW, int TH, int WPG, bool SCALAR_COUNT = false>
__device__ __forceinline__ bool locate_tile(const SplatParams& p, int wave, TileCtx& t, long long linear_group)
{
    int tx, ty;
    if (p.grid3d) {
        // 3-D grid (x = group of WPG column tiles, y = row tile, z = plane): no divisions in the prologue
        tx = blockIdx.x * WPG + wave;
        ty = blockIdx.y;
        t.plane = blockIdx.z;
        if (tx >= p.tiles_x) return false;
    } else {  // linear block index (more than 65535 planes or tile rows)
        const long long tile = linear_group * WPG + wave;
        if (tile >= p.n_tiles) return false;  // whole wave exits; waves never synchronise with each other
        if (p.n_tiles <= 0x7fffffffll) {      // 32-bit divisions (a 64-bit one costs ~60 instructions, and there are three)
            const unsigned t32 = (unsigned)tile, t2 = t32 / (unsigned)p.tiles_x;
            tx = (int)(t32 - t2 * (unsigned)p.tiles_x);
            const unsigned pl = t2 / (unsigned)p.tiles_y;
            ty = (int)(t2 - pl * (unsigned)p.tiles_y);
            t.plane = pl;
        } else {
            tx = (int)(tile % p.tiles_x);
            const long long t2 = tile / p.tiles_x;
            ty = (int)(t2 % p.tiles_y);
            t.plane = t2 / p.tiles_y;
        }
    }
    t.tx0 = tx * TW;
    t.ty0 = ty * TH;
    t.tx1 = min(t.tx0 + TW, p.W);
    t.ty1 = min(t.ty0 + TH, p.H);
    plane_objects<SCALAR_COUNT>(p, t);
    return true;
}

// which objects feed plane t.plane: objects [obj_base, obj_base + n) of centers/radii(/labels)
// SCALAR_COUNT: the plane's count is read through the scalar cache (constant address space: the array is not written by this
// launch, and the scalar cache is invalidated between launches).  hipcc chooses that by itself while the parameters are read
// straight from the kernel arguments, but falls back to a vector-memory load — ten times the latency, in front of everything a
// short-lived tile wave does — once they are a preloaded copy (preload_params)
template <bool SCALAR_COUNT>
__device__ __forceinline__ void plane_objects(const SplatParams& p, TileCtx& t)
{
    long long obj_base;
    t.cls = -1;
    if (p.plane_off) {  // flat API: the binning pre-pass left plane-sorted copies of the objects
        const int o0 = p.plane_off[t.plane];
        obj_base = o0;
        t.n = p.plane_off[t.plane + 1] - o0;
    } else {
        long long s = t.plane;
        if (p.n_classes > 0) {
            s = t.plane / p.n_classes;
            t.cls = (int)(t.plane - s * p.n_classes);
        }
        long long cnt;
        if constexpr (SCALAR_COUNT) {
            // the first round of group boxes does not depend on the count: requested first, so that both are in flight together
            if (p.n_groups > 0)
                t.box0 = reinterpret_cast<const float4*>(p.boxes_f)[t.plane * p.n_groups + min((int)(threadIdx.x & 63), p.n_groups - 1)];
            using ConstI32 = const __attribute__((address_space(4))) int;
            using ConstI64 = const __attribute__((address_space(4))) long long;
            const uintptr_t base = reinterpret_cast<uintptr_t>(p.counts);
            cnt = p.counts_i64 ? reinterpret_cast<ConstI64*>(base)[s] : (long long)reinterpret_cast<ConstI32*>(base)[s];
        } else {
            cnt = p.counts_i64 ? ((const long long*)p.counts)[s] : (long long)((const int*)p.counts)[s];
        }
        t.n = (int)max(0ll, min(cnt, (long long)p.n_max));
        obj_base = s * p.n_max;
    }
    t.centers2 = reinterpret_cast<const int2*>(p.centers) + obj_base;
    t.radii = p.radii + obj_base;
    t.labels = p.labels + obj_base;
    t.centers_f = reinterpret_cast<const float2*>(p.centers_f) + obj_base;
    t.boxes_f = reinterpret_cast<const float4*>(p.boxes_f) + (p.n_groups > 0 ? (t.plane * p.n_groups) : obj_base);
    t.stride = p.stride;
    t.radius = p.radius;
}

// One candidate per lane and round, fetched with branch-free loads (index clamped to the last object, result masked).
// The cull is VALU-bound for long object lists (lane rasters walk 10^3 candidates per tile), so the test every lane
// runs is a cheap CONSERVATIVE one in 32-bit: coordinates clamped to +-2^29 and the radius to 2^30 cannot overflow and
// never miss a real hit while H, W <= 2^29 (host-checked).  Returns the ballot of hitting lanes.
struct Cand {
    int x, y, r, label;
};
template <int SRC = 0>
__device__ __forceinline__ Cand cull_load(const TileCtx& t, int base, int lane)
{
    const int cc = min(base + lane, t.n - 1);  // n >= 1 inside the candidate loop
    if constexpr (SRC == 1) {
        // float centre + box in source pixels -> integer target at this scale, exactly targets_from_boxes_kernel below
        // (packages/draw_heatmap/tests/_test_helpers.py:20-28): r = max(1, ceil(min edge distance / stride)),
        // c = int(c / stride); IEEE division
        const float2 c = t.centers_f[cc];
        const float4 b = t.boxes_f[cc];
        const float m = fminf(fminf(c.x - b.x, c.y - b.y), fminf(b.z - c.x, b.w - c.y));
        // (a stride that is a power of two: the product with its reciprocal is the same correctly rounded value as the IEEE
        // division — both round x * 2^-k once — for a third of the instructions of this cull)
        const bool pow2 = (__float_as_uint(t.stride) & 0x007fffffu) == 0u && t.stride > 1.0e-30f && t.stride < 1.0e30f;   // uniform
        if (pow2) {
            const float inv = 1.0f / t.stride;
            int r = (int)ceilf(m * inv);
            if (r < 1) r = 1;
            return Cand{(int)(c.x * inv), (int)(c.y * inv), r, 0};
        }
        int r = (int)ceilf(__fdiv_rn(m, t.stride));
        if (r < 1) r = 1;
        return Cand{(int)__fdiv_rn(c.x, t.stride), (int)__fdiv_rn(c.y, t.stride), r, 0};
    } else if constexpr (SRC == 2) {
        // sampled polyline point -> target of the common radius, exactly targets_from_points_kernel below
        const float2 c = t.centers_f[cc];
        const bool bad = (c.x != c.x) || (c.y != c.y);
        if (bad) return Cand{0, 0, -1, 0};
        return Cand{(int)__fdiv_rn(c.x, t.stride), (int)__fdiv_rn(c.y, t.stride), t.radius, 0};
    } else {
        const int2 cxy = t.centers2[cc];
        return Cand{cxy.x, cxy.y, t.radii[cc], t.labels[cc]};
    }
}
__device__ __forceinline__ unsigned long long cull_test(const TileCtx& t, int base, int lane, const Cand& c)
{
    constexpr int kClampXY = 1 << 29, kClampR = 1 << 30;
    const int xc = min(max(c.x, -kClampXY), kClampXY), yc = min(max(c.y, -kClampXY), kClampXY);
    const int rc = min(c.r, kClampR);
    const bool hit = (base + lane < t.n) && (t.cls < 0 || c.label == t.cls) && c.r >= 0 && xc - rc < t.tx1 &&
                     xc + rc >= t.tx0 && yc - rc < t.ty1 && yc + rc >= t.ty0;
    return __ballot(hit);
}
template <int SRC = 0>
__device__ __forceinline__ unsigned long long cull_round(const TileCtx& t, int base, int lane, int& x, int& y, int& r)
{
    const Cand c = cull_load<SRC>(t, base, lane);
    x = c.x;
    y = c.y;
    r = c.r;
    return cull_test(t, base, lane, c);
}

// hit record of a lane that passed the cull: the exact clipped box of the reference (left/right/top/bottom,
// cuh:64-67, 92-95; 64-bit), relative to the tile and clamped to it; an empty exact box masks every pixel
__device__ __forceinline__ float hit_exponent_scale(const SplatParams& p, int r)   // log2(e) / (2 sigma^2), sigma = diameter / factor
{
    const float sigma = (float)(2 * r + 1) / p.factor;
    return kLog2e / (2.0f * sigma * sigma);
}
__device__ __forceinline__ Hit make_hit(const SplatParams& p, const TileCtx& t, int x, int y, int r, float c2);
__device__ __forceinline__ Hit make_hit(const SplatParams& p, const TileCtx& t, int x, int y, int r)
{
    return make_hit(p, t, x, y, r, hit_exponent_scale(p, r));
}
// (c2 given: point splats share one radius, and the two IEEE divisions behind it were repeated per lane and fetched group)
__device__ __forceinline__ Hit make_hit(const SplatParams& p, const TileCtx& t, int x, int y, int r, float c2)
{
    const long long x0 = (long long)x - min(x, r), x1 = (long long)x + min((long long)p.W - x, (long long)r + 1);
    const long long y0 = (long long)y - min(y, r), y1 = (long long)y + min((long long)p.H - y, (long long)r + 1);
    const long long xlo = max(x0, (long long)t.tx0) - t.tx0, xhi = min(x1, (long long)t.tx1) - t.tx0;
    const long long ylo = max(y0, (long long)t.ty0) - t.ty0, yhi = min(y1, (long long)t.ty1) - t.ty0;
    unsigned box = 0;  // empty: only possible for coordinates beyond the clamps of the cull
    if (xhi > xlo && yhi > ylo) box = (unsigned)xlo | ((unsigned)xhi << 8) | ((unsigned)ylo << 16) | ((unsigned)yhi << 24);
    return Hit{x, y, c2, box};
}

// the same for two candidates at once (v_max3_f32 follows the same NaN rule: NaN operands are skipped)
__device__ __forceinline__ float max3_skip_nan(float acc, float a, float b)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(acc), "v"(a), "v"(b));
    return r;
}

template <int PX, int R, bool CLEAR, int SM, int WPG, int SRC>
__device__ __forceinline__ void splat_body(const SplatParams& p, long long linear_group)
{
    constexpr int kWavesPerGroup = WPG;  // shadows the namespace constant inside the body
    constexpr int TW = 32 * PX;  // 32 lanes side by side cover one row segment of the tile
    constexpr int TH = 2 * R;    // the two half-waves take R rows each
    static_assert(R % 4 == 0, "row registers are fetched four at a time");

    __shared__ Hit s_hit[kWavesPerGroup][kCand];
    __shared__ __attribute__((aligned(16))) float s_ey[kWavesPerGroup][kCand][TH];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileCtx t;
    if (!locate_tile<TW, TH, kWavesPerGroup>(p, wave, t, linear_group)) return;
    const int tx0 = t.tx0, ty0 = t.ty0, n = t.n;
    const long long plane = t.plane;

    const int sub = lane >> 5;  // which half-wave: rows [sub*R, sub*R + R) of the tile
    const int col0 = tx0 + (lane & 31) * PX;

    float acc[R][PX];
    const float init = CLEAR ? 0.0f : __builtin_nanf("");
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int c = 0; c < PX; ++c) acc[i][c] = init;

    int total_hits = 0;
    float cover = 0.0f;  // SM == 5: this lane's share of sum (2r+1)^2 over the plane's objects (density estimate)

    for (int base = 0; base < n; base += kCand) {
        // ---- cull: conservative 32-bit test, ballot, popcount-prefix compaction into LDS
        int x, y, r;
        const unsigned long long m = cull_round<SRC>(t, base, lane, x, y, r);
        if constexpr (SM == 5) {
            const float dia = (float)(2 * min(max(r, 0), 1 << 20) + 1);
            cover += (base + lane < n) ? dia * dia : 0.0f;
        }
        const bool hit = (m >> lane) & 1ull;
        const int nh = __popcll(m);
        if (nh == 0) continue;
        if constexpr (!CLEAR) {
            // in-place: the tile is touched -> fetch its current content into the accumulators NOW, so the load
            // latency hides behind the table and accumulate phases (max is order independent)
            if (total_hits == 0 && col0 < p.W) {
                const float* plane_rd = p.hm + (size_t)plane * (size_t)p.H * (size_t)p.W;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int row = ty0 + sub * R + i;
                    if (row < p.H) {
                        if constexpr (PX == 4) {
                            const vfloat4 o = *reinterpret_cast<const vfloat4*>(plane_rd + (size_t)row * p.W + col0);
                            acc[i][0] = o.x;
                            acc[i][1] = o.y;
                            acc[i][2] = o.z;
                            acc[i][3] = o.w;
                        } else {
                            acc[i][0] = plane_rd[(size_t)row * p.W + col0];
                        }
                    }
                }
            }
        }
        if (hit) s_hit[wave][__popcll(m & ((1ull << lane) - 1ull))] = make_hit(p, t, x, y, r);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- row-factor table: ey[h][row] = k * exp(-dy^2/var), NaN outside the clipped rows
        for (int t = lane; t < nh * TH; t += 64) {
            const int h = t / TH, rr = t % TH;
            const Hit hy = s_hit[wave][h];
            const float d = (float)(ty0 + rr - hy.y);
            const float v = p.k * raw_exp2(-(d * d) * hy.c2);
            const unsigned ylo = (hy.box >> 16) & 255u, yhi = hy.box >> 24;
            s_ey[wave][h][rr] = ((unsigned)rr >= ylo && (unsigned)rr < yhi) ? v : __builtin_nanf("");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- accumulate: per hit PX column factors in registers, row factors from LDS.  Hits are taken two at a time:
        // acc = max3(acc, ex_a * ey_a, ex_b * ey_b) is 3 VALU ops per pixel for two hits instead of 4
        const unsigned colr0 = (unsigned)(lane & 31) * PX;  // first column of this lane, tile relative
        auto column_factors = [&](const Hit& hx, float (&ex)[PX]) {
            const unsigned xlo = hx.box & 255u, xhi = (hx.box >> 8) & 255u;
#pragma unroll
            for (int c = 0; c < PX; ++c) {
                const float d = (float)(col0 + c - hx.x);
                const float e = raw_exp2(-(d * d) * hx.c2);
                ex[c] = (colr0 + c >= xlo && colr0 + c < xhi) ? e : __builtin_nanf("");
            }
        };
        int h = 0;
        // (fused-clear instantiations only: the in-place ones are register bound, and the second set of factors costs
        // them a wave of occupancy — sparse in-place launches lost 5 %)
        for (; CLEAR && h + 1 < nh; h += 2) {
            float exa[PX], exb[PX];
            column_factors(s_hit[wave][h], exa);
            column_factors(s_hit[wave][h + 1], exb);
#pragma unroll
            for (int q = 0; q < R / 4; ++q) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_ey[wave][h][sub * R + 4 * q]);
                const float4 b4 = *reinterpret_cast<const float4*>(&s_ey[wave][h + 1][sub * R + 4 * q]);
                const float eya[4] = {a4.x, a4.y, a4.z, a4.w}, eyb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < PX; ++c)
                        acc[4 * q + i][c] = max3_skip_nan(acc[4 * q + i][c], exa[c] * eya[i], exb[c] * eyb[i]);
            }
        }
        for (; h < nh; ++h) {
            float ex[PX];
            column_factors(s_hit[wave][h], ex);
#pragma unroll
            for (int q = 0; q < R / 4; ++q) {
                const float4 e4 = *reinterpret_cast<const float4*>(&s_ey[wave][h][sub * R + 4 * q]);
                const float ey[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < PX; ++c) acc[4 * q + i][c] = max_skip_nan(acc[4 * q + i][c], ex[c] * ey[i]);
            }
        }
        total_hits += nh;
        // the next round overwrites the LDS lists: order it behind this round's reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    if (!CLEAR && total_hits == 0) return;  // in-place: untouched tile costs no HBM traffic
    if (col0 >= p.W) return;                 // PX == 4 requires W % 4 == 0, so a lane is all-in or all-out

    using V = typename Vec<PX>::type;
    float* plane_ptr = p.hm + (size_t)plane * (size_t)p.H * (size_t)p.W;
    // SM == 5 (in-place launches): the store policy is chosen PER PLANE from the density of its objects, known to the
    // wave for free after its cull loop: sum (2r+1)^2 over the plane's objects relative to the plane's area.  Planes that
    // are covered about once or more rewrite most of their tiles -> write-through non-temporal stores (sc1 nt: -6 % on
    // the dense rule-A batch); sparse planes touch a few tiles that the next consumer finds in L2 / Infinity Cache ->
    // plain stores (write-through costs them 27 %, profiles/r01_h1_ab_rows_store_policy.log).  Same values either way.
    bool write_through = SM == 2 || SM == 4;
    if constexpr (SM == 5 && PX == 4) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cover += __shfl_xor(cover, d);
        write_through = cover >= p.dense_area;   // wave-uniform
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = ty0 + sub * R + i;
        if (row >= p.H) break;
        V* dst = reinterpret_cast<V*>(plane_ptr + (size_t)row * p.W + col0);
        V out;
        if constexpr (PX == 4)
            out = V{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};  // in-place mode: acc already holds max(old, splats)
        else
            out = acc[i][0];
        if constexpr (SM == 1) {
            __builtin_nontemporal_store(out, dst);
        } else if constexpr (SM >= 2 && PX == 4) {
            // write-through (sc1) / write-through non-temporal (sc1 nt) 16-byte buffer store
            constexpr int aux = SM == 2 ? 16 : 18;  // sc1 | sc1+nt
            if (write_through) {
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(plane_ptr, 0, (int)((size_t)p.H * p.W * 4), 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(out, rsrc, (int)(((size_t)row * p.W + col0) * 4), 0, aux);
            } else {
                *dst = out;
            }
        } else {
            *dst = out;
        }
    }
}

template <int PX, int R, bool CLEAR, int SM, int WPG = kWavesPerGroup>
__global__ __launch_bounds__(WPG * 64) void splat_kernel(const SplatParams p)
{
    splat_body<PX, R, CLEAR, SM, WPG, 0>(p, blockIdx.x);
}

// ---------------------------------------------------------------- multi-scale: all strides of one batch in ONE launch
// A detection head wants the same objects rasterised at several strides (config 3: 4 / 8 / 16).  Per scale that is a
// target-prep launch plus a splat launch of a map of a few MB — launch bound.  Here one grid covers the tiles of every
// scale; a workgroup finds its scale from the tile prefix (wave-uniform), and the candidates are the FLOAT centres and
// boxes in source pixels, converted to that scale's integer centre / radius inside the cull (same arithmetic as
// targets_from_boxes_kernel), so the front end needs no launch and no intermediate tensors at all.
// Prologue of the multi-scale kernels.  Their tiles are short-lived waves (most of a lane raster's tiles store zeros and leave),
// and what such a wave did first was a CHAIN of dependent scalar loads: number of scales -> prefix entry after prefix entry ->
// one field of the scale's parameters, a branch, the next field ... (~20 load / wait pairs in the ISA of round 3's point splat).
// Here the scale comes from the whole prefix at once, without a branch (entries past the last scale hold the total: never
// matched), and the scale's parameters are copied in one batch of loads, held there by empty asm statements.
__device__ __forceinline__ int scale_of_group(const MultiParams& mp, long long group, long long& first)
{
    int s = 0;
    first = 0;   // tile_begin[0]
#pragma unroll
    for (int i = 1; i < kMaxScales; ++i) {
        const long long begin = mp.tile_begin[i];
        if (group >= begin) {
            s = i;
            first = begin;
        }
    }
    return s;
}
template <typename T>
__device__ __forceinline__ void pin_scalar(T& v)
{
    asm volatile("" : "+s"(v));
}
__device__ __forceinline__ SplatParams preload_params(const SplatParams& src)
{
    SplatParams p = src;
    // (the pointers are left alone: behind an asm statement hipcc no longer knows them to be global memory and turns every load
    // through them — the frame's count, a scalar load before — into a flat VMEM load)
    asm volatile("" ::"s"(p.hm), "s"(p.counts), "s"(p.centers_f), "s"(p.boxes_f));   // (requested with the rest, value untouched)
    pin_scalar(p.H);
    pin_scalar(p.W);
    pin_scalar(p.n_max);
    pin_scalar(p.n_classes);
    pin_scalar(p.tiles_x);
    pin_scalar(p.tiles_y);
    pin_scalar(p.n_tiles);
    pin_scalar(p.factor);
    pin_scalar(p.k);
    pin_scalar(p.counts_i64);
    pin_scalar(p.grid3d);
    pin_scalar(p.dense_area);
    pin_scalar(p.stride);
    pin_scalar(p.radius);
    pin_scalar(p.n_groups);
    return p;
}

template <bool CLEAR, int SM>
__global__ __launch_bounds__(64) void splat_multi_kernel(const MultiParams mp)
{
    long long first;
    const int s = scale_of_group(mp, blockIdx.x, first);
    // (the scale's parameters are NOT requested up front here, neither as copies (preload_params) nor as asm inputs: either way
    // the register-bound tile body goes from 79 to 85-124 VGPRs and loses one or two waves per SIMD)
    splat_body<4, ACCV_BOX_TILE_R, CLEAR, SM, 1, 1>(mp.scale[s], (long long)blockIdx.x - first);
}

// ---------------------------------------------------------------- small splats (lane rasters, point-like targets)
// The tile kernel above pays a full 128x16-pixel register update per hit, whatever the size of the object's box.  For
// boxes of a few pixels (a lane sample of radius 2 covers 5x5) that is 2048 pixel updates for 25 useful ones, and
// tiles that a lane crosses hold 10^2 such hits.  This variant keeps the tile in LDS instead (8 KB per wave) and, per
// hit, lets 16 lanes walk the pixels of the hit's clipped box only: v = k * exp2(-(dx^2 + dy^2) c), one LDS float-max
// atomic (ds_max_f32) per box pixel, four hits in flight per wave.  Correct for any radius, but only faster below
// ~15x15 boxes; the host selects it on the caller's ACCV_HM_SMALL_RADII hint.  Same culling, same store path, same
// clear / in-place semantics.  SRC = 2 (splat_points_multi_kernel) reads float sample points and culls in two levels.
// NW waves share one tile (NW = 4 for the lane raster): a tile crossed by several lanes at a coarse scale has 6-8 sample
// groups to walk, a serial chain of ~2.5 us per group for ONE wave (24 us for the 576 tiles of a stride-16 map, as long
// as the 8704 tiles of the stride-4 map take) — with four waves the groups of a tile are dealt round-robin over the waves
// (the LDS float-max atomics commute, also across waves), and an empty tile is stored by four waves with two store
// instructions each instead of one wave with eight.
// ---- pieces of the per-tile body (small_body)
constexpr int kSmallTW = 128, kSmallTH = 16;
// LDS row stride of the tile: 128 + 4 floats.  The row walk below puts consecutive ROWS of a splat on consecutive lanes; with a
// stride of 128 floats they would all land on one LDS bank, with 132 they are 4 banks apart (16-byte row reads stay aligned)
constexpr int kSmallLdsW = kSmallTW + 4;
using SmallTile = float (*)[kSmallLdsW];

// the hits of a cull round (or of several, merged) are walked box by box.  Pixel updates are LDS float-max atomics
// (ds_max_f32, no return value): they commute, so neither overlapping boxes of concurrent hits nor successive hits need
// any ordering — the wave just streams them.
// rows_hint > 0: no clipped box of the list is taller than that (point splats: 2 r + 1) — then a lane takes ONE ROW of one hit
// and runs along its columns (radius 2: 12 hits per pass, 5 updates per lane) instead of 16 lanes sharing a hit's box in
// row-major order (4 hits per pass, 2 trips of ~15 dependent instructions for 25 pixels).  With one wave per tile the walk
// is part of the tile's serial chain, and the tiles of a coarse scale carry dozens of hits (round 3)
template <bool WG_SCOPE>
__device__ __forceinline__ void walk_hits(const SplatParams& p, const TileCtx& t, int lane, int nh, const Hit* __restrict__ hits,
                                          SmallTile tile, int rows_hint)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the hit list is complete
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (rows_hint > 0 && rows_hint <= 7) {   // (taller boxes: the 16-lane walk below needs fewer trips per hit — measured at r = 5)
        const int per_pass = 64 / rows_hint;
        const int hl = (int)((float)lane * (1.0f / (float)rows_hint) + 1e-3f);   // lane / rows_hint for lane < 64, rows <= 7
        const int rl = lane - hl * rows_hint;
        for (int h0 = 0; h0 < nh; h0 += per_pass) {
            const int h = h0 + hl;
            if (h >= nh || hl >= per_pass) continue;
            const Hit hh = hits[h];
            const int xlo = hh.box & 255u, xhi = (hh.box >> 8) & 255u, ylo = (hh.box >> 16) & 255u, yhi = hh.box >> 24;
            const int py = ylo + rl;
            if (py >= yhi) continue;
            const float dy = (float)(t.ty0 + py - hh.y);
            const float dy2 = dy * dy;
            float* row = &tile[py][0];
            for (int px = xlo; px < xhi; ++px) {
                const float dx = (float)(t.tx0 + px - hh.x);
                const float v = p.k * raw_exp2(-(dx * dx + dy2) * hh.c2);
                __hip_atomic_fetch_max(row + px, v, __ATOMIC_RELAXED,
                                       WG_SCOPE ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    } else {   // four hits at a time, 16 lanes per hit over its box in row-major order
        const int grp = lane >> 4, l16 = lane & 15;
        for (int h0 = 0; h0 < nh; h0 += 4) {
            const int h = h0 + grp;
            if (h >= nh) continue;
            const Hit hh = hits[h];
            const int xlo = hh.box & 255u, xhi = (hh.box >> 8) & 255u, ylo = (hh.box >> 16) & 255u, yhi = hh.box >> 24;
            const int w = xhi - xlo, area = w * (yhi - ylo);  // 0 for an empty box
            const float inv_w = 1.0f / (float)max(w, 1);
            for (int q = l16; q < area; q += 16) {
                // q / w for q < 2048, w <= 128: (q + 0.5) / w is at least 1/256 away from an integer, far more
                // than the error of the reciprocal
                const int py = (int)(((float)q + 0.5f) * inv_w);
                const int px = q - py * w;
                const float dx = (float)(t.tx0 + xlo + px - hh.x), dy = (float)(t.ty0 + ylo + py - hh.y);
                const float v = p.k * raw_exp2(-(dx * dx + dy * dy) * hh.c2);
                __hip_atomic_fetch_max(&tile[ylo + py][xlo + px], v, __ATOMIC_RELAXED,
                                       WG_SCOPE ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    }
    // the next round overwrites the hit list: order it behind this round's reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool WG_SCOPE>
__device__ __forceinline__ int splat_round(const SplatParams& p, const TileCtx& t, int lane, unsigned long long m, const Cand& cand,
                                           Hit* __restrict__ hits, SmallTile tile, int rows_hint = 0)
{
    const int nh = __popcll(m);
    int rows = 0;
    if ((m >> lane) & 1ull) {
        const Hit mine = make_hit(p, t, cand.x, cand.y, cand.r);
        hits[__popcll(m & ((1ull << lane) - 1ull))] = mine;
        rows = (int)(mine.box >> 24) - (int)((mine.box >> 16) & 255u);
    }
    if (rows_hint == 0) {   // objects of any radius (splat_small_kernel): the tallest clipped box of this round, wave-uniform
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) rows = max(rows, __shfl_xor(rows, d));
        rows_hint = max(1, __builtin_amdgcn_readfirstlane(rows));
    }
    walk_hits<WG_SCOPE>(p, t, lane, nh, hits, tile, rows_hint);
    return nh;
}

// sample-group bounding box (float, source pixels) of group g; valid = false for an empty group (a group of NaN points keeps
// xmin = +inf) or a group past the plane's sample count
struct GroupBox {
    float x0, y0, x1, y1;
    bool valid;
};
template <bool FIRST = false>   // FIRST: g = lane, the box plane_objects<true> requested (t.box0)
__device__ __forceinline__ GroupBox load_group_box(const SplatParams& p, const TileCtx& t, int g)
{
    // the load is UNCONDITIONAL (index clamped; callers run only with n_groups >= 1): its address does not depend on the
    // plane's sample count, so it is in flight together with the count's load instead of behind it — one dependent round
    // trip less in front of every tile's first store
    float4 v = FIRST ? t.box0 : t.boxes_f[min(g, p.n_groups - 1)];
    // (all four components at once: left alone, hipcc splits the load and sinks three of the pieces into the short-circuit
    // evaluation of group_reaches() — up to three dependent round trips to memory where one does)
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    return GroupBox{v.x, v.y, v.z, v.w, g < p.n_groups && g * kCand < t.n && v.x <= v.z};
}
// can a sample of the group reach pixel columns [cx0, cx1) x rows [cy0, cy1) of this scale?  CONSERVATIVE and division-free
// (the exact integer test runs per candidate afterwards): a sample at source x lands on pixel int(x / stride), which lies in
// (x / stride - 1, x / stride], and is drawn over [pixel - r, pixel + r]; the comparisons below are done in source pixels
// with one extra pixel of slack on either side for the rounding of the products.
struct ReachBounds {
    float xlo, xhi, ylo, yhi;   // the group can reach the region iff box.x1 >= xlo && box.x0 < xhi && (same in y)
};
__device__ __forceinline__ ReachBounds reach_bounds(const TileCtx& t, int rc, int cx0, int cx1, int cy0, int cy1)
{
    const float r1 = (float)min(rc, 1 << 24) + 2.0f;
    return ReachBounds{((float)cx0 - r1) * t.stride, ((float)cx1 + r1) * t.stride, ((float)cy0 - r1) * t.stride,
                       ((float)cy1 + r1) * t.stride};
}
__device__ __forceinline__ bool group_reaches(const GroupBox& b, const ReachBounds& rb)
{
    return b.valid && b.x1 >= rb.xlo && b.x0 < rb.xhi && b.y1 >= rb.ylo && b.y0 < rb.yhi;
}

// data-parallel-primitive moves (gfx9 DPP controls); lanes without a source lane keep `old`
constexpr int kDppQuad0 = 0x00, kDppQuad1 = 0x55, kDppQuad2 = 0xAA, kDppQuad3 = 0xFF;   // broadcast lane 0..3 of every quad
constexpr int kDppRowShr = 0x110;                                                      // + n: lane i <- lane i - n inside rows of 16
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;                              // lane i <- lane i + 1 / i - 1, whole wave
template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false);
}
// candidates of sample group `g` (one per lane), in two steps so that the requests of several groups are in flight together
// (fetch + use in one function made every group its own round trip to memory: the use waits for the data): the raw sample ...
__device__ __forceinline__ float2 request_group_samples(const TileCtx& t, int sub_base, int lane)
{
    return t.centers_f[min(sub_base + lane, t.n - 1)];  // n >= 1 inside the candidate loop
}
// ... and its target at this scale, exactly cull_load<2>; consecutive samples that land on the same pixel are one and the same
// splat (coarse scales see several samples per pixel): the first of a run is kept, results are unchanged
// x / stride, exactly: for a stride that is a power of two the product with its reciprocal is the same correctly rounded
// value as the IEEE division (both round x * 2^-k once) and costs one instruction instead of a dozen
struct PixelScale {
    float stride, inv;
    bool pow2;   // wave-uniform
};
__device__ __forceinline__ PixelScale pixel_scale(float stride)
{
    const bool pow2 = (__float_as_uint(stride) & 0x007fffffu) == 0u && stride > 1.0e-30f && stride < 1.0e30f;
    return PixelScale{stride, pow2 ? 1.0f / stride : 0.0f, pow2};
}
__device__ __forceinline__ float to_pixels(float x, const PixelScale& ps) { return ps.pow2 ? x * ps.inv : __fdiv_rn(x, ps.stride); }
__device__ __forceinline__ Cand group_candidates(const TileCtx& t, const PixelScale& ps, const float2 c, int lane)
{
    Cand out{(int)to_pixels(c.x, ps), (int)to_pixels(c.y, ps), t.radius, 0};
    if ((c.x != c.x) || (c.y != c.y)) out = Cand{0, 0, -1, 0};
    const int nx = dpp_i<kDppWaveShr1>(0, out.x), ny = dpp_i<kDppWaveShr1>(0, out.y), nr = dpp_i<kDppWaveShr1>(-2, out.r);
    if (lane > 0 && nx == out.x && ny == out.y && nr == out.r) out.r = -1;
    return out;
}

template <int SM>
__device__ __forceinline__ void store_segment(const SplatParams& p, float* plane_ptr, int row, int col0, const vfloat4& v)
{
    if constexpr (SM >= 2) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(plane_ptr, 0, (int)((size_t)p.H * p.W * 4), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)(((size_t)row * p.W + col0) * 4), 0, 18);
    } else {
        *reinterpret_cast<vfloat4*>(plane_ptr + (size_t)row * p.W + col0) = v;
    }
}

// rows [row0, row0 + RPW) of the finished LDS tile -> the map.  Fused clear: every segment is written (once).  In place: a
// segment still at its initial value (-inf) received nothing and is neither read nor written.
template <bool CLEAR, int SM, int RPW>
__device__ __forceinline__ void write_back_rows(const SplatParams& p, const TileCtx& t, float* plane_ptr, SmallTile tile, int row0,
                                                int lane, int col0)
{
    const float init = CLEAR ? 0.0f : -__builtin_inff();
    // the rows are read from LDS in batches of up to four (all reads of a batch in flight together — one row at a time was a chain
    // of RPW LDS round trips at the end of every touched tile), in place the old segments of a batch are requested together too
    constexpr int kBatch = RPW > 4 ? 4 : RPW;
    static_assert(RPW % kBatch == 0, "rows per half-wave come in whole batches");
#pragma unroll
    for (int i0 = 0; i0 < RPW; i0 += kBatch) {
        vfloat4 out[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) out[i] = *reinterpret_cast<const vfloat4*>(&tile[row0 + i0 + i][(lane & 31) * 4]);
        if constexpr (CLEAR) {
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int row = t.ty0 + row0 + i0 + i;
                if (row < p.H) store_segment<SM>(p, plane_ptr, row, col0, out[i]);
            }
        } else {
            bool dirty[kBatch];
            vfloat4 old[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int row = t.ty0 + row0 + i0 + i;
                dirty[i] = row < p.H && !(out[i].x == init && out[i].y == init && out[i].z == init && out[i].w == init);
                old[i] = vfloat4{0.0f, 0.0f, 0.0f, 0.0f};
                if (dirty[i]) old[i] = *reinterpret_cast<const vfloat4*>(plane_ptr + (size_t)row * p.W + col0);
            }
            const float nanv = __builtin_nanf("");
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                if (!dirty[i]) continue;
                vfloat4 o = out[i];
                o.x = max_skip_nan(old[i].x, o.x == init ? nanv : o.x);
                o.y = max_skip_nan(old[i].y, o.y == init ? nanv : o.y);
                o.z = max_skip_nan(old[i].z, o.z == init ? nanv : o.z);
                o.w = max_skip_nan(old[i].w, o.w == init ? nanv : o.w);
                store_segment<SM>(p, plane_ptr, t.ty0 + row0 + i0 + i, col0, o);
            }
        }
    }
}

template <bool CLEAR, int SM, int SRC, int NW = 1, int TH = kSmallTH>
__device__ __forceinline__ void small_body(const SplatParams& p, long long linear_group, Hit (*s_hit)[kCand], SmallTile s_tile)
{
    constexpr int TW = kSmallTW;
    constexpr int RPW = TH / NW / 2;  // rows per half-wave in the init / read-back passes
    static_assert(TH % (2 * NW) == 0, "rows must split evenly over the half-waves of the workgroup");

    const int lane = threadIdx.x & 63;
    const int wave = NW > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    TileCtx t;
    if (!locate_tile<TW, TH, 1, SRC == 2>(p, 0, t, linear_group)) return;  // uniform over the workgroup
    const int sub = lane >> 5, col0 = t.tx0 + (lane & 31) * 4;
    const int row0 = wave * (TH / NW) + sub * RPW;               // first of this half-wave's rows
    float* plane_ptr = p.hm + (size_t)t.plane * (size_t)p.H * (size_t)p.W;

    // "untouched" is -inf in the LDS tile (fused-clear mode starts from 0 = the cleared map).  Round 3: the tile is set up
    // LAZILY, at the first sample (group) that can reach it — a tile nothing reaches (most tiles of a lane raster) costs
    // no LDS traffic and no barrier: fused clear stores its zeros, in place does nothing.  The condition is uniform over
    // the workgroup.
    const float init = CLEAR ? 0.0f : -__builtin_inff();
    // ONE register quad for the initial value: prepare_tile is inlined into every unrolled cull round, and hipcc otherwise
    // materialises a fresh copy of the constant per store (8 rows x 4 rounds x 4 registers: 188 VGPRs, 2 waves per SIMD)
    vfloat4 vinit = vfloat4{init, init, init, init};
    asm volatile("" : "+v"(vinit));
    bool tile_ready = false;
    auto prepare_tile = [&]() {
        if (tile_ready) return;
        tile_ready = true;
#pragma unroll
        for (int i = 0; i < RPW; ++i) *reinterpret_cast<vfloat4*>(&s_tile[row0 + i][(lane & 31) * 4]) = vinit;
        if constexpr (NW > 1) {
            __syncthreads();  // tile initialised by all waves before the first atomic of any
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // tile initialised before the first atomic
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    // one cull round over the 64 candidates [sub_base, sub_base + 64)
    auto process_round = [&](int sub_base, const Cand& cand) {
        const unsigned long long m = cull_test(t, sub_base, lane, cand);
        if (m == 0) return;
        if constexpr (NW == 1) prepare_tile();   // (NW > 1: the caller prepared it — a barrier must not sit in per-wave flow)
        splat_round<(NW > 1)>(p, t, lane, m, cand, s_hit[wave], s_tile);
    };

    if constexpr (SRC == 2) {
        // two-level cull: consecutive polyline samples are neighbours in space, so each group of 64 carries a
        // bounding box (group_boxes_kernel); a lane tests one GROUP, and only groups that can reach the tile are
        // walked candidate by candidate — a tile crossed by a lane visits 1-3 rounds instead of all of them
        const int rc = min(max(t.radius, 0), 1 << 30);
        const int rows_hint = 2 * min(rc, 64) + 1;   // every sample has the same radius: no clipped box is taller
        const float c2_tile = hit_exponent_scale(p, t.radius);   // ... and the same exponent scale
        const PixelScale ps = pixel_scale(t.stride);
        const ReachBounds rb = reach_bounds(t, rc, t.tx0, t.tx1, t.ty0, t.ty1);
        for (int g0 = 0; g0 < p.n_groups; g0 += kCand) {
            unsigned long long mg = __ballot(group_reaches(g0 == 0 ? load_group_box<true>(p, t, lane) : load_group_box(p, t, g0 + lane), rb));
            if (mg == 0) continue;
            if constexpr (NW > 1) {  // every wave found the same groups; this one walks the (k * NW + wave)-th of them
                unsigned long long mine = 0;
                int k = 0;
                for (unsigned long long rest = mg; rest; rest &= rest - 1, ++k)
                    if (k % NW == wave) mine |= rest & (~rest + 1ull);
                mg = mine;
            }
            do {  // wave-uniform; the candidates of up to four groups are fetched together (one round trip)
                constexpr int kFetch = 4;
                int sub_base[kFetch];
                float2 raw[kFetch];
                Cand cand[kFetch];
#pragma unroll
                for (int u = 0; u < kFetch; ++u) {
                    sub_base[u] = -1;
                    raw[u] = float2{0.0f, 0.0f};
                    if (mg) {
                        sub_base[u] = (g0 + __builtin_ctzll(mg)) * kCand;
                        mg &= mg - 1;
                        raw[u] = request_group_samples(t, sub_base[u], lane);
                    }
                }
                // NW > 1: the tile is set up (LDS writes + a barrier, uniform over the workgroup: every wave saw the same
                // groups and comes through here even when none of them is its own) BEHIND the candidate requests, so that
                // the barrier overlaps their flight
                if constexpr (NW > 1) prepare_tile();
#pragma unroll
                for (int u = 0; u < kFetch; ++u)
                    if (sub_base[u] >= 0) cand[u] = group_candidates(t, ps, raw[u], lane);
                // Round 3: the hits of the (up to four) fetched groups go into ONE list and are walked together when they fit
                // it — a tile of a coarse scale is crossed by several lanes, each contributing a handful of samples per
                // group, and one compaction + one walk replaces four dependent ballot / LDS / fence / walk rounds
                unsigned long long mm[kFetch];
                int total = 0;
#pragma unroll
                for (int u = 0; u < kFetch; ++u) {
                    mm[u] = sub_base[u] >= 0 ? cull_test(t, sub_base[u], lane, cand[u]) : 0ull;
                    total += __popcll(mm[u]);
                }
                if (total == 0) continue;
                if (total <= kCand) {
                    if constexpr (NW == 1) prepare_tile();
                    int at = 0;
#pragma unroll
                    for (int u = 0; u < kFetch; ++u) {
                        if ((mm[u] >> lane) & 1ull)
                            s_hit[wave][at + __popcll(mm[u] & ((1ull << lane) - 1ull))] =
                                make_hit(p, t, cand[u].x, cand[u].y, cand[u].r, c2_tile);
                        at += __popcll(mm[u]);
                    }
                    walk_hits<(NW > 1)>(p, t, lane, total, s_hit[wave], s_tile, rows_hint);
                } else {
#pragma unroll
                    for (int u = 0; u < kFetch; ++u)
                        if (mm[u]) {
                            if constexpr (NW == 1) prepare_tile();
                            splat_round<(NW > 1)>(p, t, lane, mm[u], cand[u], s_hit[wave], s_tile, rows_hint);
                        }
                }
            } while (mg);
        }
    } else {
        static_assert(SRC == 2 || NW == 1, "the candidate-level cull prepares the tile per wave");
        // long object lists are the normal case here (10^3 lane samples per plane) and the wave needs few registers,
        // so the candidate loads of kFetch rounds are issued together: one memory round trip per kFetch * 64 candidates
        constexpr int kFetch = 4;
        for (int base = 0; base < t.n; base += kFetch * kCand) {
            Cand cand[kFetch];
#pragma unroll
            for (int u = 0; u < kFetch; ++u) cand[u] = cull_load<SRC>(t, base + u * kCand, lane);
#pragma unroll
            for (int u = 0; u < kFetch; ++u) {
                const int sub_base = base + u * kCand;
                if (sub_base >= t.n) break;
                process_round(sub_base, cand[u]);
            }
        }
    }

    if (!tile_ready) {   // nothing reached the tile: fused clear = zeros, in place = no HBM traffic at all
        if constexpr (CLEAR) {
            if (col0 < p.W) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const int row = t.ty0 + row0 + i;
                    if (row < p.H) store_segment<SM>(p, plane_ptr, row, col0, vfloat4{0.0f, 0.0f, 0.0f, 0.0f});
                }
            }
        }
        return;
    }
    if constexpr (NW > 1) {
        __syncthreads();  // all atomics of all waves landed before the tile is read back
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // all atomics landed before the tile is read back
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (col0 >= p.W) return;
    write_back_rows<CLEAR, SM, RPW>(p, t, plane_ptr, s_tile, row0, lane, col0);
}

template <bool CLEAR, int SM>
__global__ __launch_bounds__(64) void splat_small_kernel(const SplatParams p)
{
    __shared__ Hit s_hit[1][kCand];
    __shared__ __attribute__((aligned(16))) float s_tile[kSmallTH][kSmallLdsW];
    small_body<CLEAR, SM, 0>(p, blockIdx.x, s_hit, s_tile);   // (preload_params: +-1 % here — the one-level cull over 10^3 candidates
                                                              // is not bound by its prologue, profiles/r03_prologue_preload_elsewhere.log)
}

// lane rasters of all scales in one launch: float sample points, two-level cull (SRC = 2), scale from the tile prefix.
// NW = 4 (four waves share a tile) when coarse scales — many sample groups per tile — make up at least half of the tiles,
// else one wave per tile (decided on the host).  What bounds the launch is the chain of dependent round trips of each
// touched tile (kernel arguments -> count / group boxes -> candidates -> LDS -> store) times the tiles a CU holds at once,
// not an instruction count (PMC: a wave waits 70-77 % of its life).  Round 3 on config 3, in the order it was found
// (profiles/r03_lane_splat_*.log, every variant bit-identical): the tile set-up on first use, zeros ahead of the cull, a
// division-free group test, candidate requests ahead of the set-up barrier, count and box loads in parallel: +-2 % with four
// waves per tile (33.5 us); strips of 2 / 4 tiles per wave 43-77 us; a per-scale choice inside one 256-thread launch 36-37 us;
// coarse tiles dealt out every 2nd / 4th workgroup 36 / 43 us; 128 x 8 tiles 28-30 us.  What paid: ONE wave per tile for the
// whole launch (9 KB of LDS: 17 tiles per CU instead of 8; the coarse tiles' long chains run underneath the fine scale's
// stream) 28.5 us, the hits of up to four sample groups compacted and walked together 27.7 us, and a row of a splat per
// lane in that walk (12 hits per pass instead of 4 — with one wave per tile the walk IS part of the chain) 24.0 us.
template <bool CLEAR, int SM, int NW, int TH = kSmallTH>
__global__ __launch_bounds__(NW * 64) void splat_points_multi_kernel(const MultiParams mp)
{
    __shared__ Hit s_hit[NW][kCand];
    __shared__ __attribute__((aligned(16))) float s_tile[TH][kSmallLdsW];
    long long first;
    const int s = scale_of_group(mp, blockIdx.x, first);
    const SplatParams p = preload_params(mp.scale[s]);
    small_body<CLEAR, SM, 2, NW, TH>(p, (long long)blockIdx.x - first, s_hit, s_tile);
}

// ---------------------------------------------------------------- fused lane raster: polylines -> maps, ONE launch
// The two-launch lane raster above reads samples that polyline_kernel wrote in a launch of its own (5.9 us of configs[3]'s 30 us:
// a 4.4 us launch floor and a chain of its own, during which the chip idles).  Here the tile wave works from the POLYLINES:
//   level 1: every lane takes one segment of the frame's polylines (a frame's L x P2 point slots, P2 = points per polyline
//            rounded up to a power of two, at most 64 in all) and tests its bounding box against the tile — what the
//            group boxes of the sampler did, with no launch in front and nothing to read but the points themselves;
//   level 2: for the polylines with a segment in reach (most tiles: none) the wave repeats the sampler's arithmetic — segment
//            lengths, the prefix sum in the sampler's own order of additions (all polylines of a round of 64 slots at once,
//            each in its block of P2 lanes), and, for the stretch of samples that falls on the segments in reach, binary
//            search + interpolation — with the accumulated distances and the points in registers (DPP / ds_bpermute instead
//            of LDS arrays), converts the samples to this scale's pixels and splats them through the same hit list / row walk
//            / write-back as splat_points_multi_kernel.
// Bit-identical to sampler + point splat (tests/test_lane_raster_fused_gpu.py): the float steps are shared with
// polyline_kernel through polyline_arith.h, the scan follows polyline_kernel<f32, 256>'s order for polylines of at most 64
// points (one segment per thread, totals of four neighbours summed left to right, Hillis-Steele scan of the at most 16
// totals, exclusive offset = inclusive - own, offsets of the four neighbours added left to right), the samples sit at the
// fractions k / (S - 1) (IEEE division; the python layer builds the sampler's table the same way), and every sample that can
// land in the tile is evaluated: a sample lies on the segment its binary search finds, inside that segment's bounding box
// (weights in [0, 1]; the tile's reach is widened by a source pixel for the rounding of the products, and a segment with a
// coordinate of 2^20 or more, where that pixel would not cover it, counts as in reach), so it belongs to a segment in reach,
// and the stretch of sample numbers taken from the accumulated distances of the first and last such segment is widened by
// two samples either side (a polyline whose length is zero or not finite is evaluated whole).
// Where it pays (scripts/lane_fused_probe.py, profiles/r03_lane_raster_fused_*.log; configs[3]'s maps, 256 samples, radius 2):
// the sampling is repeated in every tile a polyline's segments reach, at every scale — VALU work of a wave that the two-launch
// path spends once, in the sampler — so the launch saved pays for sparse lane sets only: 1 / 2 polylines of 24 points per frame
// 22.6 / 24.2 -> 17.8 / 18.6 us (0.61 / 0.59 of the HBM peak), one of 64 points 23.1 -> 17.2 us; 4 x 16 points ties (24.5 vs
// 25.1 us) and configs[3]'s own 8 x 24 points — four rounds of 64 slots, most stride-4 tiles in reach of a polyline — took
// 40 us against 30 us (the code below keeps its loops over rounds; kLaneRounds = 4 is that variant, bit-identical as well).
// fused_lane_shape() therefore admits one round of slots and segments that carry at most half a pass of samples each.
constexpr int kLaneRounds = 1;   // rounds of 64 point slots per frame
struct LaneParams {
    const float2* points;       // [B, L, P] source pixels
    const void* point_counts;   // [B * L] valid leading points per polyline (i32 / i64), null = P
    int L, P, S;
    int p2_shift;               // point slots per polyline = 1 << p2_shift >= P
    int counts_i64;
};
struct FusedLaneParams {
    MultiParams mp;             // per scale: hm, H, W, stride, radius, factor, k, tiles; counts = valid polylines per frame, n_max = L
    LaneParams lp;
};

__device__ __forceinline__ float lane_read(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
template <typename T, int N>
__device__ __forceinline__ T pick(int r, const T (&a)[N])   // a[r] for a wave-uniform r without indexing registers
{
    T v = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) v = r == i ? a[i] : v;
    return v;
}

// ---- the f32 polyline sampler at wave level: the polylines sit one per block of P2 lanes (P2 a power of two, 4..64), point j of
// a block's polyline in the block's lane j.  Bit for bit polyline_kernel<f32, 256> for polylines of at most 64 points (see
// the notes above lane_body); every lane of the wave must be active in both functions (lane exchanges).
// accumulated distance of the lane's point (valid for j < n) and the polyline's length
__device__ __forceinline__ float polyline_accumulate(float px, float py, float next_x, float next_y, int n, int j, int lb, int P2,
                                                     float& total)
{
    float len = 0.0f;   // polyline_kernel: diff = point[s] - point[s + 1], acc = fma(diff, diff, acc) per coordinate, sqrt
    if (j + 1 < n) len = sqrtf(accv_poly::seg_length2_step(accv_poly::seg_length2_step(0.0f, px - next_x), py - next_y));
    const float v0 = dpp_f<kDppQuad0>(len, len), v1 = dpp_f<kDppQuad1>(len, len), v2 = dpp_f<kDppQuad2>(len, len),
                v3 = dpp_f<kDppQuad3>(len, len);
    const float chunk = ((v0 + v1) + v2) + v3;
    const float total4 = lane_read(chunk, lb + ((j << 2) & (P2 - 1)));   // (every lane takes part in the exchange)
    const float own = j < (P2 >> 2) ? total4 : 0.0f;
    float incl = own;   // lanes j < P2 / 4 <= 16 of the block hold its totals; they sit in one row of 16
    {
        float up = dpp_f<kDppRowShr + 1>(0.0f, incl);
        if (j >= 1) incl += up;
        up = dpp_f<kDppRowShr + 2>(0.0f, incl);
        if (j >= 2) incl += up;
        up = dpp_f<kDppRowShr + 4>(0.0f, incl);
        if (j >= 4) incl += up;
        up = dpp_f<kDppRowShr + 8>(0.0f, incl);
        if (j >= 8) incl += up;
    }
    const float excl = incl - own;
    float base = lane_read(excl, lb + (j >> 2));
    const int u4 = j & 3;
    if (u4 >= 1) base += v0;
    if (u4 >= 2) base += v1;
    if (u4 >= 3) base += v2;
    const float acc_next = base != 0.0f ? len + base : len;   // accum[j + 1]
    const float up1 = dpp_f<kDppWaveShr1>(0.0f, acc_next);
    const float acc = j == 0 ? 0.0f : up1;                    // accum[j]
    total = lane_read(acc, lb + max(n - 1, 0));
    return acc;
}
// the point at distance d along the lane's polyline (acc / px / py: what polyline_accumulate saw and returned; `act` = this lane
// wants a result — idle lanes still run along)
__device__ __forceinline__ void polyline_sample_at(float d, float acc, float px, float py, float total, int n, int lb, bool act,
                                                   float& sx, float& sy)
{
    // last point whose accumulated distance is <= d (polyline_kernel; all lanes stay in the loop for the exchanges)
    int mn = 0, mx = max(n - 1, 0);
    while (__ballot(act && mx - mn > 1)) {
        const int c = (mx + mn) >> 1;
        const float v = lane_read(acc, lb + c);
        if (mx - mn > 1) {
            if (v < d) mn = c;
            else if (v > d) mx = c;
            else mn = mx = c;
        }
    }
    int idx = mn;
    if (0.0f > d) idx = -1;
    else if (total < d) idx = n - 1;
    const int ia = min(max(idx, 0), max(n - 1, 0)), ic = min(ia + 1, max(n - 1, 0));
    const float d0 = lane_read(acc, lb + ia), d1 = lane_read(acc, lb + ic);
    const float pax = lane_read(px, lb + ia), pay = lane_read(py, lb + ia);
    const float pcx = lane_read(px, lb + ic), pcy = lane_read(py, lb + ic);
    sx = pax;   // before the first / beyond the last point, or a segment shorter than epsilon
    sy = pay;
    if (idx >= 0 && idx < n - 1) {
        const float seg_len = d1 - d0;
        if (seg_len >= 1.1920928955078125e-07f) {
            float w0, w1;
            accv_poly::lerp_weights(d, d0, d1, seg_len, w0, w1);
            sx = accv_poly::lerp_coord(pax, w0, pcx, w1);
            sy = accv_poly::lerp_coord(pay, w0, pcy, w1);
        }
    }
}

template <bool CLEAR, int SM>
__device__ __forceinline__ void lane_body(const SplatParams& p, const LaneParams& lp, long long linear_group, Hit* s_hit,
                                          SmallTile s_tile)
{
    constexpr int TW = kSmallTW, TH = kSmallTH, RPW = TH / 2;
    const int lane = threadIdx.x & 63;
    TileCtx t;
    if (!locate_tile<TW, TH, 1, true>(p, 0, t, linear_group)) return;   // t.n = valid polylines of this frame (<= L)
    const int sub = lane >> 5, col0 = t.tx0 + (lane & 31) * 4, row0 = sub * RPW;
    float* plane_ptr = p.hm + (size_t)t.plane * (size_t)p.H * (size_t)p.W;

    const float init = CLEAR ? 0.0f : -__builtin_inff();
    vfloat4 vinit = vfloat4{init, init, init, init};
    asm volatile("" : "+v"(vinit));
    bool tile_ready = false;
    auto prepare_tile = [&]() {   // on first use, as small_body
        if (tile_ready) return;
        tile_ready = true;
#pragma unroll
        for (int i = 0; i < RPW; ++i) *reinterpret_cast<vfloat4*>(&s_tile[row0 + i][(lane & 31) * 4]) = vinit;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    const int sh = lp.p2_shift, P2 = 1 << sh, per_round = 64 >> sh;
    const unsigned long long p2_mask = P2 == 64 ? ~0ull : ((1ull << P2) - 1ull);
    const int rc = min(max(t.radius, 0), 1 << 30);
    const int rows_hint = 2 * min(rc, 64) + 1;
    const float c2_tile = hit_exponent_scale(p, t.radius);
    const PixelScale ps = pixel_scale(t.stride);
    ReachBounds rb = reach_bounds(t, rc, t.tx0, t.tx1, t.ty0, t.ty1);
    rb.xlo -= 1.0f;   // the interpolation's products round: a sample can leave its segment's box by 4e-7 of the coordinates
    rb.ylo -= 1.0f;
    rb.xhi += 1.0f;
    rb.yhi += 1.0f;
    const float2* pts = lp.points + (size_t)t.plane * (size_t)lp.L * (size_t)lp.P;
    const int j = lane & (P2 - 1), lb = lane - j;

    // ---- level 1: the frame's point slots, all rounds requested together; one segment per lane
    float qx[kLaneRounds], qy[kLaneRounds];
    int cn[kLaneRounds];
#pragma unroll
    for (int r = 0; r < kLaneRounds; ++r) {
        qx[r] = qy[r] = 0.0f;
        cn[r] = 0;
        if (r * per_round < lp.L) {   // uniform
            const int l = (r * 64 + lane) >> sh, lc = min(l, lp.L - 1);
            const float2 v = pts[(size_t)lc * lp.P + min(j, lp.P - 1)];
            qx[r] = v.x;
            qy[r] = v.y;
            int c = lp.P;
            if (lp.point_counts) {
                const long long at = t.plane * lp.L + lc;
                const long long c64 = lp.counts_i64 ? static_cast<const long long*>(lp.point_counts)[at]
                                                    : (long long)static_cast<const int*>(lp.point_counts)[at];
                c = (int)max(0ll, min((long long)lp.P, c64));
            }
            cn[r] = l < t.n ? c : 0;   // a polyline past the frame's count has no points
        }
    }
    unsigned long long mseg[kLaneRounds];   // per round: the segments that can reach the tile (wave-uniform)
    float nx[kLaneRounds], ny[kLaneRounds];  // point j + 1 of the lane's polyline (point j itself for a single point)
    bool any_round = false;
#pragma unroll
    for (int r = 0; r < kLaneRounds; ++r) {
        mseg[r] = 0;
        nx[r] = ny[r] = 0.0f;
        if (r * per_round < lp.L) {
            const int n = cn[r];
            const bool has_next = j + 1 < n;
            nx[r] = dpp_f<kDppWaveShl1>(qx[r], qx[r]);   // (the last slot of a polyline never starts a segment)
            ny[r] = dpp_f<kDppWaveShl1>(qy[r], qy[r]);
            if (!has_next) {
                nx[r] = qx[r];
                ny[r] = qy[r];
            }
            const bool seg = has_next || (n == 1 && j == 0);
            const float ext = fmaxf(fmaxf(fabsf(qx[r]), fabsf(qy[r])), fmaxf(fabsf(nx[r]), fabsf(ny[r])));
            const bool reach = fmaxf(qx[r], nx[r]) >= rb.xlo && fminf(qx[r], nx[r]) < rb.xhi &&
                               fmaxf(qy[r], ny[r]) >= rb.ylo && fminf(qy[r], ny[r]) < rb.yhi;
            mseg[r] = __ballot(seg && (reach || !(ext < 1048576.0f)));
            any_round = any_round || mseg[r] != 0;
        }
    }

    int list_n = 0;   // hits waiting in s_hit
    if (any_round) {
        // ---- level 2a: accumulated distances of the polylines of every round with a segment in reach (the rounds' chains
        // of lane exchanges are independent and overlap), and each polyline's stretch of sample numbers
        float acc[kLaneRounds], tot[kLaneRounds];
        int klo[kLaneRounds], khi[kLaneRounds];
        const float s1 = (float)(lp.S - 1);
#pragma unroll
        for (int r = 0; r < kLaneRounds; ++r) {
            acc[r] = tot[r] = 0.0f;
            klo[r] = 1;
            khi[r] = 0;
            if (mseg[r] != 0) {   // uniform
                const int n = cn[r];
                acc[r] = polyline_accumulate(qx[r], qy[r], nx[r], ny[r], n, j, lb, P2, tot[r]);
                // the block's segments in reach -> its stretch of sample numbers
                const unsigned long long pm = (mseg[r] >> lb) & p2_mask;
                if (pm != 0) {
                    klo[r] = 0;
                    khi[r] = lp.S - 1;
                }
                const int jf = pm ? __builtin_ctzll(pm) : 0, jl = pm ? 63 - __builtin_clzll(pm) : 0;
                const float d_lo = lane_read(acc[r], lb + jf), d_hi = lane_read(acc[r], lb + min(jl + 1, max(n - 1, 0)));
                if (pm != 0 && tot[r] > 0.0f && tot[r] < __builtin_inff()) {
                    const float inv = __builtin_amdgcn_rcpf(tot[r]) * s1;   // (approximate: the stretch is widened by two samples)
                    klo[r] = max(0, (int)floorf(d_lo * inv) - 2);
                    khi[r] = min(lp.S - 1, (int)ceilf(d_hi * inv) + 2);
                }
            }
        }
        // ---- level 2b: the samples, a round of 64 slots at a time, P2 samples per polyline and pass
#pragma unroll 1
        for (int r = 0; r < kLaneRounds; ++r) {
            if (pick(r, mseg) == 0) continue;
            const float A = pick(r, acc), total = pick(r, tot), ax = pick(r, qx), ay = pick(r, qy);
            const int n = pick(r, cn), k_lo = pick(r, klo), k_hi = pick(r, khi);
            for (int k0 = k_lo + j;; k0 += P2) {
                const bool act = k0 <= k_hi;
                if (__ballot(act) == 0) break;
                const int k = min(k0, lp.S - 1);
                const float frac = lp.S > 1 ? __fdiv_rn((float)k, s1) : 0.0f;
                const float d = accv_poly::scale_query(frac, total);
                float sx, sy;
                polyline_sample_at(d, A, ax, ay, total, n, lb, act, sx, sy);
                // sample -> target of this scale (cull_load<2>), runs of samples on one pixel are one splat
                Cand c{(int)to_pixels(sx, ps), (int)to_pixels(sy, ps), t.radius, 0};
                if ((sx != sx) || (sy != sy) || !act) c = Cand{0, 0, -1, 0};
                const int ux = dpp_i<kDppWaveShr1>(0, c.x), uy = dpp_i<kDppWaveShr1>(0, c.y), ur = dpp_i<kDppWaveShr1>(-1, c.r);
                if (j > 0 && ux == c.x && uy == c.y && ur == c.r) c.r = -1;
                constexpr int kClampXY = 1 << 29, kClampR = 1 << 30;   // the conservative 32-bit test of cull_test
                const int xc = min(max(c.x, -kClampXY), kClampXY), yc = min(max(c.y, -kClampXY), kClampXY);
                const int rr = min(c.r, kClampR);
                const bool in = c.r >= 0 && xc - rr < t.tx1 && xc + rr >= t.tx0 && yc - rr < t.ty1 && yc + rr >= t.ty0;
                const unsigned long long m = __ballot(in);
                const int nh = __popcll(m);
                if (nh == 0) continue;
                prepare_tile();
                if (list_n + nh > kCand) {   // no room: walk what is waiting first
                    walk_hits<false>(p, t, lane, list_n, s_hit, s_tile, rows_hint);
                    list_n = 0;
                }
                if (in) s_hit[list_n + __popcll(m & ((1ull << lane) - 1ull))] = make_hit(p, t, c.x, c.y, c.r, c2_tile);
                list_n += nh;
            }
        }
        if (list_n > 0) walk_hits<false>(p, t, lane, list_n, s_hit, s_tile, rows_hint);
    }

    if (!tile_ready) {   // nothing reached the tile: fused clear = zeros, in place = no HBM traffic at all
        if constexpr (CLEAR) {
            if (col0 < p.W) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const int row = t.ty0 + row0 + i;
                    if (row < p.H) store_segment<SM>(p, plane_ptr, row, col0, vfloat4{0.0f, 0.0f, 0.0f, 0.0f});
                }
            }
        }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // all atomics landed before the tile is read back
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (col0 >= p.W) return;
    write_back_rows<CLEAR, SM, RPW>(p, t, plane_ptr, s_tile, row0, lane, col0);
}

template <bool CLEAR, int SM>
__global__ __launch_bounds__(64) void lane_raster_multi_kernel(const FusedLaneParams fp)
{
    __shared__ Hit s_hit[kCand];
    __shared__ __attribute__((aligned(16))) float s_tile[kSmallTH][kSmallLdsW];
    long long first;
    const int s = scale_of_group(fp.mp, blockIdx.x, first);
    const SplatParams p = preload_params(fp.mp.scale[s]);
    lane_body<CLEAR, SM>(p, fp.lp, (long long)blockIdx.x - first, s_hit, s_tile);
}

// ---------------------------------------------------------------- the polyline sampler as extra workgroups of the box-map launch
// configs[3] prepares box maps AND lane maps every step: box maps (one launch), sampler (one launch, 5.9 us of which 4.4 us are
// the launch itself), point splat (one launch).  The sampler's 256 workgroups are nothing next to the 11 456 tile waves of the box
// maps, and nothing in the box-map launch depends on them — so they ride in it: workgroups [0, n_polylines) of
// splat_multi_sampler_kernel sample one polyline each at wave level (polyline_accumulate / polyline_sample_at: bit for bit
// polyline_kernel<f32, 256> for polylines of at most 64 points; fractions k / (S - 1); S a multiple of 64) and write the samples
// and the bounding box of every 64 of them, exactly what accv_polyline_sample_boxes writes; the other workgroups are the tiles
// of splat_multi_kernel.  The point splat that follows on the stream finds both.
struct WaveSamplerParams {
    const float2* points;      // [n_polylines, P]
    const void* point_counts;  // [n_polylines] valid leading points (i32 / i64) or null = P
    float2* samples;           // [n_polylines, S]
    float4* boxes;             // [n_polylines * S / 64]
    int n_polylines, P, S, counts_i64;
};
struct TargetsParams {
    MultiParams mp;
    WaveSamplerParams sp;
};
__device__ __forceinline__ void wave_sampler(const WaveSamplerParams& sp, int b)
{
    const int lane = threadIdx.x & 63;
    int n = sp.P;
    if (sp.point_counts) {
        const long long c = sp.counts_i64 ? static_cast<const long long*>(sp.point_counts)[b]
                                          : (long long)static_cast<const int*>(sp.point_counts)[b];
        n = (int)max(0ll, min((long long)sp.P, c));
    }
    const float2 v = sp.points[(size_t)b * sp.P + min(lane, sp.P - 1)];
    const float px = v.x, py = v.y;
    float next_x = dpp_f<kDppWaveShl1>(px, px), next_y = dpp_f<kDppWaveShl1>(py, py);
    if (!(lane + 1 < n)) {
        next_x = px;
        next_y = py;
    }
    float total;
    const float acc = polyline_accumulate(px, py, next_x, next_y, n, lane, 0, 64, total);
    float2* out = sp.samples + (size_t)b * sp.S;
    const int groups = sp.S >> 6;
    const float s1 = (float)(sp.S - 1), inf = __builtin_inff();
    for (int k0 = 0; k0 < sp.S; k0 += 64) {
        const int k = k0 + lane;
        float sx = __builtin_nanf(""), sy = sx;   // a polyline without points: NaN everywhere (polyline_kernels.cuh:216-225)
        if (n > 0) {   // uniform
            const float d = accv_poly::scale_query(sp.S > 1 ? __fdiv_rn((float)k, s1) : 0.0f, total);
            polyline_sample_at(d, acc, px, py, total, n, 0, true, sx, sy);
        }
        out[k] = float2{sx, sy};
        const bool ok = sx == sx && sy == sy;
        float x0 = ok ? sx : inf, y0 = ok ? sy : inf, x1 = ok ? sx : -inf, y1 = ok ? sy : -inf;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, sft));
            y0 = fminf(y0, __shfl_xor(y0, sft));
            x1 = fmaxf(x1, __shfl_xor(x1, sft));
            y1 = fmaxf(y1, __shfl_xor(y1, sft));
        }
        if (lane == 0) sp.boxes[(size_t)b * groups + (k0 >> 6)] = make_float4(x0, y0, x1, y1);
    }
}
template <bool CLEAR, int SM>
__global__ __launch_bounds__(64) void splat_multi_sampler_kernel(const TargetsParams tp)
{
    if ((long long)blockIdx.x < (long long)tp.sp.n_polylines) {
        wave_sampler(tp.sp, (int)blockIdx.x);
        return;
    }
    const long long group = (long long)blockIdx.x - tp.sp.n_polylines;
    long long first;
    const int s = scale_of_group(tp.mp, group, first);
    splat_body<4, ACCV_BOX_TILE_R, CLEAR, SM, 1, 1>(tp.mp.scale[s], group - first);
}

// bounding box (xmin, ymin, xmax, ymax) of every 64 consecutive points of points[b, :, :] (NaN points ignored; a group
// without valid points keeps xmin = +inf > xmax = -inf): one wave per group
__global__ __launch_bounds__(64) void group_boxes_kernel(const float2* __restrict__ points, int num_points, int n_groups,
                                                         long long total_groups, float4* __restrict__ boxes)
{
    const long long gid = blockIdx.x;
    if (gid >= total_groups) return;
    const long long b = gid / n_groups;
    const int g = (int)(gid - b * n_groups);
    const int i = g * 64 + (int)threadIdx.x;
    const float inf = __builtin_inff();
    float x0 = inf, y0 = inf, x1 = -inf, y1 = -inf;
    if (i < num_points) {
        const float2 c = points[b * num_points + i];
        if (c.x == c.x && c.y == c.y) {
            x0 = x1 = c.x;
            y0 = y1 = c.y;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, d));
        y0 = fminf(y0, __shfl_xor(y0, d));
        x1 = fmaxf(x1, __shfl_xor(x1, d));
        y1 = fmaxf(y1, __shfl_xor(y1, d));
    }
    if (threadIdx.x == 0) boxes[gid] = make_float4(x0, y0, x1, y1);
}

void note_dispatch(const char* kernel, int px, int r, bool clear, int sm, const dim3& grid, const dim3& block)
{
    snprintf(accv::dispatch_buffer(), 256, "%s<PX=%d,R=%d,CLEAR=%d,SM=%d> grid(%u,%u,%u) block(%u)", kernel, px, r, clear ? 1 : 0,
             sm, grid.x, grid.y, grid.z, block.x);
}

// one-shot, per thread: events for the splat launch of the next flat / batched call (accv_draw_heatmap_time_next_launch)
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchEvents& launch_events()
{
    static thread_local LaunchEvents ev;
    return ev;
}
LaunchEvents take_launch_events()
{
    const LaunchEvents ev = launch_events();
    launch_events() = LaunchEvents{};
    return ev;
}
// same kernel, same launch parameters; with events the kernel's own start / stop time stamps are recorded (hip_ext.h)
template <class K>
inline void launch_maybe_timed(K kernel, const dim3& grid, const dim3& block, hipStream_t stream, const LaunchEvents& ev,
                               const SplatParams& p)
{
    if (ev.start || ev.stop)
        hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev.start, ev.stop, 0, p);
    else
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, p);
}

int launch_splat_small(SplatParams p, long long planes, bool clear, int sm, hipStream_t stream, const LaunchEvents& ev)
{
    p.tiles_x = (p.W + 127) / 128;
    p.tiles_y = (p.H + 15) / 16;
    p.n_tiles = planes * p.tiles_x * p.tiles_y;
    if (p.n_tiles == 0) return ACCV_OK;
    dim3 grid;
    p.grid3d = (planes <= 65535 && p.tiles_y <= 65535) ? 1 : 0;
    if (p.grid3d) {
        grid = dim3((unsigned)p.tiles_x, (unsigned)p.tiles_y, (unsigned)planes);
    } else {
        if (p.n_tiles > INT_MAX) return accv::fail(ACCV_EINVAL, "draw_heatmap: %lld tiles exceed the grid limit", p.n_tiles);
        grid = dim3((unsigned)p.n_tiles);
    }
    if (clear) {
        if (sm >= 2)
            launch_maybe_timed(splat_small_kernel<true, 4>, grid, dim3(64), stream, ev, p);
        else
            launch_maybe_timed(splat_small_kernel<true, 0>, grid, dim3(64), stream, ev, p);
    } else {
        if (sm >= 2)
            launch_maybe_timed(splat_small_kernel<false, 4>, grid, dim3(64), stream, ev, p);
        else
            launch_maybe_timed(splat_small_kernel<false, 0>, grid, dim3(64), stream, ev, p);
    }
    note_dispatch("splat_small_kernel", 4, 8, clear, sm >= 2 ? 4 : 0, grid, dim3(64));
    return accv::check_launch("draw_heatmap small-splat kernel");
}

// ---------------------------------------------------------------- target-prep front end
// centres/boxes (float, source-image pixels) -> integer centre + radius at an output stride; one fused kernel for
// the ~8 element-wise torch ops of the reference helper (packages/draw_heatmap/tests/_test_helpers.py:20-28):
//   r = max(1, int(ceil(min(cx-x0, cy-y0, x1-cx, y1-cy) / stride))),  c = int(c / stride)   (fp32, IEEE division)
__global__ __launch_bounds__(256) void targets_from_boxes_kernel(const float2* __restrict__ centers,
                                                                 const float4* __restrict__ boxes, long long n,
                                                                 float stride, int2* __restrict__ out_centers,
                                                                 int* __restrict__ out_radii)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 c = centers[i];
        const float4 b = boxes[i];
        const float m = fminf(fminf(c.x - b.x, c.y - b.y), fminf(b.z - c.x, b.w - c.y));
        int r = (int)ceilf(__fdiv_rn(m, stride));
        if (r < 1) r = 1;
        out_radii[i] = r;
        out_centers[i] = make_int2((int)__fdiv_rn(c.x, stride), (int)__fdiv_rn(c.y, stride));
    }
}

// sampled polyline points (float, source pixels) -> splat targets of a constant radius at an output stride:
//   c = int(p / stride) (same rule as above); a NaN point (sample of an empty polyline,
//   packages/lane_helpers/ext_impl/polyline/include/polyline_kernels.cuh:216-245) gets radius -1 = never drawn
__global__ __launch_bounds__(256) void targets_from_points_kernel(const float2* __restrict__ points, long long n,
                                                                  float stride, int radius,
                                                                  int2* __restrict__ out_centers,
                                                                  int* __restrict__ out_radii)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 c = points[i];
        const bool bad = (c.x != c.x) || (c.y != c.y);
        out_radii[i] = bad ? -1 : radius;
        out_centers[i] = bad ? make_int2(0, 0) : make_int2((int)__fdiv_rn(c.x, stride), (int)__fdiv_rn(c.y, stride));
    }
}

// ---------------------------------------------------------------- flat API: group objects by plane
__global__ void bin_count_kernel(const int32_t* __restrict__ idx, int n, int planes, int* __restrict__ cnt)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int pl = idx[i];
        if (pl >= 0 && pl < planes) atomicAdd(&cnt[pl], 1);
    }
}

// single workgroup: exclusive scan of cnt[0..planes) into off[0..planes], cnt reset to 0 (reused as cursor)
__global__ __launch_bounds__(1024) void bin_scan_kernel(int* __restrict__ cnt, int* __restrict__ off, int planes)
{
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int per = (planes + 1023) / 1024;
    const int lo = min(t * per, planes), hi = min(lo + per, planes);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    s_part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;  // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int c = cnt[i];
        off[i] = run;
        cnt[i] = 0;
        run += c;
    }
    if (t == 1023) off[planes] = s_part[1023];
}

// scatters every object into its plane's segment: the splat kernel then reads plane-sorted COPIES (centres, radii)
// with unit stride instead of chasing an index list
__global__ void bin_fill_kernel(const int32_t* __restrict__ idx, int n, int planes, const int* __restrict__ off,
                                int* __restrict__ cursor, const int2* __restrict__ centers,
                                const int32_t* __restrict__ radii, int2* __restrict__ sorted_centers,
                                int32_t* __restrict__ sorted_radii)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int pl = idx[i];
        if (pl >= 0 && pl < planes) {
            const int dst = off[pl] + atomicAdd(&cursor[pl], 1);
            sorted_centers[dst] = centers[i];
            sorted_radii[dst] = radii[i];
        }
    }
}

// count + scan + fill of the three kernels above in ONE single-workgroup launch, for the common small case (a few
// thousand objects, at most kBinSmallPlanes planes): per-plane counters / cursors live in LDS
constexpr int kBinSmallPlanes = 8192, kBinSmallObjects = 1 << 16;
__global__ __launch_bounds__(1024) void bin_small_kernel(const int32_t* __restrict__ idx, int n, int planes,
                                                         const int2* __restrict__ centers,
                                                         const int32_t* __restrict__ radii, int* __restrict__ off,
                                                         int2* __restrict__ sorted_centers,
                                                         int32_t* __restrict__ sorted_radii)
{
    __shared__ int s_cnt[kBinSmallPlanes];
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    for (int i = t; i < planes; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const int pl = idx[i];
        if (pl >= 0 && pl < planes) atomicAdd(&s_cnt[pl], 1);
    }
    __syncthreads();
    const int per = (planes + 1023) / 1024;
    const int lo = min(t * per, planes), hi = min(lo + per, planes);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += s_cnt[i];
    s_part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;  // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int c = s_cnt[i];
        off[i] = run;
        s_cnt[i] = run;  // becomes the plane's write cursor
        run += c;
    }
    if (t == 1023) off[planes] = s_part[1023];
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const int pl = idx[i];
        if (pl >= 0 && pl < planes) {
            const int dst = atomicAdd(&s_cnt[pl], 1);
            sorted_centers[dst] = centers[i];
            sorted_radii[dst] = radii[i];
        }
    }
}

// ONE 16-byte store per thread: waves that issue a single store stream at ~7 TB/s, a grid-stride loop (several stores
// per wave) at 4.3-5.9 TB/s on the same boxes (profiles/r01_fill_patterns*.log)
__global__ __launch_bounds__(256) void fill_kernel(float4* __restrict__ dst, size_t n4, float value)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = make_float4(value, value, value, value);
}
__global__ void fill_tail_kernel(float* __restrict__ dst, size_t n, float value)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

// store mode SM: 0 plain, 4 write-through non-temporal (sc1 nt).  (The A/B build also instantiates 1 = non-temporal and
// 2 = write-through, and 4-wave workgroups: -DACCV_TUNE_BUILD.)
template <int PX, int R, int WPG = kWavesPerGroup>
int launch_splat(SplatParams p, long long planes, bool clear, int sm, hipStream_t stream, const LaunchEvents& ev)
{
    p.tiles_x = (p.W + 32 * PX - 1) / (32 * PX);
    p.tiles_y = (p.H + 2 * R - 1) / (2 * R);
    p.n_tiles = planes * p.tiles_x * p.tiles_y;
    if (p.n_tiles == 0) return ACCV_OK;
    const int groups_x = (p.tiles_x + WPG - 1) / WPG;
    dim3 grid, block(WPG * 64);
    p.grid3d = (planes <= 65535 && p.tiles_y <= 65535) ? 1 : 0;
    if (p.grid3d) {
        grid = dim3((unsigned)groups_x, (unsigned)p.tiles_y, (unsigned)planes);
    } else {
        const long long groups = (p.n_tiles + WPG - 1) / WPG;
        if (groups > INT_MAX) return accv::fail(ACCV_EINVAL, "draw_heatmap: %lld tiles exceed the grid limit", p.n_tiles);
        grid = dim3((unsigned)groups);
    }
#define ACCV_LAUNCH_ONE(KERNEL) launch_maybe_timed(KERNEL, grid, block, stream, ev, p)
#define ACCV_LAUNCH_SM(SMV)                                                                              \
    do {                                                                                                 \
        if (clear)                                                                                       \
            ACCV_LAUNCH_ONE((splat_kernel<PX, R, true, SMV, WPG>));                                      \
        else                                                                                             \
            ACCV_LAUNCH_ONE((splat_kernel<PX, R, false, SMV, WPG>));                                     \
    } while (0)
    if constexpr (PX == 4) {
        switch (sm) {
#ifdef ACCV_TUNE_BUILD
            case 1: ACCV_LAUNCH_SM(1); break;
            case 2: ACCV_LAUNCH_SM(2); break;
#endif
            case 4: ACCV_LAUNCH_SM(4); break;
            case 5:  // density-adaptive (in-place launches only; a fused-clear launch asking for it gets plain stores)
                if (clear) {
                    sm = 0;
                    ACCV_LAUNCH_SM(0);
                } else {
                    ACCV_LAUNCH_ONE((splat_kernel<PX, R, false, 5, WPG>));
                }
                break;
            default: sm = 0; ACCV_LAUNCH_SM(0); break;
        }
    } else {
        sm = 0;
        ACCV_LAUNCH_SM(0);
    }
#undef ACCV_LAUNCH_SM
#undef ACCV_LAUNCH_ONE
    note_dispatch("splat_kernel", PX, R, clear, sm, grid, block);
    return accv::check_launch("draw_heatmap splat kernel");
}

// compute units of the current device (cached per device ordinal; 0 when the query fails)
long long compute_units()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
        cached[dev].store(n, std::memory_order_relaxed);
        v = n;
    }
    return v;
}

int dispatch_splat(SplatParams p, long long planes, bool clear, unsigned flags, hipStream_t stream, const LaunchEvents& ev)
{
    const bool small_hint = (flags & ACCV_HM_SMALL_RADII) != 0;
    const bool vec4 = (p.W % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.hm) & 15u) == 0);
    // Tile height: 128 x 16 pixel tiles (R = 8) for every launch.  Round 1 shipped 128 x 32 for fused-clear launches above
    // 128 MB after an A/B on nine boxes (profiles/r01_h1_ab_rows_store_policy.log: -4..-7 % there); with this round's
    // build the same A/B on four boxes of the slow class puts R = 8 ahead by 1.1-2.8 % on the headline batch and 6 % on
    // small-object batches (profiles/r02_h1_flags_ab_*.log), and in-place launches always preferred it.  R = 16 stays
    // available as a hint (ACCV_HM_TILE_ROWS_16).  Store policy: plain stores for fused-clear launches; in-place launches
    // decide per plane inside the kernel (SM = 5); ACCV_HM_WRITE_THROUGH / ACCV_HM_PLAIN_STORES override.  No knob table,
    // mutex or string look-up on this path.
    const bool plane_fits_rsrc = (size_t)p.H * p.W * sizeof(float) < ((size_t)1 << 31);
    int nt = accv::tune_get("hm_nt", -1);
    if (nt < 0) {
        if (flags & ACCV_HM_WRITE_THROUGH)
            nt = 4;
        else if (flags & ACCV_HM_PLAIN_STORES)
            nt = 0;
        else
            nt = clear ? 0 : 5;   // in-place: per-plane choice by object density (see the store loop of splat_body)
    }
    if (nt >= 2 && !plane_fits_rsrc) nt = 0;
    p.dense_area = 0.75f * (float)p.H * (float)p.W;
    int rows = accv::tune_get("hm_rows", -1);
    if (rows < 0) {
        if (flags & ACCV_HM_TILE_ROWS_16)
            rows = 16;
        else if (flags & ACCV_HM_TILE_ROWS_8)
            rows = 8;
        else
            rows = 8;
        // ... with one exception (round 3): a fused-clear launch whose 128 x 16 tiles are MORE than the chip holds at once (24
        // one-wave workgroups per CU) while its 128 x 32 tiles all fit (16 per CU) runs as a single round of resident tiles
        // instead of one full round plus a short second one — the 8-frame shards of the strong-scaling split (8160 tiles on
        // 256 CUs): 18.1 / 13.8 -> 17.3 / 13.2 us slowest / fastest shard (profiles/r03_small_launch_tile_rows.log); 16 frames
        // and more, which need several rounds either way, keep R = 8 (+0.5..1.5 % with 128 x 32 there)
        if (!(flags & (ACCV_HM_TILE_ROWS_8 | ACCV_HM_TILE_ROWS_16)) && clear && vec4 && !small_hint) {
            const long long cus = compute_units();
            const long long tx = (p.W + 127) / 128;
            const long long tiles8 = planes * tx * ((p.H + 15) / 16), tiles16 = planes * tx * ((p.H + 31) / 32);
            if (cus > 0 && tiles8 > 24 * cus && tiles16 <= 16 * cus) rows = 16;
        }
    }
    if (!p.labels) p.labels = p.radii;  // branch-free candidate loads: always a readable array (ignored when cls < 0)
    if (!vec4) return launch_splat<1, 8>(p, planes, clear, 0, stream, ev);
    const int small = accv::tune_get("hm_small", -1);   // point-like objects: the caller's ACCV_HM_SMALL_RADII hint
    // the small-splat kernel has no per-plane density choice: point-like objects are the sparse case, where write-through
    // costs up to 27 %, so the adaptive default (5) means PLAIN stores there; only an explicit
    // ACCV_HM_WRITE_THROUGH selects SM = 4
    if (small > 0 || (small < 0 && small_hint)) return launch_splat_small(p, planes, clear, nt == 5 ? 0 : nt, stream, ev);
    if (rows == 16) return launch_splat<4, 16>(p, planes, clear, nt, stream, ev);
#ifdef ACCV_TUNE_BUILD
    if (accv::tune_get("hm_wpg", kWavesPerGroup) == 4) return launch_splat<4, 8, 4>(p, planes, clear, nt, stream, ev);
#endif
    return launch_splat<4, 8>(p, planes, clear, nt, stream, ev);
}

int check_common(const void* hm, int h, int w, float factor, const char* who)
{
    if (h < 0 || w < 0) return accv::fail(ACCV_EINVAL, "%s: negative heatmap extent %dx%d", who, h, w);
    if (h > (1 << 29) || w > (1 << 29))
        return accv::fail(ACCV_EINVAL, "%s: heatmap extent %dx%d exceeds 2^29 per dimension", who, h, w);
    (void)hm;
    (void)factor;
    return ACCV_OK;
}

// Workgroups are dispatched in linear order: put the COARSE scales first.  Their tiles see the same objects / lane samples on
// fewer pixels, i.e. the longest per-tile chains, and started last they are the tail of the launch; the many short tiles of
// the fine scales fill in behind them (config 3: box maps 19.4 -> 17.4 us, in-place 24.9 -> 22.0, lane raster 40.8 -> 38.7;
// profiles/r02_scale_order_ab.log).
inline void coarse_scales_first(MultiParams& mp)
{
    std::stable_sort(mp.scale, mp.scale + mp.n_scales,
                     [](const SplatParams& a, const SplatParams& b) { return a.n_tiles < b.n_tiles; });
    long long tiles = 0;
    for (int i = 0; i < mp.n_scales; ++i) {
        mp.tile_begin[i] = tiles;
        tiles += mp.scale[i].n_tiles;
    }
    mp.tile_begin[mp.n_scales] = tiles;
}
// entries of the tile prefix past the last scale = the total, so that the kernels find a workgroup's scale with plain
// comparisons (scale_of_group)
inline void seal_tile_prefix(MultiParams& mp)
{
    for (int i = mp.n_scales + 1; i <= kMaxScales; ++i) mp.tile_begin[i] = mp.tile_begin[mp.n_scales];
}

}  // namespace

extern "C" {

int accv_draw_heatmap_time_next_launch(void* start_event, void* stop_event)
{
    launch_events() = LaunchEvents{static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)};
    return ACCV_OK;
}

size_t accv_draw_heatmap_flat_workspace_bytes(int num_planes, int num_objects)
{
    if (num_planes < 0 || num_objects < 0) return 0;
    // bin counters [P] | plane offsets [P+1] | plane-sorted centres int2[N] | plane-sorted radii [N]
    return accv::align_up((size_t)num_planes * 4, 16) + accv::align_up(((size_t)num_planes + 1) * 4, 16) +
           accv::align_up((size_t)num_objects * 8, 16) + accv::align_up((size_t)num_objects * 4, 16) + 16;
}

int accv_draw_heatmap_flat_f32(float* heatmaps, int num_planes, int height, int width, const int32_t* centers,
                               const int32_t* radii, const int32_t* heatmap_idxes, int num_objects,
                               float diameter_to_sigma_factor, float k_scale, unsigned flags, void* workspace,
                               size_t workspace_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const LaunchEvents ev = take_launch_events();   // consumed by this call whether or not it launches
    if (int rc = check_common(heatmaps, height, width, diameter_to_sigma_factor, "draw_heatmap")) return rc;
    if (num_planes < 0 || num_objects < 0) return accv::fail(ACCV_EINVAL, "draw_heatmap: negative count");
    if (num_objects > (1 << 30)) return accv::fail(ACCV_EINVAL, "draw_heatmap: more than 2^30 objects");
    const bool clear = (flags & ACCV_HM_CLEAR) != 0;
    if (num_planes == 0 || height == 0 || width == 0) return ACCV_OK;
    if (!heatmaps) return accv::fail(ACCV_EINVAL, "draw_heatmap: heatmap pointer is null");
    if (num_objects > 0 && (!centers || !radii || !heatmap_idxes))
        return accv::fail(ACCV_EINVAL, "draw_heatmap: null object array");
    if (num_objects == 0 && !clear) return ACCV_OK;
    const size_t need = accv_draw_heatmap_flat_workspace_bytes(num_planes, num_objects);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15u))
        return accv::fail(ACCV_EWORKSPACE, "draw_heatmap: workspace needs %zu aligned bytes, got %zu", need,
                          workspace_bytes);

    char* ws = static_cast<char*>(workspace);
    int* cnt = reinterpret_cast<int*>(ws);
    int* off = reinterpret_cast<int*>(ws + accv::align_up((size_t)num_planes * 4, 16));
    char* sc_raw = reinterpret_cast<char*>(off) + accv::align_up(((size_t)num_planes + 1) * 4, 16);
    int2* sorted_centers = reinterpret_cast<int2*>(sc_raw);
    int32_t* sorted_radii = reinterpret_cast<int32_t*>(sc_raw + accv::align_up((size_t)num_objects * 8, 16));

    if (num_planes <= kBinSmallPlanes && num_objects <= kBinSmallObjects) {
        hipLaunchKernelGGL(bin_small_kernel, dim3(1), dim3(1024), 0, stream, heatmap_idxes, num_objects, num_planes,
                           reinterpret_cast<const int2*>(centers), radii, off, sorted_centers, sorted_radii);
    } else {
        if (hipMemsetAsync(cnt, 0, (size_t)num_planes * 4, stream) != hipSuccess)
            return accv::fail(ACCV_ELAUNCH, "draw_heatmap: memset of the bin counters failed");
        const int nb = num_objects > 0 ? min((num_objects + 255) / 256, 1024) : 1;
        if (num_objects > 0)
            hipLaunchKernelGGL(bin_count_kernel, dim3(nb), dim3(256), 0, stream, heatmap_idxes, num_objects, num_planes, cnt);
        hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, stream, cnt, off, num_planes);
        if (num_objects > 0)
            hipLaunchKernelGGL(bin_fill_kernel, dim3(nb), dim3(256), 0, stream, heatmap_idxes, num_objects, num_planes, off,
                               cnt, reinterpret_cast<const int2*>(centers), radii, sorted_centers, sorted_radii);
    }
    if (int rc = accv::check_launch("draw_heatmap binning")) return rc;

    SplatParams p{};
    p.hm = heatmaps;
    p.centers = reinterpret_cast<const int32_t*>(sorted_centers);
    p.radii = sorted_radii;
    p.plane_off = off;
    p.H = height;
    p.W = width;
    p.factor = diameter_to_sigma_factor;
    p.k = k_scale;
    return dispatch_splat(p, num_planes, clear, flags, stream, ev);
}

int accv_draw_heatmap_batched_f32(float* heatmap, int batch, int num_classes, int height, int width,
                                  const int32_t* centers, const int32_t* radii, const void* counts,
                                  const int32_t* labels, int max_num_targets, float diameter_to_sigma_factor,
                                  float k_scale, unsigned flags, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const LaunchEvents ev = take_launch_events();   // consumed by this call whether or not it launches
    if (int rc = check_common(heatmap, height, width, diameter_to_sigma_factor, "draw_heatmap_batched")) return rc;
    if (batch < 0 || max_num_targets < 0 || num_classes < 0)
        return accv::fail(ACCV_EINVAL, "draw_heatmap_batched: negative count");
    if (max_num_targets > (1 << 30))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_batched: more than 2^30 objects per sample");
    if (batch == 0 || height == 0 || width == 0) return ACCV_OK;
    // (an empty [B, 0] or [0, N] label tensor has no storage: without object slots the pointer is not looked at)
    if (max_num_targets > 0 && (num_classes > 0) != (labels != nullptr))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_batched: labels and num_classes must be given together");
    if (!heatmap) return accv::fail(ACCV_EINVAL, "draw_heatmap_batched: heatmap pointer is null");
    if (!counts) return accv::fail(ACCV_EINVAL, "draw_heatmap_batched: counts pointer is null");
    if (max_num_targets > 0 && (!centers || !radii))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_batched: null object array");
    const bool clear = (flags & ACCV_HM_CLEAR) != 0;
    if (max_num_targets == 0 && !clear) return ACCV_OK;

    SplatParams p{};
    p.hm = heatmap;
    p.centers = centers;
    p.radii = radii;
    p.labels = labels;
    p.counts = counts;
    p.H = height;
    p.W = width;
    p.n_max = max_num_targets;
    p.n_classes = num_classes;
    p.factor = diameter_to_sigma_factor;
    p.k = k_scale;
    p.counts_i64 = (flags & ACCV_HM_COUNTS_I64) ? 1 : 0;
    const long long planes = (long long)batch * (num_classes > 0 ? num_classes : 1);
    return dispatch_splat(p, planes, clear, flags, stream, ev);
}

}  // extern "C"
namespace {
// box maps of all scales in one launch; with `sampler` the polyline sampler rides in the same launch (splat_multi_sampler_kernel)
int draw_multiscale_impl(float* const* heatmaps, const int* heights, const int* widths, const float* strides, int num_scales,
                         int batch, const float* centers_xy, const float* boxes_xyxy, const void* counts, int max_num_targets,
                         float diameter_to_sigma_factor, float k_scale, unsigned flags, const WaveSamplerParams* sampler,
                         hipStream_t stream);
}
extern "C" {

int accv_draw_heatmap_multiscale_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                     int num_scales, int batch, const float* centers_xy, const float* boxes_xyxy,
                                     const void* counts, int max_num_targets, float diameter_to_sigma_factor,
                                     float k_scale, unsigned flags, void* stream_)
{
    return draw_multiscale_impl(heatmaps, heights, widths, strides, num_scales, batch, centers_xy, boxes_xyxy, counts,
                                max_num_targets, diameter_to_sigma_factor, k_scale, flags, nullptr, static_cast<hipStream_t>(stream_));
}

int accv_draw_heatmap_multiscale_sample_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                            int num_scales, int batch, const float* centers_xy, const float* boxes_xyxy,
                                            const void* counts, int max_num_targets, float diameter_to_sigma_factor,
                                            float k_scale, unsigned flags, const float* polylines_xy, int num_polylines,
                                            int points, const void* point_counts, int num_samples, float* samples,
                                            float* group_boxes, void* stream_)
{
    if (num_polylines < 0) return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale_sample: negative polyline count");
    if (points < 1 || points > 64)
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale_sample: 1..64 points per polyline (wave-level sampler), got %d",
                          points);
    if (num_samples < 64 || num_samples % 64 != 0 || num_samples > (1 << 20))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale_sample: the number of samples must be a multiple of 64 up to 2^20 "
                                       "(one group box per 64 samples), got %d", num_samples);
    if (num_polylines > 0 && (!polylines_xy || !samples || !group_boxes))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale_sample: null polylines / samples / group_boxes");
    if ((reinterpret_cast<uintptr_t>(polylines_xy) & 7u) || (reinterpret_cast<uintptr_t>(samples) & 7u) ||
        (reinterpret_cast<uintptr_t>(group_boxes) & 15u))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale_sample: polylines / samples need 8-byte, group boxes 16-byte alignment");
    WaveSamplerParams sp{};
    sp.points = reinterpret_cast<const float2*>(polylines_xy);
    sp.point_counts = point_counts;
    sp.samples = reinterpret_cast<float2*>(samples);
    sp.boxes = reinterpret_cast<float4*>(group_boxes);
    sp.n_polylines = num_polylines;
    sp.P = points;
    sp.S = num_samples;
    sp.counts_i64 = (flags & ACCV_HM_POINT_COUNTS_I64) ? 1 : 0;
    return draw_multiscale_impl(heatmaps, heights, widths, strides, num_scales, batch, centers_xy, boxes_xyxy, counts,
                                max_num_targets, diameter_to_sigma_factor, k_scale, flags, num_polylines > 0 ? &sp : nullptr,
                                static_cast<hipStream_t>(stream_));
}

}  // extern "C"
namespace {
int draw_multiscale_impl(float* const* heatmaps, const int* heights, const int* widths, const float* strides, int num_scales,
                         int batch, const float* centers_xy, const float* boxes_xyxy, const void* counts, int max_num_targets,
                         float diameter_to_sigma_factor, float k_scale, unsigned flags, const WaveSamplerParams* sampler,
                         hipStream_t stream)
{
    (void)take_launch_events();   // a pending accv_draw_heatmap_time_next_launch pair is dropped, not kept for a later call
    // the sampler's workgroups alone: whenever the box maps have nothing to launch
    auto sampler_only = [&]() -> int {
        if (!sampler) return ACCV_OK;
        TargetsParams tp{};
        tp.sp = *sampler;
        seal_tile_prefix(tp.mp);
        hipLaunchKernelGGL((splat_multi_sampler_kernel<true, 0>), dim3((unsigned)sampler->n_polylines), dim3(64), 0, stream, tp);
        note_dispatch("splat_multi_sampler_kernel", 4, 8, true, 0, dim3((unsigned)sampler->n_polylines), dim3(64));
        return accv::check_launch("draw_heatmap multi-scale splat + sampler kernel");
    };
    if (num_scales < 1 || num_scales > kMaxScales)
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: 1..%d scales supported, got %d", kMaxScales, num_scales);
    if (!heatmaps || !heights || !widths || !strides) return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: null array");
    if (batch < 0 || max_num_targets < 0) return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: negative count");
    if (max_num_targets > (1 << 30))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: more than 2^30 objects per sample");
    if (batch == 0) return sampler_only();
    if (!counts) return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: counts pointer is null");
    if (max_num_targets > 0 && (!centers_xy || !boxes_xyxy))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: null object array");
    if ((reinterpret_cast<uintptr_t>(boxes_xyxy) & 15u) || (reinterpret_cast<uintptr_t>(centers_xy) & 7u))
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: centres need 8-byte and boxes 16-byte alignment");
    const bool clear = (flags & ACCV_HM_CLEAR) != 0;
    if (max_num_targets == 0 && !clear) return sampler_only();

    MultiParams mp{};
    long long tiles = 0;
    int used = 0;
    for (int i = 0; i < num_scales; ++i) {
        if (int rc = check_common(heatmaps[i], heights[i], widths[i], diameter_to_sigma_factor, "draw_heatmap_multiscale"))
            return rc;
        if (!(strides[i] > 0.0f)) return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: stride %d is not positive", i);
        if (heights[i] == 0 || widths[i] == 0) continue;
        if (!heatmaps[i]) return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: heatmap %d is null", i);
        if (widths[i] % 4 != 0 || (reinterpret_cast<uintptr_t>(heatmaps[i]) & 15u) ||
            (size_t)heights[i] * widths[i] * sizeof(float) >= ((size_t)1 << 31))
            return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: map %d needs a width that is a multiple of 4, a 16-byte "
                                           "aligned base and planes below 2 GiB (use the per-scale calls otherwise)", i);
        SplatParams& p = mp.scale[used];
        p.hm = heatmaps[i];
        p.counts = counts;
        p.H = heights[i];
        p.W = widths[i];
        p.n_max = max_num_targets;
        p.factor = diameter_to_sigma_factor;
        p.k = k_scale;
        p.counts_i64 = (flags & ACCV_HM_COUNTS_I64) ? 1 : 0;
        p.centers_f = centers_xy;
        p.boxes_f = boxes_xyxy;
        p.stride = strides[i];
        p.tiles_x = (p.W + 127) / 128;
        p.tiles_y = (p.H + 2 * ACCV_BOX_TILE_R - 1) / (2 * ACCV_BOX_TILE_R);
        p.n_tiles = (long long)batch * p.tiles_x * p.tiles_y;
        p.grid3d = 0;
        mp.tile_begin[used] = tiles;
        tiles += p.n_tiles;
        ++used;
    }
    mp.n_scales = used;
    mp.tile_begin[used] = tiles;
    if (used == 0 || tiles == 0) return sampler_only();
    if (tiles + (sampler ? sampler->n_polylines : 0) > INT_MAX)
        return accv::fail(ACCV_EINVAL, "draw_heatmap_multiscale: %lld tiles exceed the grid limit", tiles);
    if (!(flags & ACCV_HM_CALLER_SCALE_ORDER)) coarse_scales_first(mp);
    seal_tile_prefix(mp);
    int nt = accv::tune_get("hm_nt", -1);
    if (nt < 0) nt = (flags & ACCV_HM_WRITE_THROUGH) ? 4 : 0;   // same store policy as the single-scale path
    if (sampler) {
        TargetsParams tp{};
        tp.mp = mp;
        tp.sp = *sampler;
        const dim3 grid((unsigned)(tiles + sampler->n_polylines)), block(64);
        if (clear) {
            if (nt >= 2)
                hipLaunchKernelGGL((splat_multi_sampler_kernel<true, 4>), grid, block, 0, stream, tp);
            else
                hipLaunchKernelGGL((splat_multi_sampler_kernel<true, 0>), grid, block, 0, stream, tp);
        } else {
            if (nt >= 2)
                hipLaunchKernelGGL((splat_multi_sampler_kernel<false, 4>), grid, block, 0, stream, tp);
            else
                hipLaunchKernelGGL((splat_multi_sampler_kernel<false, 0>), grid, block, 0, stream, tp);
        }
        note_dispatch("splat_multi_sampler_kernel", 4, 8, clear, nt >= 2 ? 4 : 0, grid, block);
        return accv::check_launch("draw_heatmap multi-scale splat + sampler kernel");
    }
    const dim3 grid((unsigned)tiles), block(64);
    if (clear) {
        if (nt >= 2)
            hipLaunchKernelGGL((splat_multi_kernel<true, 4>), grid, block, 0, stream, mp);
        else
            hipLaunchKernelGGL((splat_multi_kernel<true, 0>), grid, block, 0, stream, mp);
    } else {
        if (nt >= 2)
            hipLaunchKernelGGL((splat_multi_kernel<false, 4>), grid, block, 0, stream, mp);
        else
            hipLaunchKernelGGL((splat_multi_kernel<false, 0>), grid, block, 0, stream, mp);
    }
    note_dispatch("splat_multi_kernel", 4, 8, clear, nt >= 2 ? 4 : 0, grid, block);
    return accv::check_launch("draw_heatmap multi-scale splat kernel");
}
}  // namespace
extern "C" {


size_t accv_draw_points_workspace_bytes(int batch, int num_points)
{
    if (batch < 0 || num_points < 0) return 0;
    return (size_t)batch * (size_t)((num_points + 63) / 64) * sizeof(float4) + 16;
}

int accv_draw_points_multiscale_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                    int num_scales, int batch, const float* points_xy, const void* counts, int num_points,
                                    int radius, float diameter_to_sigma_factor, float k_scale, unsigned flags,
                                    void* workspace, size_t workspace_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    (void)take_launch_events();   // a pending accv_draw_heatmap_time_next_launch pair is dropped, not kept for a later call
    if (num_scales < 1 || num_scales > kMaxScales)
        return accv::fail(ACCV_EINVAL, "draw_points_multiscale: 1..%d scales supported, got %d", kMaxScales, num_scales);
    if (!heatmaps || !heights || !widths || !strides) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: null array");
    if (batch < 0 || num_points < 0) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: negative count");
    if (num_points > (1 << 30)) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: more than 2^30 points per sample");
    if (batch == 0) return ACCV_OK;
    if (!counts) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: counts pointer is null");
    if (num_points > 0 && !points_xy) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: null point array");
    if (reinterpret_cast<uintptr_t>(points_xy) & 7u)
        return accv::fail(ACCV_EINVAL, "draw_points_multiscale: points need 8-byte alignment");
    const bool clear = (flags & ACCV_HM_CLEAR) != 0;
    if (num_points == 0 && !clear) return ACCV_OK;
    const size_t need = accv_draw_points_workspace_bytes(batch, num_points);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15u))
        return accv::fail(ACCV_EWORKSPACE, "draw_points_multiscale: workspace needs %zu aligned bytes, got %zu", need,
                          workspace_bytes);
    const int n_groups = (num_points + 63) / 64;

    MultiParams mp{};
    long long tiles = 0;
    int used = 0;
    for (int i = 0; i < num_scales; ++i) {
        if (int rc = check_common(heatmaps[i], heights[i], widths[i], diameter_to_sigma_factor, "draw_points_multiscale"))
            return rc;
        if (!(strides[i] > 0.0f)) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: stride %d is not positive", i);
        if (heights[i] == 0 || widths[i] == 0) continue;
        if (!heatmaps[i]) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: heatmap %d is null", i);
        if (widths[i] % 4 != 0 || (reinterpret_cast<uintptr_t>(heatmaps[i]) & 15u) ||
            (size_t)heights[i] * widths[i] * sizeof(float) >= ((size_t)1 << 31))
            return accv::fail(ACCV_EINVAL, "draw_points_multiscale: map %d needs a width that is a multiple of 4, a 16-byte "
                                           "aligned base and planes below 2 GiB (use the per-scale calls otherwise)", i);
        SplatParams& p = mp.scale[used];
        p.hm = heatmaps[i];
        p.counts = counts;
        p.H = heights[i];
        p.W = widths[i];
        p.n_max = num_points;
        p.factor = diameter_to_sigma_factor;
        p.k = k_scale;
        p.counts_i64 = (flags & ACCV_HM_COUNTS_I64) ? 1 : 0;
        p.centers_f = points_xy;
        p.boxes_f = static_cast<const float*>(workspace);
        p.stride = strides[i];
        p.radius = radius;
        p.n_groups = n_groups;
        p.tiles_x = (p.W + 127) / 128;
        p.tiles_y = (p.H + 15) / 16;
        p.n_tiles = (long long)batch * p.tiles_x * p.tiles_y;
        p.grid3d = 0;
        mp.tile_begin[used] = tiles;
        tiles += p.n_tiles;
        ++used;
    }
    mp.n_scales = used;
    mp.tile_begin[used] = tiles;
    if (used == 0 || tiles == 0) return ACCV_OK;
    if (tiles > INT_MAX) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: %lld tiles exceed the grid limit", tiles);
    if (!(flags & ACCV_HM_CALLER_SCALE_ORDER)) coarse_scales_first(mp);
    seal_tile_prefix(mp);
    const long long total_groups = (long long)batch * n_groups;
    if (total_groups > INT_MAX) return accv::fail(ACCV_EINVAL, "draw_points_multiscale: too many point groups");
    if (total_groups > 0 && !(flags & ACCV_HM_GROUP_BOXES_GIVEN))
        hipLaunchKernelGGL(group_boxes_kernel, dim3((unsigned)total_groups), dim3(64), 0, stream,
                           reinterpret_cast<const float2*>(points_xy), num_points, n_groups, total_groups,
                           static_cast<float4*>(workspace));
    int nt = accv::tune_get("hm_nt", -1);
    if (nt < 0) nt = (flags & ACCV_HM_WRITE_THROUGH) ? 4 : 0;
    // a tile of a coarse scale is crossed by several lanes and has many sample groups to walk: share it among four waves
    // when some scale averages >= 24 samples per tile (config 3: 7.5 / 30 / 113 at strides 4 / 8 / 16); fine scales alone
    // keep one wave per tile (mostly empty tiles would only pay the barriers: 21.7 -> 23.8 us at stride 4)
    // Round 3: ... and only when those coarse tiles are at least half of the launch.  Next to a fine scale with several times
    // their tile count the coarse tiles' long chains run underneath the fine scale's stream either way, and one wave per tile
    // (9 KB of LDS, 17-20 tiles resident per CU instead of 8) is what the fine scale wants: config 3 (strides 4 / 8 / 16)
    // 32.9 -> 28.7 us, its empty-tile floor 21 -> 15 us (profiles/r03_lane_splat_modes_sweep4_tile_height.log)
    long long coarse_tiles = 0;
    for (int i = 0; i < used; ++i)
        if ((double)batch * num_points >= 24.0 * (double)mp.scale[i].n_tiles) coarse_tiles += mp.scale[i].n_tiles;
    bool heavy = 2 * coarse_tiles >= tiles;
    if (const int nw = accv::tune_get("pts_nw", -1); nw > 0) heavy = nw == 4;   // A/B build only
#ifdef ACCV_TUNE_BUILD
    if (accv::tune_get("pts_th", 16) == 8) {   // experiment: 128 x 8 tiles, one wave each (5 KB of LDS: 32 tiles per CU)
        long long t8 = 0;
        for (int i = 0; i < used; ++i) {
            SplatParams& p = mp.scale[i];
            p.tiles_y = (p.H + 7) / 8;
            p.n_tiles = (long long)batch * p.tiles_x * p.tiles_y;
            mp.tile_begin[i] = t8;
            t8 += p.n_tiles;
        }
        mp.tile_begin[used] = t8;
        seal_tile_prefix(mp);
        const dim3 grid8((unsigned)t8);
        if (clear)
            hipLaunchKernelGGL((splat_points_multi_kernel<true, 0, 1, 8>), grid8, dim3(64), 0, stream, mp);
        else
            hipLaunchKernelGGL((splat_points_multi_kernel<false, 0, 1, 8>), grid8, dim3(64), 0, stream, mp);
        note_dispatch("splat_points_multi_kernel(TH=8)", 4, 4, clear, 0, grid8, dim3(64));
        return accv::check_launch("draw_heatmap multi-scale point splat kernel");
    }
#endif
    const dim3 grid((unsigned)tiles), block(heavy ? 256 : 64);
#define ACCV_LAUNCH_POINTS(CL, SMV)                                                                       \
    do {                                                                                                  \
        if (heavy)                                                                                        \
            hipLaunchKernelGGL((splat_points_multi_kernel<CL, SMV, 4>), grid, block, 0, stream, mp);      \
        else                                                                                              \
            hipLaunchKernelGGL((splat_points_multi_kernel<CL, SMV, 1>), grid, block, 0, stream, mp);      \
    } while (0)
    if (clear) {
        if (nt >= 2)
            ACCV_LAUNCH_POINTS(true, 4);
        else
            ACCV_LAUNCH_POINTS(true, 0);
    } else {
        if (nt >= 2)
            ACCV_LAUNCH_POINTS(false, 4);
        else
            ACCV_LAUNCH_POINTS(false, 0);
    }
#undef ACCV_LAUNCH_POINTS
    note_dispatch("splat_points_multi_kernel", 4, 8, clear, nt >= 2 ? 4 : 0, grid, block);
    return accv::check_launch("draw_heatmap multi-scale point splat kernel");
}

// shape rule of the fused lane raster (one place: the entry point below and accv_draw_polylines_fused_applicable)
namespace {
struct FusedLaneShape {
    bool ok;
    int p2_shift;
};
FusedLaneShape fused_lane_shape(const int* heights, const int* widths, int num_scales, int batch, int lanes, int points,
                                int num_samples)
{
    FusedLaneShape out{false, 0};
    if (lanes < 1 || points < 1 || points > 64 || num_samples < 1 || num_samples > (1 << 20) || batch < 1) return out;
    int sh = 2;   // at least four slots per polyline: the polylines in reach are a 64-bit mask
    while ((1 << sh) < points) ++sh;
    if (((long long)lanes << sh) > kLaneRounds * 64) return out;
    // a pass evaluates 2^sh samples per polyline: more than a pass or two per segment in reach (few, long segments carrying many
    // samples: 8 polylines x 8 points x 256 samples 42 us against 30 us) and the sampler launch is the cheaper way
    if ((long long)num_samples * 2 > (long long)std::max(points - 1, 1) << sh) return out;
    // one wave per tile only: launches that the point splat would run with four waves per tile (coarse scales are at least
    // half of the tiles) stay on the two-launch path
    long long tiles = 0, coarse = 0;
    for (int i = 0; i < num_scales; ++i) {
        if (heights[i] <= 0 || widths[i] <= 0) continue;
        const long long nt = (long long)batch * ((widths[i] + 127) / 128) * ((heights[i] + 15) / 16);
        tiles += nt;
        if ((double)batch * lanes * num_samples >= 24.0 * (double)nt) coarse += nt;
    }
    if (tiles == 0 || 2 * coarse >= tiles) return out;
    out.ok = true;
    out.p2_shift = sh;
    return out;
}
}  // namespace

int accv_draw_polylines_fused_applicable(const int* heights, const int* widths, int num_scales, int batch, int lanes,
                                         int points, int num_samples)
{
    if (!heights || !widths || num_scales < 1 || num_scales > kMaxScales) return 0;
    if (accv::tune_get("lane_fused", 1) == 0) return 0;   // A/B build only
    return fused_lane_shape(heights, widths, num_scales, batch, lanes, points, num_samples).ok ? 1 : 0;
}

int accv_draw_polylines_multiscale_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                       int num_scales, int batch, const float* polylines_xy, int lanes, int points,
                                       const void* point_counts, const void* lane_counts, int num_samples, int radius,
                                       float diameter_to_sigma_factor, float k_scale, unsigned flags, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (num_scales < 1 || num_scales > kMaxScales)
        return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: 1..%d scales supported, got %d", kMaxScales, num_scales);
    if (!heatmaps || !heights || !widths || !strides) return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: null array");
    if (batch < 0) return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: negative batch");
    if (batch == 0) return ACCV_OK;
    if (radius < 0) return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: negative radius");
    const FusedLaneShape shape = fused_lane_shape(heights, widths, num_scales, batch, lanes, points, num_samples);
    if (!shape.ok)
        return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: shape outside the fused kernel (1..64 points per polyline, "
                                       "lanes x points rounded up to a power of two <= %d, samples <= (points - 1) x that power "
                                       "of two / 2, fine scales in the majority: accv_draw_polylines_fused_applicable); use "
                                       "accv_polyline_sample_boxes + accv_draw_points_multiscale_f32", kLaneRounds * 64);
    if (!polylines_xy || !lane_counts)
        return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: null polylines / lane_counts");
    if (reinterpret_cast<uintptr_t>(polylines_xy) & 7u)
        return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: polylines need 8-byte alignment");
    const bool clear = (flags & ACCV_HM_CLEAR) != 0;

    FusedLaneParams fp{};
    MultiParams& mp = fp.mp;
    long long tiles = 0;
    int used = 0;
    for (int i = 0; i < num_scales; ++i) {
        if (int rc = check_common(heatmaps[i], heights[i], widths[i], diameter_to_sigma_factor, "draw_polylines_multiscale"))
            return rc;
        if (!(strides[i] > 0.0f)) return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: stride %d is not positive", i);
        if (heights[i] == 0 || widths[i] == 0) continue;
        if (!heatmaps[i]) return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: heatmap %d is null", i);
        if (widths[i] % 4 != 0 || (reinterpret_cast<uintptr_t>(heatmaps[i]) & 15u) ||
            (size_t)heights[i] * widths[i] * sizeof(float) >= ((size_t)1 << 31))
            return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: map %d needs a width that is a multiple of 4, a 16-byte "
                                           "aligned base and planes below 2 GiB", i);
        SplatParams& p = mp.scale[used];
        p.hm = heatmaps[i];
        p.counts = lane_counts;
        p.H = heights[i];
        p.W = widths[i];
        p.n_max = lanes;
        p.factor = diameter_to_sigma_factor;
        p.k = k_scale;
        p.counts_i64 = (flags & ACCV_HM_COUNTS_I64) ? 1 : 0;
        p.stride = strides[i];
        p.radius = radius;
        p.n_groups = 0;
        p.tiles_x = (p.W + 127) / 128;
        p.tiles_y = (p.H + 15) / 16;
        p.n_tiles = (long long)batch * p.tiles_x * p.tiles_y;
        p.grid3d = 0;
        mp.tile_begin[used] = tiles;
        tiles += p.n_tiles;
        ++used;
    }
    mp.n_scales = used;
    mp.tile_begin[used] = tiles;
    if (used == 0 || tiles == 0) return ACCV_OK;
    if (tiles > INT_MAX) return accv::fail(ACCV_EINVAL, "draw_polylines_multiscale: %lld tiles exceed the grid limit", tiles);
    if (!(flags & ACCV_HM_CALLER_SCALE_ORDER)) coarse_scales_first(mp);
    seal_tile_prefix(mp);
    fp.lp.points = reinterpret_cast<const float2*>(polylines_xy);
    fp.lp.point_counts = point_counts;
    fp.lp.L = lanes;
    fp.lp.P = points;
    fp.lp.S = num_samples;
    fp.lp.p2_shift = shape.p2_shift;
    fp.lp.counts_i64 = (flags & ACCV_HM_POINT_COUNTS_I64) ? 1 : 0;
    int nt = accv::tune_get("hm_nt", -1);
    if (nt < 0) nt = (flags & ACCV_HM_WRITE_THROUGH) ? 4 : 0;
    const dim3 grid((unsigned)tiles), block(64);
    if (clear) {
        if (nt >= 2)
            hipLaunchKernelGGL((lane_raster_multi_kernel<true, 4>), grid, block, 0, stream, fp);
        else
            hipLaunchKernelGGL((lane_raster_multi_kernel<true, 0>), grid, block, 0, stream, fp);
    } else {
        if (nt >= 2)
            hipLaunchKernelGGL((lane_raster_multi_kernel<false, 4>), grid, block, 0, stream, fp);
        else
            hipLaunchKernelGGL((lane_raster_multi_kernel<false, 0>), grid, block, 0, stream, fp);
    }
    note_dispatch("lane_raster_multi_kernel", 4, 8, clear, nt >= 2 ? 4 : 0, grid, block);
    return accv::check_launch("draw_heatmap fused lane raster kernel");
}

int accv_heatmap_targets_from_boxes_f32(const float* centers_xy, const float* boxes_xyxy, long long num_objects,
                                        float stride, int32_t* out_centers, int32_t* out_radii, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (num_objects < 0) return accv::fail(ACCV_EINVAL, "targets_from_boxes: negative count");
    if (num_objects == 0) return ACCV_OK;
    if (!centers_xy || !boxes_xyxy || !out_centers || !out_radii)
        return accv::fail(ACCV_EINVAL, "targets_from_boxes: null pointer");
    if ((reinterpret_cast<uintptr_t>(boxes_xyxy) & 15u) || (reinterpret_cast<uintptr_t>(centers_xy) & 7u) ||
        (reinterpret_cast<uintptr_t>(out_centers) & 7u))
        return accv::fail(ACCV_EINVAL, "targets_from_boxes: centres need 8-byte and boxes 16-byte alignment");
    const unsigned grid = (unsigned)std::min<long long>((num_objects + 255) / 256, 4096);
    hipLaunchKernelGGL(targets_from_boxes_kernel, dim3(grid), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(centers_xy), reinterpret_cast<const float4*>(boxes_xyxy),
                       num_objects, stride, reinterpret_cast<int2*>(out_centers), out_radii);
    return accv::check_launch("targets_from_boxes");
}

int accv_heatmap_targets_from_points_f32(const float* points_xy, long long num_points, float stride, int radius,
                                         int32_t* out_centers, int32_t* out_radii, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (num_points < 0) return accv::fail(ACCV_EINVAL, "targets_from_points: negative count");
    if (num_points == 0) return ACCV_OK;
    if (!points_xy || !out_centers || !out_radii) return accv::fail(ACCV_EINVAL, "targets_from_points: null pointer");
    if ((reinterpret_cast<uintptr_t>(points_xy) & 7u) || (reinterpret_cast<uintptr_t>(out_centers) & 7u))
        return accv::fail(ACCV_EINVAL, "targets_from_points: points and centres need 8-byte alignment");
    const unsigned grid = (unsigned)std::min<long long>((num_points + 255) / 256, 4096);
    hipLaunchKernelGGL(targets_from_points_kernel, dim3(grid), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(points_xy), num_points, stride, radius,
                       reinterpret_cast<int2*>(out_centers), out_radii);
    return accv::check_launch("targets_from_points");
}

int accv_fill_f32(float* dst, size_t count, float value, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (count == 0) return ACCV_OK;
    if (!dst) return accv::fail(ACCV_EINVAL, "fill: null pointer");
    size_t head = 0;
    while (((reinterpret_cast<uintptr_t>(dst + head)) & 15u) && head < count) ++head;
    if (head) hipLaunchKernelGGL(fill_tail_kernel, dim3(1), dim3(64), 0, stream, dst, head, value);
    const size_t n4 = (count - head) / 4;
    for (size_t done = 0; done < n4;) {  // one launch unless the buffer has more than 2^31 * 256 vectors
        const size_t part = std::min(n4 - done, (size_t)0x7fffffff * 256);
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((part + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<float4*>(dst + head) + done, part, value);
        done += part;
    }
    const size_t tail = count - head - n4 * 4;
    if (tail) hipLaunchKernelGGL(fill_tail_kernel, dim3(1), dim3(64), 0, stream, dst + head + n4 * 4, tail, value);
    return accv::check_launch("fill");
}
}
