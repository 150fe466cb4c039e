// Floating-point steps of the f32 polyline sampler that a second kernel has to reproduce BIT FOR BIT: the fused lane raster
// (draw_heatmap.hip, lane_body) evaluates the samples of a polyline inside the tile wave instead of reading what
// polyline_kernel wrote, and both must land on the same pixels.  hipcc contracts a * b + c into an fma where it likes
// (-ffp-contract=fast is the HIP default), so the few expressions whose rounding matters are spelled out here once —
// explicit fma where the sampler has always used one, contraction switched off where it has not (checked against the ISA
// of the round-2 sampler: v_fmac for the squared length, v_pk_mul + v_add for the interpolation) — and used by both.
// Semantics: packages/lane_helpers/ext_impl/polyline/include/polyline_common.cuh:58-163 of the reference.
#pragma once
#include <hip/hip_runtime.h>

namespace accv_poly {

// squared length of a 2-D segment, accumulated coordinate by coordinate: acc = fma(diff, diff, acc) starting from 0
__device__ __forceinline__ float seg_length2_step(float acc, float diff) { return __builtin_fmaf(diff, diff, acc); }

// weights of the two end points for the query distance d inside [d0, d1] (len = d1 - d0 >= epsilon)
__device__ __forceinline__ void lerp_weights(float d, float d0, float d1, float len, float& w0, float& w1)
{
#pragma clang fp contract(off)
    w1 = (d - d0) / len;
    w0 = (d1 - d) / len;
}
// one interpolated coordinate: two rounded products, one rounded sum
__device__ __forceinline__ float lerp_coord(float a, float w0, float c, float w1)
{
#pragma clang fp contract(off)
    return a * w0 + c * w1;
}
// query distance of a relative query
__device__ __forceinline__ float scale_query(float fraction, float total)
{
#pragma clang fp contract(off)
    return fraction * total;
}

}  // namespace accv_poly
