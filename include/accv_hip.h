/*
 * accv_hip.h — C-ABI of libaccv_hip.so: the MI355X (gfx950) drop-in for the native layer of ACCV-Lab's
 * per-step data/target-preparation hot path.  Plain pointers and sizes only; no torch/ATen types.
 *
 * Conventions
 *   - every entry point returns 0 on success or a negative ACCV_E* code; accv_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises, nothing allocates device memory — the caller owns every
 *     buffer including the workspace (hipGraph-capturable, re-entrant);
 *   - device pointers unless a parameter is documented as host memory.
 *
 * Each function cites the reference interface it replaces (paths relative to the ACCV-Lab checkout).
 */
#ifndef ACCV_HIP_H
#define ACCV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACCV_OK 0
#define ACCV_EINVAL (-1)   /* bad argument (shape, null pointer, unsupported dtype code) */
#define ACCV_ELAUNCH (-2)  /* HIP runtime reported an error at launch / enqueue */
#define ACCV_EWORKSPACE (-3) /* workspace too small */
#define ACCV_ERUNTIME (-4) /* other runtime failure (host allocation, thread pool, ...) */

/* flags for the draw_heatmap entry points */
#define ACCV_HM_CLEAR 1u       /* fused clear: result = max(0, splats); every pixel is written exactly once */
#define ACCV_HM_COUNTS_I64 2u  /* `counts` points to int64 (RaggedBatch.sample_sizes) instead of int32 */
#define ACCV_HM_SMALL_RADII 4u /* hint: radii of a few pixels (lane samples, key points; boxes up to ~15x15): take the
                                 kernel that walks each object's box instead of updating whole tiles.  Results do
                                 not depend on the hint; objects of any size stay correct, only slower. */
#define ACCV_HM_WRITE_THROUGH 8u /* hint: write the map with write-through non-temporal stores (sc1 nt) instead of plain
                                    ones.  Pays 1-7 % for launches that rewrite far more than L2 + Infinity Cache hold AND
                                    are compute heavy (dense batches, in-place over most of the frame); costs up to 27 %
                                    for sparse in-place launches, which is why it is not the default.  Same results. */
#define ACCV_HM_GROUP_BOXES_GIVEN 16u /* accv_draw_points_multiscale_f32 only: `workspace` already holds the group boxes
                                         (written by accv_polyline_sample_boxes); skip the box launch */

#define ACCV_HM_PLAIN_STORES 128u /* hint: plain stores for every plane.  Default: fused-clear launches store plain; in-place
                                    launches choose per plane — write-through non-temporal where the plane's objects
                                    cover >= 3/4 of its area (sum of (2r+1)^2), plain elsewhere.  Same results. */
#define ACCV_HM_TILE_ROWS_16 32u /* hint: 128 x 32 pixel wave tiles (half the waves, twice the registers per wave) */
#define ACCV_HM_TILE_ROWS_8 64u  /* hint: 128 x 16 pixel wave tiles (the default).  Same results either way. */
#define ACCV_HM_CALLER_SCALE_ORDER 256u /* multi-scale calls only: keep the caller's order of scales in the launch.  Default:
                                         * coarse scales (fewest tiles, longest per-tile work) are dispatched first.  Same
                                         * results either way. */

const char* accv_last_error(void);
int accv_version(void);
/* Which kernel instantiation and launch geometry the LAST draw_heatmap entry point called on this thread selected
 * (thread-local, e.g. "splat_kernel<PX=4,R=16,CLEAR=1,SM=0> grid(15,34,64) block(64)"); "" before the first call.
 * For benchmarks and profiles: the name a rocprofv3 kernel trace shows for that launch starts with the same text. */
const char* accv_draw_heatmap_last_dispatch(void);

/* ------------------------------------------------------------------------------------------------ H1
 * Gaussian heat-map rasteriser.
 */

/* Profiling aid (no reference counterpart): the splat kernel of the NEXT accv_draw_heatmap_flat_f32 /
 * accv_draw_heatmap_batched_f32 call made by the calling thread records `start_event` when it begins to execute and
 * `stop_event` when it has finished (hipExtLaunchKernel), i.e. the kernel's own duration, without the dispatch gap that
 * separates back-to-back launches on a stream.  Both are hipEvent_t with timing enabled (either may be NULL).  One-shot:
 * that call consumes the pair whether or not it launches (a call that returns early leaves the events unrecorded), and a
 * multi-scale draw call in between drops it; (NULL, NULL) cancels.  bench.py derives roofline.kernel_ms from it; kernel and launch parameters are unchanged. */
int accv_draw_heatmap_time_next_launch(void* start_event, void* stop_event);

/* Replaces draw_heatmap_launcher / draw_heatmap_cuda  (packages/draw_heatmap/accvlab/draw_heatmap/csrc/
 * draw_heatmap_cuda.cu:29-41,62-89; kernel include/draw_heatmap_cuda_kernel.cuh:51-74).
 * heatmaps f32[P,H,W] (in/out), centers i32[N,2] (x,y), radii i32[N], heatmap_idxes i32[N].
 * Objects whose plane index is outside [0,P) are ignored (the reference writes out of bounds).
 * workspace: accv_draw_heatmap_flat_workspace_bytes(P,N) bytes of device memory, 16-byte aligned. */
size_t accv_draw_heatmap_flat_workspace_bytes(int num_planes, int num_objects);
int accv_draw_heatmap_flat_f32(float* heatmaps, int num_planes, int height, int width, const int32_t* centers,
                               const int32_t* radii, const int32_t* heatmap_idxes, int num_objects,
                               float diameter_to_sigma_factor, float k_scale, unsigned flags, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Replaces draw_heatmap_batched_launcher / draw_heatmap_batched_classwise_launcher / draw_heatmap_batched_cuda
 * (draw_heatmap_cuda.cu:43-60,91-124,126-165; kernel cuh:76-108).
 * heatmap f32[B,H,W] (num_classes == 0, labels == NULL) or f32[B,C,H,W] (class-wise);
 * centers i32[B,Nmax,2], radii i32[B,Nmax], labels i32[B,Nmax] or NULL, counts i32[B] (or i64[B] with
 * ACCV_HM_COUNTS_I64 — removes the per-call cast of draw_heatmap_batched.py:63).
 * Labels outside [0,C) are ignored (the reference device-asserts, cuh:102). Needs no workspace. */
int accv_draw_heatmap_batched_f32(float* heatmap, int batch, int num_classes, int height, int width,
                                  const int32_t* centers, const int32_t* radii, const void* counts,
                                  const int32_t* labels, int max_num_targets, float diameter_to_sigma_factor,
                                  float k_scale, unsigned flags, void* stream);

/* Multi-scale target maps in ONE launch (BASELINE config 3; SURVEY §8 f2 fused into H1): for every scale s,
 *   (c, r) = targets_from_boxes(centers_xy, boxes_xyxy, strides[s])   — packages/draw_heatmap/tests/_test_helpers.py:20-28
 *   draw_heatmap_batched(heatmaps[s] f32[batch, heights[s], widths[s]], c, r, counts, factor, k)
 * with the float -> integer conversion done inside the kernel's culling step (no intermediate tensors).  Results are
 * identical to the per-scale calls.  heatmaps / heights / widths / strides are HOST arrays of `num_scales` (<= 4)
 * entries; centers f32[batch, Nmax, 2], boxes f32[batch, Nmax, 4] (source pixels), counts as in the batched call.
 * Every map needs width % 4 == 0, a 16-byte aligned base and planes below 2 GiB (else use the per-scale calls). */
int accv_draw_heatmap_multiscale_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                     int num_scales, int batch, const float* centers_xy, const float* boxes_xyxy,
                                     const void* counts, int max_num_targets, float diameter_to_sigma_factor,
                                     float k_scale, unsigned flags, void* stream);

/* accv_draw_heatmap_multiscale_f32 with the polyline sampler riding in the same launch (BASELINE config 3 prepares box maps and
 * lane maps every step; the sampler's few hundred workgroups cost nothing next to the box maps' tiles, a launch of their own
 * costs 5.9 us).  The first 13 arguments are those of accv_draw_heatmap_multiscale_f32 and mean the same; in addition
 * polylines_xy f32[num_polylines, points, 2] (1..64 points; point_counts[num_polylines] valid leading points, int32 or int64
 * with ACCV_HM_POINT_COUNTS_I64, null = all) are sampled at the num_samples (a multiple of 64) arc-length fractions
 * k / (num_samples - 1) into samples f32[num_polylines, num_samples, 2], and the bounding box of every 64 consecutive samples
 * goes to group_boxes f32[num_polylines * num_samples / 64, 4] — what accv_polyline_sample_boxes writes for relative distances
 * = those fractions (polyline_common.cuh:58-163 semantics; finite samples and boxes bit for bit, NaN samples are NaN with an
 * unspecified sign bit), i.e. the input of
 * accv_draw_points_multiscale_f32 with ACCV_HM_GROUP_BOXES_GIVEN, which follows on the same stream. */
int accv_draw_heatmap_multiscale_sample_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                            int num_scales, int batch, const float* centers_xy, const float* boxes_xyxy,
                                            const void* counts, int max_num_targets, float diameter_to_sigma_factor,
                                            float k_scale, unsigned flags, const float* polylines_xy, int num_polylines,
                                            int points, const void* point_counts, int num_samples, float* samples,
                                            float* group_boxes, void* stream);

/* Lane raster of all scales in TWO launches (BASELINE config 3): sampled polyline points f32[batch, N, 2]
 * (source pixels; output of accv_polyline_sample, NaN = sample of an empty polyline) are drawn into every
 * heatmaps[s] f32[batch, heights[s], widths[s]] as Gaussians of `radius` around int(p / strides[s]) — for each scale the
 * result of accv_heatmap_targets_from_points_f32 + accv_draw_heatmap_batched_f32 with ACCV_HM_SMALL_RADII.  Launch 1
 * writes the bounding box of every 64 consecutive points into `workspace` (accv_draw_points_workspace_bytes); launch 2
 * covers the tiles of all (<= 4) scales and culls first by group box, then by point.  counts[b] (int32, or int64 with
 * ACCV_HM_COUNTS_I64) = number of leading points of sample b that are drawn.  Same map constraints as the multi-scale
 * box call.  The reference has no polyline rasteriser (polyline/functions.py:27-111 only samples). */
size_t accv_draw_points_workspace_bytes(int batch, int num_points);
int accv_draw_points_multiscale_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                    int num_scales, int batch, const float* points_xy, const void* counts, int num_points,
                                    int radius, float diameter_to_sigma_factor, float k_scale, unsigned flags,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* Lane raster of all scales in ONE launch: the tile waves work from the polylines themselves — polylines_xy
 * f32[batch, lanes, points, 2] (source pixels), point_counts[batch * lanes] valid leading points per polyline (int32, or
 * int64 with ACCV_HM_POINT_COUNTS_I64; null = all), lane_counts[batch] valid leading polylines per frame (int32, or int64
 * with ACCV_HM_COUNTS_I64); every polyline is sampled at the num_samples arc-length fractions k / (num_samples - 1) (IEEE
 * float division; one sample at fraction 0 for num_samples == 1).  Result: bit for bit what accv_polyline_sample_boxes
 * (relative distances = those fractions, for every polyline) + accv_draw_points_multiscale_f32 write, without the sampler
 * launch, the sample buffer and the workspace: a wave tests the SEGMENTS of the frame's polylines against its tile and
 * repeats the sampler's arithmetic (polyline_common.cuh:58-163 semantics; csrc/polyline_arith.h) only for the stretch of
 * samples on segments in reach.  That work is repeated per tile and scale, so the launch saved pays for SPARSE lane sets
 * (one or two polylines per frame: 22.6-24.2 -> 17.8-18.6 us on config 3's maps) and the kernel takes only those:
 * accv_draw_polylines_fused_applicable(...) == 1 for 1..64 points per polyline, lanes x P2 <= 64 with P2 = points rounded
 * up to a power of two (>= 4), num_samples <= (points - 1) x P2 / 2, fine scales in the majority of the tiles; other shapes
 * return ACCV_EINVAL — use the two-launch composition for them.  Same map constraints as the calls above. */
#define ACCV_HM_POINT_COUNTS_I64 512u /* accv_draw_polylines_multiscale_f32: `point_counts` points to int64 */
int accv_draw_polylines_fused_applicable(const int* heights, const int* widths, int num_scales, int batch, int lanes,
                                         int points, int num_samples);
int accv_draw_polylines_multiscale_f32(float* const* heatmaps, const int* heights, const int* widths, const float* strides,
                                       int num_scales, int batch, const float* polylines_xy, int lanes, int points,
                                       const void* point_counts, const void* lane_counts, int num_samples, int radius,
                                       float diameter_to_sigma_factor, float k_scale, unsigned flags, void* stream);

/* Target-prep front end: float centres [n,2] (x,y) and boxes [n,4] (x0,y0,x1,y1) in source pixels ->
 * int32 centres [n,2] = int(c / stride) and radii [n] = max(1, int(ceil(min edge distance / stride))) in ONE kernel.
 * Semantics of get_centers_and_radii (packages/draw_heatmap/tests/_test_helpers.py:20-28; the DALI path uses
 * get_center_from_bboxes/get_radii_from_bboxes).  fp32 with IEEE division. */
int accv_heatmap_targets_from_boxes_f32(const float* centers_xy, const float* boxes_xyxy, long long num_objects,
                                        float stride, int32_t* out_centers, int32_t* out_radii, void* stream);

/* Lane front end (the "polyline raster" of BASELINE config 3): sampled polyline points f32[n,2] (x,y) in
 * source pixels (output of accv_polyline_sample) -> int32 centres [n,2] = int(p / stride) and radii [n] = `radius`;
 * NaN points (samples of empty polylines, polyline_kernels.cuh:216-245) get radius -1, which the splat never draws.
 * The reference has no rasteriser for polylines; this feeds its sampler's output to accv_draw_heatmap_batched_f32. */
int accv_heatmap_targets_from_points_f32(const float* points_xy, long long num_points, float stride, int radius,
                                         int32_t* out_centers, int32_t* out_radii, void* stream);

/* ------------------------------------------------------------------------------------------------ H2
 * Ragged-batch kernels (batching_helpers).  Shapes are given after flattening all batch dimensions to
 * `batch` and all trailing data dimensions to `row_bytes` (= elements per index * element size): the
 * indexed dimension sits between them.  COPY entry points are dtype-agnostic byte movers (bit-exact).
 * `indices` is [batch, idx_stride] of int32 (idx_i64 == 0) or int64; only the first `w_idx` slots of a
 * row are considered and of those only j < counts[i].  Negative indices wrap once; indices still out
 * of range are skipped and, if `err_counter` (device int*) is non-null, counted there (the reference
 * device-asserts: batched_indexing_access_cuda_impl.cu:79,143,185).  `counts` is int32 or int64.
 */

/* dst[i, j, :] = src[i, indices[i,j], :]   — forward of indexing_kernel
 * (batched_indexing_access_cuda_impl.cu:52-113, host batched_indexing_access_cuda.cpp:54-86).
 * src [batch, w_src, row], dst [batch, w_idx, row]; dst must be pre-filled by the caller. */
int accv_ragged_gather(const void* src, void* dst, const void* indices, const void* counts, long long batch,
                       long long w_src, long long w_idx, long long idx_stride, long long row_bytes, int idx_i64,
                       int counts_i64, int* err_counter, void* stream);

/* The same gather, but the kernel also writes the filler: dst[i, j, :] = `fill_bits` pattern (element of `elem_size`
 * in {1,2,4,8} bytes, little endian) for j >= counts[i] and for skipped out-of-range indices, so dst may be
 * uninitialised — one launch for torch::full + indexing_kernel of batched_indexing_access_cuda.cpp:82-85. */
int accv_ragged_gather_fill(const void* src, void* dst, const void* indices, const void* counts, long long batch,
                            long long w_src, long long w_idx, long long idx_stride, long long row_bytes,
                            uint64_t fill_bits, int elem_size, int idx_i64, int counts_i64, int* err_counter,
                            void* stream);

/* dst[i, indices[i,j], :] = src[i, j, :]   — overwrite direction of indexing_kernel (cpp:88-146 with
 * backward_accumulate == false).  src [batch, w_idx, row], dst [batch, w_dst, row]. */
int accv_ragged_scatter(const void* src, void* dst, const void* indices, const void* counts, long long batch,
                        long long w_idx, long long idx_stride, long long w_dst, long long row_bytes, int idx_i64,
                        int counts_i64, int* err_counter, void* stream);

/* dst[i, dst_indices[i,j], :] = src[i, src_indices[i,j], :] — map_values_by_index_pairs_kernel
 * (cu:115-160, cpp:170-200), overwrite mode. */
int accv_ragged_map_pairs(const void* src, void* dst, const void* src_indices, const void* dst_indices,
                          const void* counts, long long batch, long long w_src, long long w_idx, long long idx_stride,
                          long long w_dst, long long row_bytes, int idx_i64, int counts_i64, int* err_counter,
                          void* stream);

/* dst[i, indices[i,j], :] = const — insert_const_at_indices_kernel (cu:162-194; also builds bool masks,
 * cpp:202-228).  `elem_bits` holds the element's byte pattern (little endian) of `elem_size` in {1,2,4,8}. */
int accv_ragged_insert_const(void* dst, const void* indices, const void* counts, long long batch, long long w_idx,
                             long long idx_stride, long long w_dst, long long row_bytes, uint64_t elem_bits,
                             int elem_size, int idx_i64, int counts_i64, int* err_counter, void* stream);

/* data[i, j, :] = filler for j >= counts[i] — set_ragged_batch_padded_to_filler_value_kernel
 * (cu:196-213, cpp:230-245; CPU twin batched_indexing_access_cpu_impl.cpp:27-67). */
int accv_ragged_pad_fill(void* data, const void* counts, long long batch, long long width, long long row_bytes,
                         uint64_t elem_bits, int elem_size, int counts_i64, void* stream);

/* dst[i, dst_indices[i,j], k] += src[i, (src_indices ? src_indices[i,j] : j), k] with atomics — the
 * accumulate mode of indexing_kernel / map_values_by_index_pairs_kernel (cu:39-50, 103-108, 152-156); the
 * caller clears the touched slots first where "set first, then add" semantics are required.
 * acc_dtype: 0 f32, 1 f64, 2 i32, 3 i64, 4 f16, 5 bf16;  row_elems = elements per index. */
int accv_ragged_accumulate(const void* src, void* dst, const void* src_indices_or_null, const void* dst_indices,
                           const void* counts, long long batch, long long w_src, long long w_idx, long long idx_stride,
                           long long w_dst, long long row_elems, int acc_dtype, int idx_i64, int counts_i64,
                           int* err_counter, void* stream);

/* Ragged compaction front end (replaces the torch boolean indexing of batched_bool_indexing.py:195-221 and
 * batched_processing_py.py:245-268, 577-628): for every row the positions of non-zero mask bytes, in order,
 * as int64, zero-filled behind; out_sizes[i] = number of hits.  Only the first valid_counts[i] columns are
 * looked at when valid_counts is given.  out_indices is [batch, width]. */
int accv_ragged_mask_to_indices(const void* mask_u8, const void* valid_counts_or_null, int valid_i64, long long batch,
                                long long width, long long* out_indices, long long* out_sizes, void* stream);
/* The same with a caller-provided workspace of accv_ragged_mask_to_indices_workspace_bytes(batch, width) bytes (0 = not
 * needed): FEW, VERY WIDE rows (a dense anchor mask of a small batch) are cut into 4096-byte segments handled by one
 * workgroup each instead of one workgroup per row — in ONE launch for rows of up to 4 segments (every segment workgroup
 * counts its whole row itself; the workspace is not touched), in a count launch + a write launch through the workspace
 * beyond that.  Same results; without (enough) workspace the one-workgroup-per-row kernels run for the longer rows. */
size_t accv_ragged_mask_to_indices_workspace_bytes(long long batch, long long width);
int accv_ragged_mask_to_indices_ws(const void* mask_u8, const void* valid_counts_or_null, int valid_i64, long long batch,
                                   long long width, long long* out_indices, long long* out_sizes, void* workspace,
                                   size_t workspace_bytes, void* stream);


/* F3 — matched gather + element-wise loss + masked per-sample sum in ONE launch.  The caller pattern of
 * packages/batching_helpers/example/loss_computation.py:37-43 (batched_indexing_access of ground truth and prediction
 * through the two index lists of a matching) and :85-86 (sum_over_targets of the per-object loss):
 *   out[i] = sum_{j < counts[i]} w * sum_k l(a[i, idx_a[i,j], k] - b[i, idx_b[i,j], k]),  w = weights[i, idx_a[i,j]] or 1
 * kind: 0 = |d| (L1), 1 = d^2, 2 = smooth-L1 with `beta` (torch.nn.functional.smooth_l1_loss).  a f32[batch, w_a, row],
 * b f32[batch, w_b, row], indices [batch, idx_stride] int32/int64 (first w_idx slots), counts int32/int64, out f32[batch].
 * Deterministic (fixed-order tree reduction per sample).  Pairs with an out-of-range index contribute nothing. */
int accv_matched_pair_reduce_f32(const float* a, const float* b, const void* idx_a, const void* idx_b, const void* counts,
                                 const float* weights_or_null, long long batch, long long w_a, long long w_b,
                                 long long w_idx, long long idx_stride, long long row_elems, int kind, float beta,
                                 int idx_i64, int counts_i64, float* out, void* stream);
/* Its backward: grad_a[i, idx_a[i,j], k] += g[i] * w * l'(d), grad_b[...] -= the same, grad_w[i, idx_a[i,j]] += g[i] * l(d)
 * (float atomics; the gradient tensors must be zero-initialised by the caller; any of them may be NULL). */
int accv_matched_pair_reduce_bwd_f32(const float* a, const float* b, const void* idx_a, const void* idx_b,
                                     const void* counts, const float* weights_or_null, const float* grad_out,
                                     long long batch, long long w_a, long long w_b, long long w_idx,
                                     long long idx_stride, long long row_elems, int kind, float beta, int idx_i64,
                                     int counts_i64, float* grad_a_or_null, float* grad_b_or_null,
                                     float* grad_w_or_null, void* stream);

/* Round 3: the same two launches for every dtype the replaced gathers accept and for the reference example's own per-object
 * losses.  dtype: 0 f32, 1 f16, 2 bf16, 3 f64 — of a, b and weights; arithmetic, `out` and the gradient buffers are f32 (f64 for
 * f64 data).  kind 0-2 as above, plus
 *   3 = 1 - IoU of (x0, y0, x1, y1) boxes (row_elems == 4) with negative intersection extents clamped to 0 and the union
 *       clamped to `eps`: _per_object_bbox_overlap_loss, packages/batching_helpers/example/loss_computation.py:240-274;
 *       the backward follows torch's autograd of that code (clamps pass no gradient, max / min ties split it evenly);
 *   4 = L1 between one-hot class labels and scores: `a` holds INTEGER labels [batch, w_a] (int32, or int64 with
 *       ACCV_MP_LABELS_I64), b the scores [batch, w_b, row_elems]; sum_c |[c == label] - b[..., c]|
 *       (loss_computation.py:37-43 class branch, :225-238); a label outside [0, row_elems) matches no class.
 * flags: ACCV_MP_IDX_I64 / ACCV_MP_COUNTS_I64 / ACCV_MP_LABELS_I64. */
#define ACCV_MP_IDX_I64 1u
#define ACCV_MP_COUNTS_I64 2u
#define ACCV_MP_LABELS_I64 4u
int accv_matched_pair_reduce(const void* a, const void* b, const void* idx_a, const void* idx_b, const void* counts,
                             const void* weights_or_null, long long batch, long long w_a, long long w_b, long long w_idx,
                             long long idx_stride, long long row_elems, int kind, int dtype, float beta, float eps,
                             unsigned flags, void* out, void* stream);
int accv_matched_pair_reduce_bwd(const void* a, const void* b, const void* idx_a, const void* idx_b, const void* counts,
                                 const void* weights_or_null, const void* grad_out, long long batch, long long w_a,
                                 long long w_b, long long w_idx, long long idx_stride, long long row_elems, int kind,
                                 int dtype, float beta, float eps, unsigned flags, void* grad_a_or_null,
                                 void* grad_b_or_null, void* grad_w_or_null, void* stream);

/* combine_data / split on device (batched_processing_py.py:410-423, ragged_batch.py:870-934):
 * unpack == 0: padded[i, j, :] = flat[offsets[i] + j, :] for j < sizes[i], zero bytes elsewhere;
 * unpack != 0: the inverse copy (flat <- padded, valid entries only).  offsets/sizes are device int64. */
int accv_ragged_pack(const void* flat, void* padded, const long long* offsets, const long long* sizes, long long batch,
                     long long width, long long row_bytes, int unpack, void* stream);

/* ------------------------------------------------------------------------------------------------ H3
 * Multi-tensor copier: pack planner, pinned arena, threaded staging + chunked host->device transfer, and a
 * device-side coalescing kernel.  Tensors, streams and events stay with the caller.
 */

/* Byte layout of the packed chunks — integer-exact counterpart of compute_pack_plan / layout_packed_offsets
 * (packages/multi_tensor_copier/accvlab/multi_tensor_copier/csrc/multi_tensor_copier.cpp:419-433, 513-549,
 * 553-590).  candidate[i] != 0 marks leaves that satisfy make_pack_candidate (:481-507: host tensor, target
 * is a GPU, contiguous, 0 < bytes <= 262144).  required_align = round_up(max(min_align, elem), elem); buckets
 * {16,8,4,2,1} are laid out in that order, insertion order inside a bucket; a new chunk starts when
 * offset + bytes > max_chunk_bytes and the chunk is not empty.  Fewer than 2 packed leaves => nothing is packed
 * (*out_num_chunks == 0, all offsets -1).  out_chunk_sizes must have room for n entries. Host arrays. */
int accv_mtc_plan(long long n, const long long* nbytes, const int* elem_size, const unsigned char* candidate,
                  long long min_align, long long max_chunk_bytes, long long* out_offset, long long* out_chunk,
                  long long* out_chunk_sizes, long long* out_num_chunks);

/* Pinned (page-locked) host buffers from a size-class cache over hipHostMalloc; replaces the per-call pinned
 * allocations of allocate_staging_buffers (multi_tensor_copier.cpp:597-641).  acquire returns NULL on failure. */
void* accv_pinned_acquire(size_t bytes);
void accv_pinned_release(void* ptr);
void accv_pinned_trim(void);
size_t accv_pinned_total_bytes(void);
int accv_mtc_worker_count(void);

/* DataLoader hook: host-only pack of `n_items` buffers into `dst` at the byte offsets of accv_mtc_plan
 * (single chunk).  Plain memcpy loop on the calling thread; makes no HIP call and uses no worker pool, so it is safe
 * inside forked DataLoader worker processes.  No reference counterpart (the reference packs in the consumer process,
 * fill_cpu_staging_buffers, multi_tensor_copier.cpp:647-679). */
int accv_mtc_pack_host(long long n_items, const void* const* src, const long long* nbytes, const long long* offset,
                       void* dst, long long dst_bytes);

/* fill_cpu_staging_buffers + enqueue_packed_transfer (multi_tensor_copier.cpp:647-679, 683-730): per chunk c,
 * memcpy src[i] -> staging[c] + offset[i] for i in order[item_begin[c] .. item_begin[c+1]) on up to `threads`
 * workers, then one hipMemcpyAsync(device[c] <- staging[c], chunk_bytes[c]) on `stream` (skipped when device[c]
 * is NULL).  All arrays are host memory; src/staging are host pointers, device[c] device pointers. */
int accv_mtc_stage_h2d(long long n_items, const void* const* src, const long long* nbytes, const long long* offset,
                       const long long* order, long long n_chunks, const long long* item_begin, void* const* staging,
                       void* const* device, const long long* chunk_bytes, void* stream, int threads);

/* The same staging + transfers on a native orchestration thread of the library (counterpart of the reference's
 * CopyThreadPool worker running schedule_copies, multi_tensor_copier.cpp:288-349, 863-883): the argument arrays are copied,
 * the job is queued and `*ticket_out` returned at once; no Python and no interpreter lock are involved in the work.
 * accv_mtc_async_wait(ticket) blocks until every transfer of the job has been ENQUEUED on `stream` (completion is the
 * caller's stream event) and returns the job's status; accv_mtc_async_poll(ticket) is 1 when it has finished, else 0.
 * A ticket is forgotten by the wait that returns its status.  device_index < 0 selects no device (a staging-only job whose
 * device[] entries are all NULL). */
int accv_mtc_stage_h2d_async(long long n_items, const void* const* src, const long long* nbytes, const long long* offset,
                             const long long* order, long long n_chunks, const long long* item_begin,
                             void* const* staging, void* const* device, const long long* chunk_bytes, void* stream,
                             int threads, int device_index, long long* ticket_out);
int accv_mtc_async_wait(long long ticket);
int accv_mtc_async_poll(long long ticket);
/* Orderly end of that thread (counterpart of ~CopyThreadPool joining its workers, multi_tensor_copier.cpp:300-312): runs
 * what is still queued, stops and joins; later accv_mtc_stage_h2d_async calls fail with ACCV_ERUNTIME.  Idempotent.  The
 * python package calls it from atexit, i.e. before the HIP runtime is torn down.  A forked child starts with fresh
 * (empty) pool / arena / orchestrator state: the parent's threads do not exist there. */
void accv_mtc_shutdown(void);
/* Number of tickets the orchestrator still remembers.  Bounded: a ticket is forgotten by the wait that returns its
 * status, and finished tickets nobody waited for are dropped once 1024 later jobs were submitted. */
long long accv_mtc_async_tickets_held(void);

/* One kernel that gathers (scatter == 0) many small device tensors into `packed`, or fans `packed` out again
 * (scatter != 0).  items: array of {const void* ptr; long long offset_in_packed; long long nbytes;} readable by
 * the device.  New component (the reference copies device tensors one by one, multi_tensor_copier.cpp:775-820). */
int accv_mtc_coalesce(const void* items, long long n_items, void* packed, int scatter, void* stream);

/* hipMemcpyAsync wrapper: kind 1 = H2D, 2 = D2H, 3 = D2D, anything else = default. */
int accv_memcpy_async(void* dst, const void* src, size_t bytes, int kind, void* stream);

/* ------------------------------------------------------------------------------------------------ lane_helpers
 * Batched polyline arc-length interpolation / lengths.  Replaces the four entry points of
 * packages/lane_helpers/ext_impl/polyline/src/polyline.cpp:101-398 (polyline_interpolation, _polyline_lengths and
 * their _var_size_batch forms; kernels include/polyline_kernels.cuh:390-455, semantics include/polyline_common.cuh
 * :58-163).  points [batch, max_points, dims], distances [batch, max_distances] (same dtype), optional per-polyline
 * counts (int32/int64; NULL = all valid).  out_points [batch, max_distances, dims] and/or out_lengths [batch] may be
 * NULL.  dtype: 0 f32, 1 f64, 2 f16, 3 bf16 (accumulation fp32, fp64 for f64).  relative != 0: distances are
 * fractions of the total length.  Polylines too long for LDS need accv_polyline_scratch_bytes() of device scratch. */
size_t accv_polyline_scratch_bytes(long long batch, int max_points, int dtype);
int accv_polyline_sample(const void* points, const void* distances, const void* point_counts, const void* dist_counts,
                         void* out_points, void* out_lengths, long long batch, int max_points, int max_distances,
                         int num_dims, int dtype, int counts_i64, int relative, void* scratch, size_t scratch_bytes,
                         void* stream);

/* The same operation on HOST memory (CPU tensors): counterpart of the reference's CPU implementation
 * (ext_impl/polyline/src/polyline_cpu.cpp:28-132): float32 / float64 only (dtype 0 / 1), double accumulation
 * (at::acc_type<dtype, false>), polylines split over up to `threads` host threads (0 = automatic).  No HIP call. */
int accv_polyline_sample_host(const void* points, const void* distances, const void* point_counts, const void* dist_counts,
                              void* out_points, void* out_lengths, long long batch, int max_points, int max_distances,
                              int num_dims, int dtype, int counts_i64, int relative, int threads);

/* The same sampler with one more output (float32 samples of 2-D points only): out_group_boxes f32[batch, ceil(Q/64), 4] =
 * (xmin, ymin, xmax, ymax) of every 64 consecutive samples of a polyline, NaN samples ignored, (+inf, +inf, -inf, -inf) for
 * a group without valid samples — what accv_draw_points_multiscale_f32 culls by (ACCV_HM_GROUP_BOXES_GIVEN saves its own
 * box launch).  NULL = plain accv_polyline_sample. */
int accv_polyline_sample_boxes(const void* points, const void* distances, const void* point_counts, const void* dist_counts,
                               void* out_points, void* out_lengths, float* out_group_boxes, long long batch, int max_points,
                               int max_distances, int num_dims, int dtype, int counts_i64, int relative, void* scratch,
                               size_t scratch_bytes, void* stream);

/* Streaming fill used by bench.py as the measured write-bandwidth ceiling (not part of the reference API). */
int accv_fill_f32(float* dst, size_t count, float value, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACCV_HIP_H */
