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

const char* accv_last_error(void);
int accv_version(void);

/* ------------------------------------------------------------------------------------------------ H1
 * Gaussian heat-map rasteriser.
 */

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

/* Streaming fill used by bench.py as the measured write-bandwidth ceiling (not part of the reference API). */
int accv_fill_f32(float* dst, size_t count, float value, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACCV_HIP_H */
