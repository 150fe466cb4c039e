// Host-side fast path of accvlab.draw_heatmap.draw_heatmap_batched for the plain case: the reference launcher's checks
// (draw_heatmap_cuda.cu:91-165: CUDA + contiguous, float32 map, int32 objects, matching extents), the current stream and ONE
// call into the C-ABI of libaccv_hip.so, in C++.  On detection-head-sized maps the kernel runs for 3-4 us and the python
// formulation of these steps costs 8-10 us per call (profiles/r02_bench_published.jsonl).  Plumbing only — no device code, no
// HIP calls of its own.  Anything unusual (CPU or non-contiguous tensors, other dtypes, mismatching shapes, another device
// current) makes the function DECLINE (return false): the python implementation then runs and raises the reference's errors.
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include "accv_hip.h"  // prototypes only: the entry points are called through addresses handed over by the python side

namespace py = pybind11;

namespace {

struct NativeApi {
    decltype(&accv_draw_heatmap_batched_f32) batched = nullptr;
    decltype(&accv_last_error) last_error = nullptr;
} g_api;

void bind_native(uint64_t batched, uint64_t last_error)
{
    g_api.batched = reinterpret_cast<decltype(g_api.batched)>(batched);
    g_api.last_error = reinterpret_cast<decltype(g_api.last_error)>(last_error);
}

inline bool plain(const at::Tensor& t, int device)
{
    return t.defined() && t.is_cuda() && t.is_contiguous() && (int)t.get_device() == device;
}

// true = drawn; false = declined (nothing was done)
bool draw_batched(const at::Tensor& heatmap, const at::Tensor& centers, const at::Tensor& radii, const at::Tensor& counts,
                  const c10::optional<at::Tensor>& labels, double factor, double k_scale, uint64_t flags)
{
    if (!g_api.batched || !heatmap.defined() || !heatmap.is_cuda() || heatmap.dim() < 3) return false;
    const int dev = (int)heatmap.get_device();
    if (dev != (int)c10::hip::current_device()) return false;
    if (!plain(heatmap, dev) || !plain(centers, dev) || !plain(radii, dev) || !plain(counts, dev)) return false;
    if (heatmap.scalar_type() != at::kFloat || centers.scalar_type() != at::kInt || radii.scalar_type() != at::kInt) return false;
    unsigned f = (unsigned)flags;
    if (counts.scalar_type() == at::kLong)
        f |= ACCV_HM_COUNTS_I64;
    else if (counts.scalar_type() != at::kInt)
        return false;
    if (centers.dim() != 3 || centers.size(2) != 2 || radii.dim() != 2 || counts.dim() != 1) return false;
    const int64_t batch = heatmap.size(0), n_max = radii.size(1);
    if (centers.size(0) != batch || radii.size(0) != batch || counts.size(0) != batch || centers.size(1) != n_max) return false;
    int64_t n_classes = 0, h, w;
    const void* lab = nullptr;
    if (labels.has_value()) {
        const at::Tensor& l = *labels;
        if (!plain(l, dev) || l.scalar_type() != at::kInt || l.dim() != 2 || l.size(0) != batch || l.size(1) != n_max) return false;
        if (heatmap.dim() != 4 || heatmap.size(1) < 1) return false;
        n_classes = heatmap.size(1);
        h = heatmap.size(2);
        w = heatmap.size(3);
        lab = l.data_ptr();
    } else {
        if (heatmap.dim() != 3) return false;
        h = heatmap.size(1);
        w = heatmap.size(2);
    }
    if (batch > INT_MAX || n_classes > INT_MAX || h > INT_MAX || w > INT_MAX || n_max > INT_MAX) return false;
    void* stream = c10::hip::getCurrentHIPStream(dev).stream();
    const int rc = g_api.batched(heatmap.data_ptr<float>(), (int)batch, (int)n_classes, (int)h, (int)w,
                                 static_cast<const int32_t*>(centers.data_ptr()), static_cast<const int32_t*>(radii.data_ptr()),
                                 counts.data_ptr(), static_cast<const int32_t*>(lab), (int)n_max, (float)factor, (float)k_scale, f,
                                 stream);
    if (rc != 0) {
        const char* msg = g_api.last_error ? g_api.last_error() : "";
        TORCH_CHECK(false, "draw_heatmap_batched: ", msg ? msg : "error", " (status ", rc, ")");
    }
    return true;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("bind_native", &bind_native);
    m.def("draw_batched", &draw_batched);
}
