// Host-side fast path of accvlab.lane_helpers.polyline for CUDA tensors: argument checks, result allocation, current stream and
// ONE call into the C-ABI of libaccv_hip.so (accv_polyline_sample) in C++.  The kernel behind interpolate / lengths runs for
// 2-4 us at lane-sized inputs, so the python formulation of these steps (7.5 us per call) WAS the operator cost
// (profiles/r02_bench_published.jsonl).  Plumbing only — no device code, no HIP calls of its own.  Anything unusual (CPU or
// non-contiguous tensors, mismatching dtypes / devices, another device current, shapes the reference rejects) makes the
// functions DECLINE (return None): the python implementation then runs and raises the reference's error messages
// (ext_impl/polyline/src/polyline.cpp:40-81).
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include "accv_hip.h"  // prototypes only: the entry points are called through addresses handed over by the python side

namespace py = pybind11;

namespace {

struct NativeApi {
    decltype(&accv_polyline_sample) sample = nullptr;
    decltype(&accv_polyline_scratch_bytes) scratch_bytes = nullptr;
    decltype(&accv_last_error) last_error = nullptr;
} g_api;

void bind_native(uint64_t sample, uint64_t scratch_bytes, uint64_t last_error)
{
    g_api.sample = reinterpret_cast<decltype(g_api.sample)>(sample);
    g_api.scratch_bytes = reinterpret_cast<decltype(g_api.scratch_bytes)>(scratch_bytes);
    g_api.last_error = reinterpret_cast<decltype(g_api.last_error)>(last_error);
}

inline int dtype_code(const at::Tensor& t)
{
    switch (t.scalar_type()) {
        case at::kFloat: return 0;
        case at::kDouble: return 1;
        case at::kHalf: return 2;
        case at::kBFloat16: return 3;
        default: return -1;
    }
}
inline bool plain_cuda(const at::Tensor& t)
{
    return t.defined() && t.is_cuda() && t.is_contiguous() && !t.requires_grad() &&
           (int)t.get_device() == (int)c10::hip::current_device();
}

// want_points: samples [B, Q, D] at `distances`; else lengths [B].  None = declined.
py::object run(const at::Tensor& points, const c10::optional<at::Tensor>& distances, bool relative, bool want_points)
{
    if (!g_api.sample || !g_api.scratch_bytes || !plain_cuda(points) || points.dim() != 3) return py::none();
    const int code = dtype_code(points);
    if (code < 0) return py::none();
    const int64_t b = points.size(0), pmax = points.size(1), dims = points.size(2);
    int64_t qmax = 0;
    const void* dist_ptr = nullptr;
    if (want_points) {
        if (!distances.has_value()) return py::none();
        const at::Tensor& d = *distances;
        if (!plain_cuda(d) || d.dim() != 2 || d.size(0) != b || d.scalar_type() != points.scalar_type() ||
            d.get_device() != points.get_device())
            return py::none();
        qmax = d.size(1);
        dist_ptr = d.data_ptr();
    }
    if (b == 0 || pmax > INT_MAX || qmax > INT_MAX || dims > INT_MAX || (want_points && (qmax == 0 || dims == 0))) return py::none();
    at::Tensor out = want_points ? at::empty({b, qmax, dims}, points.options()) : at::empty({b}, points.options());
    const size_t sb = g_api.scratch_bytes(b, (int)pmax, code);
    at::Tensor scratch;
    if (sb) scratch = at::empty({(int64_t)sb}, points.options().dtype(at::kByte));
    void* stream = c10::hip::getCurrentHIPStream(points.get_device()).stream();
    const int rc = g_api.sample(points.data_ptr(), dist_ptr, nullptr, nullptr, want_points ? out.data_ptr() : nullptr,
                                want_points ? nullptr : out.data_ptr(), b, (int)pmax, (int)qmax, (int)dims, code, 0,
                                relative ? 1 : 0, sb ? scratch.data_ptr() : nullptr, sb, stream);
    if (rc != 0) {
        const char* msg = g_api.last_error ? g_api.last_error() : "";
        TORCH_CHECK(false, "polyline: ", msg ? msg : "error", " (status ", rc, ")");
    }
    return py::cast(out);
}

py::object interpolate(const at::Tensor& points, const at::Tensor& distances, bool relative)
{
    return run(points, distances, relative, true);
}
py::object lengths(const at::Tensor& points) { return run(points, c10::nullopt, false, false); }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("bind_native", &bind_native);
    m.def("interpolate", &interpolate);
    m.def("lengths", &lengths);
}
