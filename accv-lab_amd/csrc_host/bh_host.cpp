// Host-side fast path of accvlab.batching_helpers for the per-sample python loops of the reference
// (combine_data: one slice-assign per sample, batched_processing_py.py:410-427; RaggedBatch.split: one view per sample,
// ragged_batch.py:870-934): the same loops in C++ over ATen, so a 64-sample pack / split costs one python call.
// Plumbing only — no device code, no HIP calls.  Anything unusual (mixed devices or dtypes, gradients, non-tensor
// leaves) makes these functions decline (return None), and the python implementation handles it and raises the
// reference's error messages.
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <functional>

#include <cstring>
#include <vector>

#include "accv_hip.h"  // prototypes only: the entry points are called through addresses handed over by the python side

namespace py = pybind11;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Native fast path of the hottest ragged operators: argument checks, result allocation, current stream and ONE call
// into the C-ABI of libaccv_hip.so, all in C++ (~1.5 us per call; the python formulation of the same steps costs
// 6-11 us, more than the kernels run).  The function addresses come from the ctypes handle of the python package
// (bind_native), so there is one copy of the library.  Anything unusual (other device current, non-contiguous or CPU
// tensors, unsupported dtypes) makes these functions DECLINE (return None / false): the python implementation then
// runs and raises the reference's error messages.
struct NativeApi {
    decltype(&accv_ragged_gather) gather = nullptr;
    decltype(&accv_ragged_scatter) scatter = nullptr;
    decltype(&accv_ragged_gather_fill) gather_fill = nullptr;
    decltype(&accv_last_error) last_error = nullptr;
    decltype(&accv_ragged_mask_to_indices_ws) mask_to_indices_ws = nullptr;
    decltype(&accv_ragged_mask_to_indices_workspace_bytes) mask_to_indices_ws_bytes = nullptr;
} g_api;

void bind_mask_to_indices(uint64_t fn, uint64_t ws_bytes)
{
    g_api.mask_to_indices_ws = reinterpret_cast<decltype(g_api.mask_to_indices_ws)>(fn);
    g_api.mask_to_indices_ws_bytes = reinterpret_cast<decltype(g_api.mask_to_indices_ws_bytes)>(ws_bytes);
}

void bind_native(uint64_t gather, uint64_t scatter, uint64_t gather_fill, uint64_t last_error)
{
    g_api.gather = reinterpret_cast<decltype(g_api.gather)>(gather);
    g_api.scatter = reinterpret_cast<decltype(g_api.scatter)>(scatter);
    g_api.gather_fill = reinterpret_cast<decltype(g_api.gather_fill)>(gather_fill);
    g_api.last_error = reinterpret_cast<decltype(g_api.last_error)>(last_error);
}

inline bool plain_cuda(const at::Tensor& t) { return t.defined() && t.is_cuda() && t.is_contiguous(); }
inline int index_code(const at::Tensor& t) { return t.scalar_type() == at::kLong ? 1 : (t.scalar_type() == at::kInt ? 0 : -1); }
inline bool on_current_device(const at::Tensor& t) { return (int)t.get_device() == (int)c10::hip::current_device(); }
inline void* stream_of(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.get_device()).stream(); }
inline void check_status(int rc, const char* what)
{
    if (rc != 0) {
        const char* msg = g_api.last_error ? g_api.last_error() : "";
        TORCH_CHECK(false, what, ": ", msg ? msg : "error", " (status ", rc, ")");
    }
}
inline int64_t row_elems(const at::Tensor& t, int64_t first)
{
    int64_t n = 1;
    for (int64_t d = first; d < t.dim(); ++d) n *= t.size(d);
    return n;
}

// out[i, j] = src[i, indices[i, j]] for j < counts[i], j < w_idx (single batch dimension); false = declined
bool gather_rows(const at::Tensor& src, const at::Tensor& indices, const at::Tensor& counts, int64_t w_idx, const at::Tensor& out)
{
    if (!g_api.gather || !plain_cuda(src) || !plain_cuda(indices) || !plain_cuda(counts) || !plain_cuda(out)) return false;
    if (src.dim() < 2 || indices.dim() != 2 || index_code(indices) < 0 || index_code(counts) < 0) return false;
    if (!on_current_device(src) || indices.get_device() != src.get_device() || out.get_device() != src.get_device()) return false;
    if (out.numel() == 0 || w_idx == 0) return true;
    check_status(g_api.gather(src.data_ptr(), out.data_ptr(), indices.data_ptr(), counts.data_ptr(), src.size(0), src.size(1),
                              w_idx, indices.size(1), row_elems(src, 2) * (int64_t)src.element_size(), index_code(indices),
                              index_code(counts), nullptr, stream_of(src)),
                 "gather_rows");
    return true;
}

// out[i, indices[i, j]] = src[i, j] for j < counts[i], j < w_idx; false = declined
bool scatter_rows(const at::Tensor& src, const at::Tensor& indices, const at::Tensor& counts, int64_t w_idx, const at::Tensor& out)
{
    if (!g_api.scatter || !plain_cuda(src) || !plain_cuda(indices) || !plain_cuda(counts) || !plain_cuda(out)) return false;
    if (src.dim() < 2 || out.dim() < 2 || indices.dim() != 2 || index_code(indices) < 0 || index_code(counts) < 0) return false;
    if (!on_current_device(src) || indices.get_device() != src.get_device() || out.get_device() != src.get_device()) return false;
    if (src.numel() == 0 || w_idx == 0) return true;
    check_status(g_api.scatter(src.data_ptr(), out.data_ptr(), indices.data_ptr(), counts.data_ptr(), src.size(0), w_idx,
                               indices.size(1), out.size(1), row_elems(src, 2) * (int64_t)src.element_size(),
                               index_code(indices), index_code(counts), nullptr, stream_of(src)),
                 "scatter_rows");
    return true;
}

// batched_indexing_access_cuda.forward (reference cpp:54-86): result [*batch, K, *data] = gathered rows, `fill_bits`
// (the filler's little-endian byte pattern in the data type) elsewhere; None = declined
py::object forward_gather_fill(const at::Tensor& data, const at::Tensor& indices, const at::Tensor& counts, uint64_t fill_bits)
{
    if (!g_api.gather_fill || !plain_cuda(data) || !plain_cuda(indices) || !plain_cuda(counts)) return py::none();
    const int64_t nb = counts.dim();
    if (nb < 1 || indices.dim() != nb + 1 || data.dim() < nb + 1 || index_code(indices) < 0 || index_code(counts) < 0)
        return py::none();
    if (!on_current_device(data) || indices.get_device() != data.get_device() || counts.get_device() != data.get_device())
        return py::none();
    const auto st = data.scalar_type();
    if (!(st == at::kFloat || st == at::kDouble || st == at::kHalf || st == at::kBFloat16 || st == at::kInt || st == at::kLong))
        return py::none();
    int64_t batch = 1;
    for (int64_t d = 0; d < nb; ++d) {
        if (data.size(d) != counts.size(d) || indices.size(d) != counts.size(d)) return py::none();
        batch *= counts.size(d);
    }
    std::vector<int64_t> shape(indices.sizes().begin(), indices.sizes().end());
    for (int64_t d = nb + 1; d < data.dim(); ++d) shape.push_back(data.size(d));
    at::Tensor res = at::empty(shape, data.options());
    if (indices.numel() == 0 || res.numel() == 0) return py::none();   // (the python path returns torch.full there)
    const int64_t esz = (int64_t)data.element_size();
    check_status(g_api.gather_fill(data.data_ptr(), res.data_ptr(), indices.data_ptr(), counts.data_ptr(), batch, data.size(nb),
                                   indices.size(nb), indices.size(nb), row_elems(data, nb + 1) * esz, fill_bits, (int)esz,
                                   index_code(indices), index_code(counts), nullptr, stream_of(data)),
                 "forward");
    return py::cast(res);
}

// mask_to_indices (extension of batched_indexing_access_cuda): positions of the True entries of every row of a contiguous 2-D
// bool mask, in order, int64 [B, M] zero-filled behind, plus int64 counts [B].  The kernels take 3-7 us; the python
// formulation of "three allocations + two C-ABI calls" took 12-14 us (8 x 65 536).  None = declined.
py::object mask_to_indices(const at::Tensor& mask, const c10::optional<at::Tensor>& valid)
{
    if (!g_api.mask_to_indices_ws || !plain_cuda(mask) || mask.dim() != 2 || mask.scalar_type() != at::kBool) return py::none();
    if (!on_current_device(mask) || mask.size(0) == 0) return py::none();
    const void* vptr = nullptr;
    int v64 = 0;
    if (valid.has_value() && valid->defined()) {
        if (!plain_cuda(*valid) || valid->get_device() != mask.get_device() || valid->numel() != mask.size(0) || index_code(*valid) < 0)
            return py::none();
        vptr = valid->data_ptr();
        v64 = index_code(*valid);
    }
    const int64_t b = mask.size(0), w = mask.size(1);
    const auto opts = mask.options().dtype(at::kLong);
    at::Tensor idx = at::empty({b, w}, opts), sizes = at::empty({b}, opts), ws;
    const size_t ws_bytes = g_api.mask_to_indices_ws_bytes(b, w);
    if (ws_bytes) ws = at::empty({(int64_t)ws_bytes}, mask.options().dtype(at::kByte));
    check_status(g_api.mask_to_indices_ws(mask.data_ptr(), vptr, v64, b, w, reinterpret_cast<long long*>(idx.data_ptr<int64_t>()), reinterpret_cast<long long*>(sizes.data_ptr<int64_t>()),
                                          ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(mask)),
                 "mask_to_indices");
    return py::make_tuple(idx, sizes);
}

// depth-first flatten of nested list/tuple structures into tensor leaves; false = something else was found.
// The leaves are POINTERS to the tensors inside their python objects (alive for the duration of the call: the caller's
// structure holds them): a C++ copy of a tensor handle moves the TensorImpl's reference count between 1 and 2, and each such
// transition calls into the interpreter to pin / unpin the python object (c10/util/intrusive_ptr.h, "PyObject preservation") —
// ~150 ns per leaf for a copy that nothing needs (csrc_host/mtc_host.cpp, LeafRef: the same finding at 10 000 leaves).
bool flatten(PyObject* obj, std::vector<const at::Tensor*>& out)
{
    if (THPVariable_Check(obj)) {
        out.push_back(&THPVariable_Unpack(obj));
        return true;
    }
    if (PyList_CheckExact(obj)) {
        const Py_ssize_t n = PyList_GET_SIZE(obj);
        for (Py_ssize_t i = 0; i < n; ++i)
            if (!flatten(PyList_GET_ITEM(obj, i), out)) return false;
        return true;
    }
    if (PyTuple_CheckExact(obj)) {
        const Py_ssize_t n = PyTuple_GET_SIZE(obj);
        for (Py_ssize_t i = 0; i < n; ++i)
            if (!flatten(PyTuple_GET_ITEM(obj, i), out)) return false;
        return true;
    }
    return false;
}

// combine_data, flatten mode, CPU target: (padded [B, width, *inner], sizes int64 [B]) or None
py::object pack_cpu(const py::object& data, bool pin, int64_t max_bytes)
{
    std::vector<const at::Tensor*> leaves;
    if (!flatten(data.ptr(), leaves) || leaves.empty()) return py::none();
    const at::Tensor* proto = nullptr;
    int64_t width = 0;
    for (const at::Tensor* tp : leaves) {
        const at::Tensor& t = *tp;
        if (!t.defined() || !t.device().is_cpu() || t.requires_grad() || t.dim() < 1 || t.is_sparse() || t.is_quantized())
            return py::none();
        width = std::max<int64_t>(width, t.size(0));
        if (!proto && t.numel() > 0) proto = tp;
    }
    if (!proto) return py::none();  // nothing but empty samples: rare, python handles it
    const auto inner = proto->sizes().slice(1);
    for (const at::Tensor* tp : leaves) {
        const at::Tensor& t = *tp;
        if (t.numel() == 0) continue;
        if (t.scalar_type() != proto->scalar_type() || t.sizes().slice(1) != inner) return py::none();
    }
    const int64_t b = (int64_t)leaves.size();
    std::vector<int64_t> shape{b, width};
    shape.insert(shape.end(), inner.begin(), inner.end());
    {
        int64_t bytes = (int64_t)proto->element_size();
        for (auto d : shape) bytes *= d;
        if (max_bytes > 0 && bytes > max_bytes) return py::none();  // caller prefers another route for big batches
    }
    // `pin`: the padded batch goes to a GPU next — build it in pinned memory so that ONE asynchronous copy moves it
    at::Tensor padded = at::zeros(shape, proto->options().pinned_memory(pin));
    at::Tensor sizes = at::empty({b}, at::TensorOptions().dtype(at::kLong).pinned_memory(pin));
    int64_t* sz = sizes.data_ptr<int64_t>();
    int64_t row_elems = 1;
    for (auto s : inner) row_elems *= s;
    const size_t row_bytes = (size_t)row_elems * proto->element_size();
    char* base = static_cast<char*>(padded.data_ptr());
    for (int64_t i = 0; i < b; ++i) {
        const at::Tensor& t = *leaves[(size_t)i];
        const int64_t n = t.numel() == 0 ? 0 : t.size(0);
        sz[i] = n;
        if (n == 0) continue;
        char* dst = base + (size_t)i * (size_t)width * row_bytes;
        if (t.is_contiguous()) {
            std::memcpy(dst, t.data_ptr(), (size_t)n * row_bytes);
        } else {
            padded.select(0, i).narrow(0, 0, n).copy_(t);
        }
    }
    return py::make_tuple(padded, sizes);
}

// combine_data, flatten mode, all samples on ONE device (typically the GPU): the python loop that trims / checks the
// samples and the concatenation, in one call.  Returns (flat [total, *inner], sizes int64 [B] on the CPU,
// meta int64 [2, B] = (row offsets, sizes) on the CPU — pinned if `pin` —, width) or None.
py::object cat_leaves(const py::object& data, bool pin)
{
    std::vector<const at::Tensor*> leaves;
    if (!flatten(data.ptr(), leaves) || leaves.empty()) return py::none();
    const at::Tensor* proto = nullptr;
    int64_t width = 0;
    for (const at::Tensor* tp : leaves) {
        const at::Tensor& t = *tp;
        if (!t.defined() || t.requires_grad() || t.dim() < 1 || t.is_sparse() || t.is_quantized()) return py::none();
        width = std::max<int64_t>(width, t.size(0));
        if (!proto && t.numel() > 0) proto = tp;
    }
    if (!proto) return py::none();
    const auto inner = proto->sizes().slice(1);
    const int64_t b = (int64_t)leaves.size();
    at::Tensor meta = at::empty({2, b}, at::TensorOptions().dtype(at::kLong).pinned_memory(pin));
    int64_t* off = meta.data_ptr<int64_t>();
    int64_t* sz = off + b;
    std::vector<std::reference_wrapper<const at::Tensor>> parts;   // (references, not handle copies: see flatten)
    parts.reserve(leaves.size());
    int64_t total = 0;
    for (int64_t i = 0; i < b; ++i) {
        const at::Tensor& t = *leaves[(size_t)i];
        const int64_t n = t.numel() == 0 ? 0 : t.size(0);
        off[i] = total;
        sz[i] = n;
        total += n;
        if (n == 0) continue;
        if (t.device() != proto->device() || t.scalar_type() != proto->scalar_type() || t.sizes().slice(1) != inner)
            return py::none();
        parts.push_back(std::cref(t));
    }
    at::Tensor flat = parts.size() == 1 ? parts[0].get().contiguous() : at::cat(at::ITensorListRef(parts), 0);
    at::Tensor sizes = at::empty({b}, at::TensorOptions().dtype(at::kLong));
    std::memcpy(sizes.data_ptr<int64_t>(), sz, (size_t)b * sizeof(int64_t));
    return py::make_tuple(flat, sizes, meta, width);
}

// RaggedBatch.split for one (flattened) batch dimension: views flat[i].narrow(0, 0, sizes[i]) (+ transpose(0, back))
// RaggedBatch.mask for host data: bool [*sizes.shape, n] with row i = (arange(n) < sizes[i]) — the four torch calls of the python
// formulation cost 12 us for 64 samples, a quarter of configs[0]'s pack + mask + split
at::Tensor mask_cpu(const at::Tensor& sizes, int64_t n)
{
    TORCH_CHECK(sizes.device().is_cpu() && n >= 0, "mask_cpu: host sizes and a non-negative width expected");
    const at::Tensor s = sizes.scalar_type() == at::kLong ? sizes.contiguous() : sizes.to(at::kLong).contiguous();
    std::vector<int64_t> shape(s.sizes().begin(), s.sizes().end());
    shape.push_back(n);
    at::Tensor out = at::empty(shape, at::TensorOptions().dtype(at::kBool));
    const int64_t* sz = s.data_ptr<int64_t>();
    bool* dst = out.data_ptr<bool>();
    const int64_t rows = s.numel();
    for (int64_t i = 0; i < rows; ++i) {
        const int64_t k = std::min<int64_t>(std::max<int64_t>(sz[i], 0), n);
        std::memset(dst + i * n, 1, (size_t)k);
        std::memset(dst + i * n + k, 0, (size_t)(n - k));
    }
    return out;
}

std::vector<at::Tensor> split_views(const at::Tensor& flat, const std::vector<int64_t>& sizes, int64_t back)
{
    TORCH_CHECK(flat.dim() >= 2, "split_views needs [batch, width, ...]");
    TORCH_CHECK((int64_t)sizes.size() == flat.size(0), "one size per sample expected");
    const int64_t width = flat.size(1);
    std::vector<at::Tensor> out;
    out.reserve(sizes.size());
    // tensors outside autograd: build each view's TensorImpl directly on the storage (no dispatcher round trips);
    // tensors that require grad keep the differentiable select/narrow views
    const bool direct = !flat.requires_grad() && !flat.is_sparse() && flat.has_storage() && !flat.is_inference();
    std::vector<int64_t> vsizes(flat.sizes().begin() + 1, flat.sizes().end());
    std::vector<int64_t> vstrides(flat.strides().begin() + 1, flat.strides().end());
    if (direct && back) {
        TORCH_CHECK(back < (int64_t)vsizes.size(), "dimension out of range");
        std::swap(vstrides[0], vstrides[(size_t)back]);
    }
    for (int64_t i = 0; i < (int64_t)sizes.size(); ++i) {
        const int64_t n = sizes[(size_t)i];
        TORCH_CHECK(n >= 0 && n <= width, "sample size ", n, " outside [0, ", width, "]");
        if (direct) {
            auto impl = c10::make_intrusive<at::TensorImpl>(c10::Storage(flat.storage()), flat.key_set(), flat.dtype());
            std::vector<int64_t> sz = vsizes;
            sz[0] = n;                                  // the ragged dimension (dim 0 of the sample)
            if (back) std::swap(sz[0], sz[(size_t)back]);
            impl->set_sizes_and_strides(sz, vstrides);
            impl->set_storage_offset(flat.storage_offset() + i * flat.stride(0));
            // share the base's version counter as a real ATen view does: an in-place write through a sample must be
            // seen by autograd's "modified in place" check of anything that saved the padded tensor
            impl->set_version_counter(flat.unsafeGetTensorImpl()->version_counter());
            out.emplace_back(std::move(impl));
            continue;
        }
        at::Tensor s = flat.select(0, i);
        if (n != width) s = s.narrow(0, 0, n);
        out.push_back(back ? s.transpose(0, back) : s);
    }
    return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "host fast path of accvlab.batching_helpers (per-sample pack / split loops)";
    m.def("pack_cpu", &pack_cpu, py::arg("data"), py::arg("pin") = false, py::arg("max_bytes") = 0);
    m.def("cat_leaves", &cat_leaves, py::arg("data"), py::arg("pin") = false);
    m.def("split_views", &split_views);
    m.def("mask_cpu", &mask_cpu);
    m.def("bind_native", &bind_native);
    m.def("gather_rows", &gather_rows);
    m.def("scatter_rows", &scatter_rows);
    m.def("forward_gather_fill", &forward_gather_fill);
    m.def("bind_mask_to_indices", &bind_mask_to_indices);
    m.def("mask_to_indices", &mask_to_indices, py::arg("mask"), py::arg("valid_counts") = py::none());
}
