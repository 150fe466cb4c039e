// _fastcall — a thin CPython-C-API trampoline into the C-ABI of libaccv_hip.so (include/accv_hip.h).
//
// ctypes spends ~4 us per call converting a dozen arguments (profiles/r01_host_overhead_before_after.log); the kernels
// behind the ragged operators run for 2-5 us, so the binding cost IS the operator cost at the sizes of this path.
// This module calls the same exported functions through their addresses (taken from the ctypes handle, so there is one
// copy of the library and one symbol table) with METH_FASTCALL argument passing: ~0.3 us per call.
//
//   call_ints(fn, a0, a1, ...)            -> int    every argument is an integer / pointer class value
//   call_f2(fn, f0, f1, a0, a1, ...)      -> int    two leading float arguments + integer class values
//
// Accepted argument objects: int (any sign, passed modulo 2^64), None (-> 0), bool, torch.Tensor (-> data_ptr()),
// objects with the buffer protocol such as ctypes / numpy arrays (-> address of their memory).
// x86-64 System V only (the ABI of the ROCm image): integer-class arguments travel in rdi..r9 and then in 8-byte stack
// slots in declaration order whatever their C width, float arguments in xmm0.. independently of their position, and the
// caller pops the stack — so one 20-slot signature serves every entry point of the header with up to 20 integer-class
// and 2 float parameters (extra slots are ignored by the callee).  Plumbing only: no device code, no HIP calls.
#include <torch/extension.h>

#include <cstdint>

#if !defined(__x86_64__) || defined(_WIN32)
#error "_fastcall relies on the x86-64 System V calling convention"
#endif

namespace {

constexpr Py_ssize_t kMaxInts = 20;

bool convert(PyObject* const* args, Py_ssize_t n, uint64_t (&a)[kMaxInts])
{
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* o = args[i];
        if (o == Py_None) {
            a[i] = 0;
        } else if (PyLong_Check(o)) {  // bool included
            const unsigned long long v = PyLong_AsUnsignedLongLongMask(o);
            if (v == (unsigned long long)-1 && PyErr_Occurred()) return false;
            a[i] = v;
        } else if (THPVariable_Check(o)) {
            a[i] = reinterpret_cast<uint64_t>(THPVariable_Unpack(o).data_ptr());
        } else if (PyObject_CheckBuffer(o)) {  // ctypes arrays, numpy arrays: the address of their memory
            Py_buffer view;
            if (PyObject_GetBuffer(o, &view, PyBUF_SIMPLE) != 0) return false;
            a[i] = reinterpret_cast<uint64_t>(view.buf);
            PyBuffer_Release(&view);  // the caller's reference keeps the memory alive for the duration of the call
        } else if (PyIndex_Check(o)) {  // numpy integers and friends
            PyObject* idx = PyNumber_Index(o);
            if (!idx) return false;
            const unsigned long long v = PyLong_AsUnsignedLongLongMask(idx);
            Py_DECREF(idx);
            if (v == (unsigned long long)-1 && PyErr_Occurred()) return false;
            a[i] = v;
        } else {
            PyErr_Format(PyExc_TypeError, "fastcall: argument %zd must be int, None or a torch.Tensor, got %s", i,
                         Py_TYPE(o)->tp_name);
            return false;
        }
    }
    for (Py_ssize_t i = n; i < kMaxInts; ++i) a[i] = 0;
    return true;
}

using IntFn = int (*)(uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t,
                      uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);
using F2Fn = int (*)(float, float, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t,
                     uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t,
                     uint64_t);

PyObject* call_ints(PyObject*, PyObject* const* args, Py_ssize_t nargs)
{
    if (nargs < 1 || nargs > kMaxInts + 1) {
        PyErr_SetString(PyExc_TypeError, "call_ints(fn, up to 20 integer-class arguments)");
        return nullptr;
    }
    const unsigned long long fn = PyLong_AsUnsignedLongLong(args[0]);
    if (fn == (unsigned long long)-1 && PyErr_Occurred()) return nullptr;
    uint64_t a[kMaxInts];
    if (!convert(args + 1, nargs - 1, a)) return nullptr;
    const int rc = reinterpret_cast<IntFn>(fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12],
                                               a[13], a[14], a[15], a[16], a[17], a[18], a[19]);
    return PyLong_FromLong(rc);
}

PyObject* call_f2(PyObject*, PyObject* const* args, Py_ssize_t nargs)
{
    if (nargs < 3 || nargs > kMaxInts + 3) {
        PyErr_SetString(PyExc_TypeError, "call_f2(fn, f0, f1, up to 20 integer-class arguments)");
        return nullptr;
    }
    const unsigned long long fn = PyLong_AsUnsignedLongLong(args[0]);
    if (fn == (unsigned long long)-1 && PyErr_Occurred()) return nullptr;
    const double f0 = PyFloat_AsDouble(args[1]), f1 = PyFloat_AsDouble(args[2]);
    if ((f0 == -1.0 || f1 == -1.0) && PyErr_Occurred()) return nullptr;
    uint64_t a[kMaxInts];
    if (!convert(args + 3, nargs - 3, a)) return nullptr;
    const int rc = reinterpret_cast<F2Fn>(fn)((float)f0, (float)f1, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9],
                                              a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18], a[19]);
    return PyLong_FromLong(rc);
}

PyMethodDef kMethods[] = {
    {"call_ints", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(call_ints)), METH_FASTCALL,
     "call_ints(fn_address, *integer_class_args) -> status"},
    {"call_f2", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(call_f2)), METH_FASTCALL,
     "call_f2(fn_address, f0, f1, *integer_class_args) -> status"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef kModule = {PyModuleDef_HEAD_INIT, "_fastcall", "fast trampoline into the C-ABI of libaccv_hip.so", -1, kMethods,
                       nullptr, nullptr, nullptr, nullptr};

}  // namespace

extern "C" __attribute__((visibility("default"))) PyObject* PyInit__fastcall(void) { return PyModule_Create(&kModule); }
