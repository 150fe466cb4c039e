// Shared host-side helpers for libaccv_hip.so (error reporting, launch checks). gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "accv_hip.h"

namespace accv {

char* error_buffer();  // thread-local, 512 bytes
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Checks the sticky launch error after a kernel launch / async enqueue.
inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ACCV_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ACCV_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// runtime tuning knobs (bench / A-B experiments only; defaults are the shipped configuration)
int tune_get(const char* key, int fallback);

}  // namespace accv
