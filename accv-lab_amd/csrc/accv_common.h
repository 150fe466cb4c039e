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

char* dispatch_buffer();  // thread-local, 256 bytes: description of the last draw_heatmap dispatch

#ifdef ACCV_TUNE_BUILD
// runtime tuning knobs: ONLY in the A/B build (make tune -> libaccv_hip_tune.so, scripts/h1_variants.py --alt-lib);
// the shipped library has no knob table, no mutex and no string look-ups on its dispatch path
int tune_get(const char* key, int fallback);
#else
inline int tune_get(const char*, int fallback) { return fallback; }
#endif

}  // namespace accv
