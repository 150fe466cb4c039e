// Error reporting, version and the last-dispatch string of libaccv_hip.so (+ the knob table of the A/B build).
#include "accv_common.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace accv {

char* error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

char* dispatch_buffer()
{
    static thread_local char buf[256] = {0};
    return buf;
}

#ifdef ACCV_TUNE_BUILD
static std::mutex g_tune_mutex;
static std::map<std::string, int>& tune_map()
{
    static std::map<std::string, int> m;
    return m;
}

int tune_get(const char* key, int fallback)
{
    std::lock_guard<std::mutex> lock(g_tune_mutex);
    auto it = tune_map().find(key);
    return it == tune_map().end() ? fallback : it->second;
}
#endif

}  // namespace accv

extern "C" {

const char* accv_last_error(void) { return accv::error_buffer(); }

int accv_version(void) { return 100; }

const char* accv_draw_heatmap_last_dispatch(void) { return accv::dispatch_buffer(); }

#ifdef ACCV_TUNE_BUILD
// A/B build only (not in the public header, not in the shipped library): in-process selection of kernel variants.
int accv_tune_set(const char* key, int value)
{
    if (!key) return ACCV_EINVAL;
    std::lock_guard<std::mutex> lock(accv::g_tune_mutex);
    accv::tune_map()[key] = value;
    return ACCV_OK;
}
#endif
}
