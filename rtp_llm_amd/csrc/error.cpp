// Thread-local error string + ABI version for libmi355_decode.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <set>
#include <utility>
#include "../../include/mi355_decode.h"

static thread_local char g_err[512] = "";

void mi355_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mi355_last_error(void) { return g_err; }
extern "C" int mi355_abi_version(void) { return MI355_ABI_VERSION; }

// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// (kernel, device): the attribute is per device, and launch functions may be entered from several host threads.
int mi355_raise_dynamic_lds(const void* func, const char* name) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { mi355_set_error("%s: hipGetDevice failed", name); return MI355_ERR_HIP; }
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({func, dev})) return MI355_OK;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        mi355_set_error("%s: cannot raise the dynamic LDS limit: %s", name, hipGetErrorString(e));
        return MI355_ERR_HIP;
    }
    done.insert({func, dev});
    return MI355_OK;
}
