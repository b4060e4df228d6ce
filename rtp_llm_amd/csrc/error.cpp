// Thread-local error string + ABI version for libmi355_decode.so.
#include <stdarg.h>
#include <stdio.h>
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
