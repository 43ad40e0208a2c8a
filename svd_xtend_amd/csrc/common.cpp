// common.cpp -- error reporting and library identity for libsvdx.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/svdx.h"

static thread_local char g_err[512] = "";

void svdx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int svdx_version(void) { return SVDX_VERSION; }

extern "C" int svdx_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return -1;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return 0;
}

extern "C" int svdx_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        svdx_set_error("no HIP device");
        return 0;
    }
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        svdx_set_error("hipGetDeviceProperties failed");
        return 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        svdx_set_error("device arch %s is not gfx950 (libsvdx is MI355X-only)", prop.gcnArchName);
        return 0;
    }
    return 1;
}
