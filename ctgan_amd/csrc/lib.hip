// lib.hip - library-level entry points: version, thread-local error state, launch checking.
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

int ctgan_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int ctgan_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ctgan_fail(CTGAN_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return CTGAN_OK;
}

extern "C" {
int ctgan_version(void) { return CTGAN_ABI_VERSION; }
const char* ctgan_last_error(void) { return g_err; }
}
