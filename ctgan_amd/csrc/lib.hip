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

// ---- shader-clock probe (bench.py roofline leg; include/ctgan_hip_debug.h) ------------------------------------------------------------
// One wave that brackets a region with the two SQ time bases: s_memrealtime (constant 100 MHz) and s_memtime (one tick per shader
// cycle, MI355X_MICROARCH.md): d(memtime) / d(memrealtime) x 100 MHz = the shader clock the chip sustained over the region.  The wave
// sleeps between polls of `*flag` (set by the measured stream after the region) and gives up after `max_real_ticks` - it can never
// outlive that, whatever the queues do.  out[0..3] = real0, shader0, real1, shader1; out[4] = 1 when the flag ended the wait.
__global__ void clock_probe_kernel(const volatile int* flag, unsigned long long max_real_ticks, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    int seen = 0;
    while (r1 - r0 < max_real_ticks) {
        if (flag != nullptr && *flag != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(64);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
    out[0] = r0; out[1] = s0; out[2] = r1; out[3] = s1; out[4] = (unsigned long long)seen;
}

extern "C" int ctgan_debug_clock_probe(const int* flag, uint64_t max_real_ticks, uint64_t* out, void* stream) {
    if (out == nullptr) return ctgan_fail(CTGAN_E_BADARG, "clock_probe: out == NULL");
    if (max_real_ticks == 0 || max_real_ticks > 200000000ull) return ctgan_fail(CTGAN_E_BADARG, "clock_probe: max_real_ticks must be in (0, 2 s]");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (const volatile int*)flag,
                       (unsigned long long)max_real_ticks, (unsigned long long*)out);
    return ctgan_check_launch("clock_probe");
}
