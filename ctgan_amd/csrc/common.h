// common.h - shared host-side helpers of libctgan_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ctgan_hip.h"
#include "../../include/ctgan_hip_debug.h"

// thread-local last-error message (no exception crosses the C ABI)
int ctgan_fail(int code, const char* fmt, ...);
// hipGetLastError() after a launch -> CTGAN_E_LAUNCH with the HIP error string
int ctgan_check_launch(const char* what);
// split-K plan of the weight-gradient GEMM (shared by the launcher and the workspace query)
void ctgan_wgrad_split(int tiles, int Kg, int* splits, int* chunk);

static inline unsigned ctgan_blocks(long long n, int per_block, int cap = 4096) {
    long long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// skinny.hip: small-N linear layers (critic heads)
bool ctgan_is_small_linear(const ctgan_conv_desc* d);
int ctgan_small_linear_fwd(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu,
                           hipStream_t st);
int ctgan_small_linear_dgrad(const ctgan_conv_desc* d, const float* gy, const float* w, const float* bias, float* gx,
                             hipStream_t st);
int ctgan_small_linear_wgrad(const ctgan_conv_desc* d, const float* x, const float* gy, float* gw, float* gb, hipStream_t st);
// per-thread name of the kernel variant the last conv call dispatched to
void ctgan_set_last_kernel(const char* name);
// ... and the device symbol of that launch as rocprofv3 prints it (template arguments spelled the compiler's way), so that
// bench.py's per-kernel table can be looked up in a committed rocprof summary.  ctgan_set_last_kernel clears it.
void ctgan_set_last_symbol(const char* fmt, ...);

// fewch.hip: direct kernels for convs with <= 4 channels on one side.  fwd / dgrad / wgrad return 1 when they
// handled the call, 0 when the caller should fall through to the GEMM kernels, < 0 on error.
bool ctgan_fewch_handles(const ctgan_conv_desc* d);
int ctgan_fewch_fwd(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, const float* mask, const float* resid,
                    float* y, int relu, int relu_in, hipStream_t st);
int ctgan_fewch_fwd_bn(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu_in, const float* mean,
                       const float* rstd, const float* scale, const float* offset, int groups, int tanh_out, hipStream_t st);
int ctgan_fewch_dgrad(const ctgan_conv_desc* d, const float* dy, const float* w, const float* bias, float* dx, hipStream_t st);
size_t ctgan_fewch_wgrad_workspace(const ctgan_conv_desc* d);
int ctgan_fewch_wgrad(const ctgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                      int relu_x, hipStream_t st);
int ctgan_fewch_wgrad2(const ctgan_conv_desc* d, const float* x, const float* dy, int N0, int relu_x, int bias0, const float* x1,
                       const float* dy1, int N1, int relu_x1, int bias1, float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t st);
