/* ctgan_hip_debug.h - test / A-B switches and launch introspection of libctgan_hip.so.
 *
 * NOT part of the drop-in ABI (include/ctgan_hip.h): nothing a caller of the operator library needs, no stability promise, process-wide
 * state.  The GPU tests use them to pin a kernel variant (so that both routes of a planner decision are compared with the oracle) and the
 * tools under tools/ to time one variant against another.  Exported from the same shared object; bound by ctgan_amd/_lib.py next to the
 * product table (tests/test_abi_and_layout.py keeps header, exports and ctypes table in step for both headers).
 */
#ifndef CTGAN_HIP_DEBUG_H
#define CTGAN_HIP_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tests only: 1 = route every conv through the table-driven generic kernels                   */
void ctgan_debug_force_generic(int on);
/* tests only: 0 = the split-K reduction of the fp32 weight gradients as one thread per float4 everywhere (the form before the four-lanes-per-float4
   kernel for small outputs with many slabs; both give the same bits) */
void ctgan_debug_reduce_lanes(int on);
/* tests only: which halo-patch kernel of the split mode takes the launches that qualify - 1: filter through an LDS stage
   (conv16x3h_kernel), 2: filter fragments streamed from L2 (conv16x3hf_kernel), 0: back to the default (2) */
void ctgan_debug_x3_halo_version(int version);
/* tests / A-B: 0 = the stride-2 data gradients of the split mode on the slice kernel instead of the four-phase halo kernel (conv16x3p_kernel) */
void ctgan_debug_x3_s2halo(int on);
/* tests / A-B: 0 = the stride-2 forward launches of the split mode on the slice kernel instead of conv16x3sf_kernel (filter fragments from L2);
   2 = on that kernel but without its K split (launches of 128 .. 383 tiles of 64 positions) */
void ctgan_debug_x3_s2fwd(int on);
/* tests / A-B: which four-phase data gradients of the folded 4x4 / stride-2 filters run on conv16x3sf_kernel (one phase per workgroup, slice staging)
   instead of conv16x3p_kernel (four phases from one dy patch): 0 (default) every launch of >= 768 workgroups, -1 none */
void ctgan_debug_x3_s2dgrad_sf(int on);
/* Tests / A-B: 0 = the 3x3 many -> few convs (generator output conv, data gradient of the first critic conv) on the row-ring kernel
 * instead of the one-pixel-per-lane kernel with the filter as scalar operands (csrc/fewch.hip, round 5).                          */
void ctgan_debug_m2f_px(int on);
/* tests only: which weight-gradient kernels the last ctgan_conv2d16_wgrad_group call on this thread launched - bit 0: the filter-column
   kernel (wgrad16c_group_kernel), bit 1: the slice kernel (wgrad16_group_kernel)                                                     */
int ctgan_debug_last_wgrad_group_kinds(void);
/* ... and which members (bit i = groups[i]) rode the filter-column kernel                                                             */
unsigned ctgan_debug_last_wgrad_group_col_mask(void);
/* tests / A-B: which stride-1 halo-patch launches of the split mode run on conv16x3hk_kernel (one channel chunk per wave, 64-pixel x 64-kout tiles,
   round 6) instead of the pixel-tiled kernels (or, at 64 rows, the fp32 pipe): mode 0 none, 1 (default) launches of at most max_wgs workgroups (default
   768; max_wgs <= 0 keeps the current value), 2 every launch that qualifies */
void ctgan_debug_x3_hk(int mode, int max_wgs);
/* bench.py's roofline leg: a one-wave kernel that reads s_memrealtime (constant 100 MHz) and s_memtime (shader cycles) when it starts, polls
   `*flag` (device int32; may be NULL) and reads both again when the flag is non-zero or after max_real_ticks (100 MHz ticks, <= 2 s): launched on a
   SIDE stream next to a measured launch, out[0..3] = real0, shader0, real1, shader1 give the shader clock sustained over that launch
   (d shader / d real x 100 MHz); out[4] = 1 when the flag ended the wait.  `out`: five uint64 in device memory. */
int ctgan_debug_clock_probe(const int* flag, uint64_t max_real_ticks, uint64_t* out, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* CTGAN_HIP_DEBUG_H */
