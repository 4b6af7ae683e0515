// mma16.h - the 16-bit matrix-core helpers shared by igemm16.hip and wgrad16c.hip: operand conversion (bf16 / fp16 rounding, or the
// exact three-term bf16 split of an fp32 value) and the v_mfma_f32_32x32x16 wrappers.  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ctgan_hip.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

template <int MMA> struct Cvt;
template <> struct Cvt<CTGAN_MMA_BF16> {
    static __device__ __forceinline__ unsigned pk(float a, float b) {
        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
        v2 v; v.x = (__bf16)a; v.y = (__bf16)b;
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
// fp32 values as three bf16 terms (x = h + m + l, each the nearest-even bf16 of what the previous ones left: 24 significand bits):
// the product x*w = hh + hm + mh + mm + hl + lh (+ terms below 2^-24 relative that are dropped) - six bf16 MFMAs that accumulate in
// fp32 reproduce an fp32 multiply-accumulate to fp32 rounding accuracy at 6/16 of the fp32 MFMA's cycle cost.
template <> struct Cvt<CTGAN_MMA_F32X3> : Cvt<CTGAN_MMA_BF16> {};
template <> struct Cvt<CTGAN_MMA_F16> {
    static __device__ __forceinline__ unsigned pk(float a, float b) {
        typedef _Float16 v2 __attribute__((ext_vector_type(2)));
        v2 v; v.x = (_Float16)a; v.y = (_Float16)b;
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

template <int MMA> constexpr int planes() { return MMA == CTGAN_MMA_F32X3 ? 3 : 1; }
// the 16-bit pieces of the pair (a, b), packed (a low, b high): one rounded piece, or the three terms of the split
template <int MMA>
__device__ __forceinline__ void split_pk(float a, float b, unsigned (&o)[planes<MMA>()]) {
    if constexpr (planes<MMA>() == 1) {
        o[0] = Cvt<MMA>::pk(a, b);
    } else {
        // remainder of an element = x - piece: v_dot2c_f32_bf16 with the packed constants (-1, 0) / (0, -1) subtracts the low / high
        // piece of the pair in ONE instruction (no unpacking); the difference is exactly representable, so the result is exact
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        // (the constants are made opaque: folded into the instruction, (-1, 0) becomes the inline operand "-1.0", which the hardware
        // does not read as that bf16 pair - tools/dot2_check.hip)
        unsigned klo = 0x0000BF80u, khi = 0xBF800000u;
        asm("" : "+s"(klo));
        asm("" : "+s"(khi));
        const bf2 lo = __builtin_bit_cast(bf2, klo), hi = __builtin_bit_cast(bf2, khi);
#ifdef CTGAN_SPLIT_NONE
        // diagnosis build (tools/nosplit_probe.sh; results are wrong by design): the three terms cost ONE pack - what a consumer would pay if
        // its producer had stored the split planes (an upper bound of that change's gain, before the extra operand bytes)
        o[0] = o[1] = o[2] = Cvt<MMA>::pk(a, b);
        return;
#endif
#ifdef CTGAN_SPLIT_SUB
        const unsigned h0 = Cvt<MMA>::pk(a, b);
        const float ra0 = a - __builtin_bit_cast(float, h0 << 16), rb0 = b - __builtin_bit_cast(float, h0 & 0xFFFF0000u);
        const unsigned m0 = Cvt<MMA>::pk(ra0, rb0);
        const float sa0 = ra0 - __builtin_bit_cast(float, m0 << 16), sb0 = rb0 - __builtin_bit_cast(float, m0 & 0xFFFF0000u);
        o[0] = h0; o[1] = m0; o[2] = Cvt<MMA>::pk(sa0, sb0);
        return;
#endif
        const unsigned h = Cvt<MMA>::pk(a, b);
        const float ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, h), lo, a, false);
        const float rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, h), hi, b, false);
        const unsigned m = Cvt<MMA>::pk(ra, rb);
        const float sa = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, m), lo, ra, false);
        const float sb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, m), hi, rb, false);
        o[0] = h; o[1] = m; o[2] = Cvt<MMA>::pk(sa, sb);
    }
}

}  // namespace
