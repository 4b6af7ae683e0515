// philox.h - Philox4x32-10 counter-based generator shared by the RNG kernels and the conv epilogues.
#pragma once
#include <stdint.h>

namespace ctgan_philox {
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// counter layout: { element-block index, stream id, step lo, step hi }, key = seed
__device__ __forceinline__ void draw4(uint64_t seed, uint32_t sid, uint64_t step, uint32_t blk, uint32_t (&c)[4]) {
    c[0] = blk; c[1] = sid; c[2] = (uint32_t)step; c[3] = (uint32_t)(step >> 32);
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1), 24 bits
}  // namespace ctgan_philox
