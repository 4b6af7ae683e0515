// Is the v_dot2c_f32_bf16 remainder (x - bf16 piece) bit-identical to the unpack-and-subtract form?  (tools check, not product code)
// MODE 1: the packed constants (-1, 0) / (0, -1) come from registers - exact.  MODE 0: literal constants, which the compiler folds into
// the instruction; (-1, 0) becomes the inline operand "-1.0", which the hardware does not read as that bf16 pair - expect mismatches
// (why split_pk in csrc/igemm16.hip hides its constants from the optimiser).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ unsigned pk(float a, float b) { bf2 v; v.x = (__bf16)a; v.y = (__bf16)b; return __builtin_bit_cast(unsigned, v); }
template <int MODE>
__global__ void k(const float* in, unsigned* o_ref, unsigned* o_dot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float a = in[2 * i], b = in[2 * i + 1];
    const unsigned h = pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xFFFF0000u);
    unsigned klo = 0x0000BF80u, khi = 0xBF800000u;
    if (MODE) { asm volatile("" : "+s"(klo)); asm volatile("" : "+s"(khi)); }
    const float da = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, h), __builtin_bit_cast(bf2, klo), a, false);
    const float db = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, h), __builtin_bit_cast(bf2, khi), b, false);
    o_ref[2 * i] = __builtin_bit_cast(unsigned, ra); o_ref[2 * i + 1] = __builtin_bit_cast(unsigned, rb);
    o_dot[2 * i] = __builtin_bit_cast(unsigned, da); o_dot[2 * i + 1] = __builtin_bit_cast(unsigned, db);
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(2 * n * sizeof(float));
    srand(1);
    for (int i = 0; i < 2 * n; ++i) { h[i] = ((float)rand() / RAND_MAX - 0.5f) * powf(2.f, (float)(rand() % 40 - 30)); }
    float* d; unsigned *r, *t; hipMalloc(&d, 2 * n * 4); hipMalloc(&r, 2 * n * 4); hipMalloc(&t, 2 * n * 4);
    hipMemcpy(d, h, 2 * n * 4, hipMemcpyHostToDevice);
    unsigned* hr = (unsigned*)malloc(2 * n * 4); unsigned* ht = (unsigned*)malloc(2 * n * 4);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, d, r, t); else hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, d, r, t);
        hipMemcpy(hr, r, 2 * n * 4, hipMemcpyDeviceToHost); hipMemcpy(ht, t, 2 * n * 4, hipMemcpyDeviceToHost);
        long bad = 0; int first = -1;
        for (int i = 0; i < 2 * n; ++i) if (hr[i] != ht[i]) { if (first < 0) first = i; ++bad; }
        printf("mode %d: %ld of %d remainders differ", mode, bad, 2 * n);
        if (first >= 0) { float x = h[first], a, b; memcpy(&a, &hr[first], 4); memcpy(&b, &ht[first], 4); printf("  first: x=%a ref=%a dot=%a (slot %d)", x, a, b, first & 1); }
        printf("\n");
    }
    return 0;
}
