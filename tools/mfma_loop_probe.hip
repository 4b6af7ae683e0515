// What keeps a fragment-read + MFMA loop below the MFMA issue rate?  Variants of one consumer-style loop, 4 waves per workgroup,
// one workgroup per CU (probe for the split-mode conv kernels; not product code).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: fragments come from LDS (24 ds_read_b128 per 48 MFMAs), bit 1: s_barrier per 48 MFMAs, bit 2: 8 waves (4 idle at the barrier)
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const bool idle = threadIdx.x >= 256;
    for (int i = threadIdx.x; i < 61440 / 2; i += blockDim.x) smem[i] = (unsigned short)(0x3f80 + (i & 7));
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 f[12];
    for (int q = 0; q < 12; ++q) f[q] = u32x4{0x3f803f80u + q, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const int l31 = lane & 31, h = lane >> 5;
    for (int it = 0; it < iters; ++it) {
        if (!idle) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (MODE & 1) {
#pragma unroll
                    for (int q = 0; q < 12; ++q)
                        f[q] = *reinterpret_cast<const u32x4*>(&smem[(q % 3) * 10240 + ((wave & 1) * 64 + (q / 6) * 32 + l31) * 40 + ks * 16 + h * 8 + (q & 1) * 5120]);
                }
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(c * 2 + (i >> 1)) % 12]), __builtin_bit_cast(bf16x8, f[(c * 2 + 6 + (i & 1)) % 12]), acc[i], 0, 0, 0);
            }
        }
        if (idle) {                                        // partner-wave work per 48 MFMAs of the consumer on the same SIMD
            if (MODE & 8) {                                // 100 dependent-free VALU
#pragma unroll
                for (int q = 0; q < 100; ++q) acc[q & 3][q & 15] = __builtin_fmaf(acc[q & 3][q & 15], 1.0001f, 0.5f);
            }
            if (MODE & 16) {                               // 12 ds_write_b64 + 6 ds_write_b128 into the upper half of the LDS
                unsigned short* dst = smem + 61440 / 2 + tid * 8;
#pragma unroll
                for (int q = 0; q < 12; ++q) *reinterpret_cast<uint2*>(dst + q * 2048 + (tid & 1) * 4) = make_uint2(it, q);
#pragma unroll
                for (int q = 0; q < 6; ++q) *reinterpret_cast<uint4*>(dst + q * 2048) = make_uint4(it, q, 1, 2);
            }
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
double run(int blocks, int iters, float* out) {
    const int threads = (MODE & 4) ? 512 : 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 122880, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 122880, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 4 * iters * 48 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 512 * sizeof(float));
    const int it = 4000;
    printf("one workgroup per CU (256 workgroups), bf16 32x32x16 TFLOP/s (x 1/6 = fp32-equivalent of the split mode)\n");
    printf("registers only                       %7.1f\n", run<0>(256, it, out));
    printf("LDS fragments                        %7.1f\n", run<1>(256, it, out));
    printf("barrier per 48 MFMAs                 %7.1f\n", run<2>(256, it, out));
    printf("LDS fragments + barrier              %7.1f\n", run<3>(256, it, out));
    printf("LDS fragments + barrier + 4 idle     %7.1f\n", run<7>(256, it, out));
    printf("  + partner: 100 VALU                %7.1f\n", run<7 | 8>(256, it, out));
    printf("  + partner: 18 LDS stores           %7.1f\n", run<7 | 16>(256, it, out));
    printf("  + partner: both                    %7.1f\n", run<7 | 24>(256, it, out));
    printf("registers only, 1536 workgroups      %7.1f\n", run<0>(1536, it / 6, out));
    printf("LDS + barrier, 1536 workgroups       %7.1f\n", run<3>(1536, it / 6, out));
    return 0;
}
