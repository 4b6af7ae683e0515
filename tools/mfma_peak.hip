// Sustained MFMA issue rate of the chip at its real clocks (calibration for the roofline discussion in DESIGN.md):
// every wave runs a dependency-free chain of MFMAs on 4 accumulators, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void peak_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
    const float fa = (float)threadIdx.x, fb = (float)(threadIdx.x + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
double run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(peak_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(peak_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_mfma = KIND == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    return (double)blocks * 4 * iters * 16 * flop_per_mfma / (ms * 1e-3) / 1e12;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    for (int blocks : {256, 512, 1024}) {
        for (int iters : {2000, 20000}) {
            printf("blocks %4d iters %5d : bf16 32x32x16 %8.1f TFLOP/s   f32 32x32x2 %7.1f TFLOP/s\n", blocks, iters, run<0>(blocks, iters, out), run<1>(blocks, iters, out));
        }
    }
    return 0;
}
