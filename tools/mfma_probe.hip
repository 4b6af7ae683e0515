// mfma_probe.hip - what limits a 2x2-accumulator fp32 MFMA loop on gfx950?  Standalone probe (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Variants (template V):
//   0 pure MFMA, operands constant registers                (expected ~155 TF)
//   1 + B operand from LDS (ds_read_b32 per MFMA pair), A from registers loaded per 16 steps (ds_read_b128)
//   2 + s_barrier per 64 MFMAs
//   3 + 8 ds_write_b128 per 64 MFMAs
//   4 + 8 global float4 loads per 64 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (128 * 36 + 32 * 128)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < 2 * (128 * 36 + 32 * 128); i += 256) lds[i] = 0.001f * (i & 15);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float a0 = 1.0f + lane, a1 = 0.5f, b0 = 0.25f, b1 = 2.f;
    const float* As = lds;
    const float* Bs = lds + 128 * 36;
    const int a_rd = ((wave >> 1) * 64 + l31) * 36 + h * 16;
    const int b_rd = (h * 16) * 128 + (wave & 1) * 64 + l31;
    float4 ra[8];
    for (int i = 0; i < 8; ++i) ra[i] = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const int st = (it & 1) * (128 * 36 + 32 * 128);
        if (V == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else {
            float4 a[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) a[i][v] = *reinterpret_cast<const float4*>(&As[st + a_rd + i * 32 * 36 + v * 4]);
            float b[2][2];
            b[0][0] = Bs[st + b_rd]; b[0][1] = Bs[st + b_rd + 32];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) { b[(s + 1) & 1][0] = Bs[st + b_rd + (s + 1) * 128]; b[(s + 1) & 1][1] = Bs[st + b_rd + (s + 1) * 128 + 32]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 av = a[i][s >> 2];
                    const float ae = (s & 3) == 0 ? av.x : (s & 3) == 1 ? av.y : (s & 3) == 2 ? av.z : av.w;
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b[s & 1][0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b[s & 1][1], acc[i][1], 0, 0, 0);
                }
                if (V >= 3 && s == 0) {
                    float* dst = lds + ((it + 1) & 1) * (128 * 36 + 32 * 128);
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&dst[((tid >> 3) + i * 32) * 36 + (tid & 7) * 4]) = ra[i];
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&dst[128 * 36 + ((tid >> 5) + i * 8) * 128 + (tid & 31) * 4]) = ra[4 + i];
                }
                if (V >= 4 && s == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const float4*>(g + (((size_t)blockIdx.x * 64 + it * 8 + i) * 256 + tid) * 4 % (1 << 24));
                }
            }
            if (V >= 2) __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    for (int i = 0; i < 8; ++i) s += ra[i].x;
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char* name, int blocks, const float* g, float* out) {
    const int iters = 36 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256>>>(g, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) probe<V><<<blocks, 256>>>(g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-44s blocks %5d  %8.1f us  %7.1f TFLOP/s\n", name, blocks, ms * 1e3, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *g, *out; hipMalloc(&g, (1 << 24) * 4 + 4096); hipMalloc(&out, 4096 * 256 * 4);
    hipMemset(g, 0, (1 << 24) * 4 + 4096);
    for (int blocks : {256, 512, 1024}) {
        run<0>("0 pure MFMA (const operands)", blocks, g, out);
        run<1>("1 + LDS operand reads", blocks, g, out);
        run<2>("2 + barrier / 64 MFMA", blocks, g, out);
        run<3>("3 + 8 ds_write_b128 / 64 MFMA", blocks, g, out);
        run<4>("4 + 8 global float4 loads / 64 MFMA", blocks, g, out);
    }
    return 0;
}
