// Go / no-go probe for Winograd F(2x2, 3x3) in the split mode (not product code).
//   hipcc --offload-arch=gfx950 -O3 tools/winograd_probe.hip -o gpurun_out/winograd_probe && gpurun_out/winograd_probe
//
// The 3x3 stride-1 layers run on conv16x3hf_kernel: per 48 MFMAs a wave reads 24 pixel fragments from LDS, loads 12 filter fragments (1 KB
// each) from L2 and - amortised over the taps that reuse a staged patch - spends 28 VALU + 4 LDS stores on splitting fp32 inputs into three
// bf16 planes.  A Winograd kernel with fp32-accurate operands (16 position GEMMs over K = C = 128, all position accumulators resident,
// 64 tiles x 128 kout per workgroup of 8 waves, two passes of 8 positions: DESIGN 7-2) does 2.25x fewer MFMAs per output, but every MFMA
// then carries 4.9x the transform + split work (each input pixel feeds four transformed values instead of one, none of them reused by a
// second tap): per 48 MFMAs 24 LDS fragment reads, 24 filter fragments from L2, ~136 VALU and 12 LDS stores.  This probe runs both
// instruction mixes in one loop structure - 8 waves per workgroup, one workgroup per CU, every wave BOTH multiplies and does its share of the
// staging work, one barrier per 48 MFMAs - and prints the sustained bf16 MFMA rate of each.  Projected speed-up of the main loop =
// 2.25 x rate(winograd mix) / rate(direct mix); the output transform (36 adds per 16 accumulators) and the larger tile (fewer
// workgroups on the 16x16 layers) come on top, on the wrong side.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NLDS, int NGLB, int NVALU, int NST>
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ filt, unsigned filt_items, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 61440 / 2; i += blockDim.x) smem[i] = (unsigned short)(0x3f80 + (i & 7));
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 f[12], gl[12];
    for (int q = 0; q < 12; ++q) { f[q] = u32x4{0x3f803f80u + q, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; gl[q] = f[q]; }
    const int l31 = lane & 31, h = lane >> 5;
    unsigned gpos = (unsigned)(blockIdx.x * 977 + wave * 64 + lane) % filt_items;
    float v0 = 1.f + lane, v1 = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // pixel-side fragments from LDS: NLDS per 48 MFMAs
#pragma unroll
            for (int q = 0; q < NLDS / 2; ++q)
                f[q % 12] = *reinterpret_cast<const u32x4*>(&smem[(q % 3) * 10240 + ((wave & 1) * 64 + ((q / 6) & 1) * 32 + l31) * 40 + ks * 16 + h * 8 + (q & 1) * 5120]);
            // filter-side fragments from L2 (a 1.5 MB image, streamed): NGLB per 48 MFMAs, consumed one half-step later
#pragma unroll
            for (int q = 0; q < NGLB / 2; ++q) {
                gl[q % 12] = filt[gpos];
                gpos += 64; if (gpos >= filt_items) gpos -= filt_items;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(c * 2 + (i >> 1)) % 12]), __builtin_bit_cast(bf16x8, gl[(c * 2 + (i & 1)) % 12]), acc[i], 0, 0, 0);
            // this wave's share of the staging work (transform adds + the three-term split), independent of the accumulators
#pragma unroll
            for (int q = 0; q < NVALU / 2; ++q) { v0 = __builtin_fmaf(v0, 1.0001f, v1); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f); }
            unsigned short* dst = smem + 61440 / 2 + tid * 4;
#pragma unroll
            for (int q = 0; q < NST / 2; ++q) *reinterpret_cast<uint2*>(dst + q * 2048) = make_uint2(__float_as_uint(v0), (unsigned)q);
        }
        __syncthreads();
    }
    float s = v0 + v1;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NLDS, int NGLB, int NVALU, int NST>
double run(const u32x4* filt, unsigned items, int blocks, int iters, float* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NLDS, NGLB, NVALU, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NLDS, NGLB, NVALU, NST>), dim3(blocks), dim3(512), 122880, 0, filt, items, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NLDS, NGLB, NVALU, NST>), dim3(blocks), dim3(512), 122880, 0, filt, items, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 8 * iters * 48 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 512 * sizeof(float));
    const unsigned items = 1536 * 1024 / 16;               // 1.5 MB of 16-byte fragments lanes
    u32x4* filt; hipMalloc(&filt, (size_t)items * 16); hipMemset(filt, 0x3f, (size_t)items * 16);
    const int it = 3000;
    printf("8 waves per workgroup, one workgroup per CU (256 workgroups), one barrier per 48 MFMAs per wave; sustained bf16 32x32x16 TFLOP/s\n");
    const double base = run<0, 0, 0, 0>(filt, items, 256, it, out);
    printf("MFMAs only                                                         %7.1f\n", base);
    const double lds = run<24, 0, 0, 0>(filt, items, 256, it, out);
    printf("+ 24 LDS fragment reads                                            %7.1f\n", lds);
    const double dir = run<24, 12, 28, 4>(filt, items, 256, it, out);
    printf("direct mix   (24 LDS, 12 L2 fragments,  28 VALU,  4 LDS stores)    %7.1f\n", dir);
    const double w1 = run<24, 24, 136, 12>(filt, items, 256, it, out);
    printf("winograd mix (24 LDS, 24 L2 fragments, 136 VALU, 12 LDS stores)    %7.1f\n", w1);
    const double w2 = run<24, 24, 100, 12>(filt, items, 256, it, out);
    printf("winograd mix, optimistic transform (100 VALU)                      %7.1f\n", w2);
    const double w3 = run<24, 24, 200, 12>(filt, items, 256, it, out);
    printf("winograd mix, pessimistic transform (200 VALU)                     %7.1f\n", w3);
    printf("projected main-loop speed-up 2.25 x winograd / direct: %.2f (optimistic %.2f, pessimistic %.2f); keep only above 1.35\n",
           2.25 * w1 / dir, 2.25 * w2 / dir, 2.25 * w3 / dir);
    return 0;
}
