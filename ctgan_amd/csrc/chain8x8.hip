// chain8x8.hip - vertical fusion of the critic's 8x8 residual blocks in the split mode (round 6).
//
// Blocks 3 and 4 of the ResNet critic (TF/CT_gan_cifar_resnet.py:109-141 ResidualBlock, :174-178 the two blocks behind the dropouts) work on
// 8x8 x 128-channel images.  Each of their 3x3 convs is a launch of at most 384 images - 16-24 K pixels - which the pixel-tiled kernels
// run at 0.33-0.41 of the split mode's peak: a launch is mostly ramp, operand staging and the filter stream out of L2.  But
// a whole 8x8 x 128 image fits in LDS, and the chain conv -> mask -> conv -> (+ residual, dropout) of a block - forward, data gradient
// and the forward-mode pass of the gradient penalty's double backward (tf.gradients inside the loss, :284) alike - couples no two
// images.  So ONE workgroup takes ONE image through up to FOUR convs:
//
//   value = x[image]                                            (64 pixels x 128 channels, fp32, dense channels-last)
//   for step in pre, layer 1..4:
//       layer:  value = conv3x3_same(value, W_step)             six bf16 MFMAs per fp32 product (split mode), fp32 accumulate
//               value = mask_step > 0 ? value : 0               ReLU mask of a forward tensor (constants of the backward passes)
//       value += slot[resid_step]                               residual: a value saved earlier in the chain
//       value *= floor(keep + u) / keep                         tf.nn.dropout's mask, redrawn from the forward's Philox stream (:173-177)
//       slot[save_step] = value
//       value = post_mask_step > 0 ? value : 0
//       out_step[image] = value                                 every intermediate the weight gradients read goes to HBM once
//
// which is the merged backward of blocks 4-3 (critic_schedule.py phase B: two data gradients per block, mask / residual / ranged dropout
// epilogues) with dgrad filter images, and the penalty's double backward through blocks 3-4 (phase C) with forward images.
//
// Workgroup = 8 waves = 4 channel chunks x 2 kout halves: wave (c, k2) multiplies the 64 pixels by kout 64*k2 .. 64*k2+63 over input
// channels 32c .. 32c+31 - conv16x3hk_kernel's inner loop (2 x 2 accumulators, 9 taps x 2 k-steps, filter fragments of its (2 kout blocks,
// chunk c) streamed from L2 one tap ahead, same FRAG image as conv16x3hf / hk).  LDS: the image's halo patch as three bf16 planes per
// chunk (96 KB); after the last MFMA the same memory takes the 4 x 64 x 128 fp32 partial sums, every thread adds the four chunks of its four
// (pixel, channel quad) items in the fixed order 0..3, applies the step's epilogue, stores the result and - after a barrier - writes it
// back as the NEXT layer's patch (split into its three bf16 terms; the halo is zeroed again because the partial sums used that memory).
// Residual slots and dropout draws are thread-local: the (pixel, channel quad) -> thread map is the same in every layer.
// Deterministic: fixed summation order, no atomics.  One image per workgroup: 64-384 workgroups of 512 threads.
//
// MEASURED (round 6; profiles/r06_chain_probe.txt, a diagnosis build with parts of the conv phase compiled out): NOT faster than the
// launches it replaces - 115-120 us per four-conv chain at 256 AND at 64 images against 4 x 32 us (256 rows) / 82 us (64 rows) - and therefore
// OFF by default (kernels.CHAIN8X8, CTGAN_CHAIN8X8=1).  A layer costs ~29 us per workgroup: ~5 us of epilogue + restaging, and a conv phase of
// ~23 us of which 14.5 us remain with the MFMAs compiled out, with the filter stream re-reading one tap (L1 hits) and without the LDS fragment
// reads: the 864 KB of filter fragments a workgroup pulls per layer arrive at ~27 B per clock and CU through the vector-memory path, whatever
// their source.  One image per workgroup means 256 B of filter per MFMA (conv16x3hf at 128-pixel tiles: 128 B; at 32-pixel tiles: 512 B and the
// same bound); only more pixels per workgroup change that, and two images' patches do not fit LDS beside the partial sums.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "mma16.h"
#include "philox.h"

namespace {

constexpr int CH_MMA = CTGAN_MMA_F32X3, CH_NP = 3, CH_BK = 32, CH_LDS_K = CH_BK + 8;
constexpr int CH_HW = 8, CH_PX = 64, CH_C = 128, CH_PW = CH_HW + 2, CH_NPX = CH_PW * CH_PW;      // 10 x 10 halo patch
constexpr int CH_PPLANE = CH_NPX * CH_LDS_K;                 // 16-bit elements per (chunk, plane)
constexpr int CH_CHUNK = CH_NP * CH_PPLANE;                  // ... per chunk region
constexpr int CH_LDE = CH_C + 4;                             // floats per pixel row of a chunk's partial sums
constexpr int CH_PART = CH_PX * CH_LDE;                      // floats per chunk of partial sums
constexpr size_t CH_LDS_BYTES = (size_t)4 * CH_PART * 4 > (size_t)4 * CH_CHUNK * 2 ? (size_t)4 * CH_PART * 4 : (size_t)4 * CH_CHUNK * 2;

struct ChainStep {
    const unsigned short* Wf; unsigned wf_bytes;            // FRAG image of the step's filter (null: no conv - the pre step)
    const float* mask; const float* post_mask; float* out;
    int resid, save, drop, pad;
};
struct ChainDropSpec { float keep; unsigned sid_lo, sid_hi; int n_split; };
struct ChainParams {
    const float* x;
    int n_steps;                                            // 1 (pre) + convs
    ChainStep s[5];
    ChainDropSpec d[2];
    unsigned long long seed;
    const unsigned long long* ctr;
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void chain8x8_kernel(const ChainParams p) {      // (one workgroup of 8 waves per CU: 135 KB of LDS)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    float* const part = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = wave & 3, k2 = wave >> 2;
    const int img = blockIdx.x;
    const long long ioff = (long long)img * (CH_PX * CH_C);
    const unsigned long long step_ctr = p.ctr ? p.ctr[0] : 0;
    const int h = lane >> 5, l31 = lane & 31;

    // the thread's four (pixel, channel quad) items: item = it * 512 + tid -> pixel item / 32, quad item % 32 - the same in every step
    float4 val[4], slot0[4], slot1[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = it * 512 + tid;
        val[it] = *reinterpret_cast<const float4*>(p.x + ioff + (long long)item * 4);
        slot0[it] = make_float4(0.f, 0.f, 0.f, 0.f); slot1[it] = slot0[it];
    }
    int pix[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tp = j * 32 + l31;
        pix[j] = ((tp >> 3) * CH_PW + (tp & 7)) * CH_LDS_K + h * 8;
    }
    constexpr int QW[6] = {2, 0, 1, 1, 0, 0}, QX[6] = {0, 2, 1, 0, 1, 0};      // (filter piece, pixel piece): l*h, h*l, m*m, m*h, h*m, h*h
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;

    // ONE copy of the step's code, looped (five inlined copies were 60 KB of straight-line code run once per workgroup: 128 us per launch, most of
    // it instruction fetch); the step's record is picked with constant indices - a run-time index into the kernel arguments would move the
    // table to scratch
    const int ns = p.n_steps;
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
        ChainStep L;
        switch (s) {
            case 0: L = p.s[0]; break;
            case 1: L = p.s[1]; break;
            case 2: L = p.s[2]; break;
            case 3: L = p.s[3]; break;
            default: L = p.s[4]; break;
        }
        const bool last = s + 1 == ns;
        if (L.Wf) {
            // ---- conv: this wave's chunk x kout half; the patch of `val` was staged at the end of the previous step
            const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(L.Wf), 0, L.wf_bytes, 0x00020000);
            const unsigned a_voff = (unsigned)lane * 16u;
            unsigned a_soff0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)(2 * k2) * 4 + chunk) * 9) * 6144));
            unsigned a_soff1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)(2 * k2 + 1) * 4 + chunk) * 9) * 6144));
            const unsigned short* Xs = smem + chunk * CH_CHUNK;
            u32x4 fa[2][2][2][CH_NP];                     // [register set][kout block][k step][plane]
            auto loadA = [&](auto setc) __attribute__((always_inline)) {
                constexpr int SET = decltype(setc)::value;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int q = 0; q < CH_NP; ++q) {
                        fa[SET][0][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, a_soff0 + (unsigned)((ks * CH_NP + q) * 1024), 0));
                        fa[SET][1][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, a_soff1 + (unsigned)((ks * CH_NP + q) * 1024), 0));
                    }
                a_soff0 += 6144u; a_soff1 += 6144u;
            };
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            int tap_off = 0, s_cnt = 0;
            auto tap = [&](auto curc, auto nxtc, bool has_next) __attribute__((always_inline)) {
                constexpr int CUR = decltype(curc)::value;
                if (has_next) loadA(nxtc);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 fx[CH_NP][2];
#pragma unroll
                    for (int q = 0; q < CH_NP; ++q)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            fx[q][j] = *reinterpret_cast<const u32x4*>(&Xs[q * CH_PPLANE + pix[j] + tap_off * CH_LDS_K + ks * 16]);
#pragma unroll
                    for (int cl = 0; cl < 6; ++cl)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = Cvt<CH_MMA>::mma(fa[CUR][i][ks][QW[cl]], fx[QX[cl]][j], acc[i][j]);
                }
                if (++s_cnt == 3) { s_cnt = 0; tap_off += CH_PW - 2; } else ++tap_off;
            };
            loadA(set0{});
#pragma unroll 1
            for (int t2 = 0; t2 < 4; ++t2) { tap(set0{}, set1{}, true); tap(set1{}, set0{}, true); }      // taps 0..7: a LOOP (the instruction cache keeps its body)
            tap(set0{}, set1{}, false);
            // (a third register set - fragments two taps ahead - spills: 290 registers; and the stream is not latency- but rate-bound, see below)
            __syncthreads();                               // every wave has read the patch: its memory becomes the partial sums
            float* es = part + chunk * CH_PART;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<float4*>(&es[(j * 32 + l31) * CH_LDE + k2 * 64 + i * 32 + 8 * q + 4 * h]) = v;
                    }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int item = it * 512 + tid, px = item >> 5, c4 = item & 31;
                float4 v = *reinterpret_cast<const float4*>(&part[px * CH_LDE + c4 * 4]);
#pragma unroll
                for (int w = 1; w < 4; ++w) {              // chunk order 0, 1, 2, 3 whatever the waves' timing
                    const float4 a = *reinterpret_cast<const float4*>(&part[w * CH_PART + px * CH_LDE + c4 * 4]);
                    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                }
                val[it] = v;
            }
            __syncthreads();                               // the partial sums are consumed: their memory becomes the next patch
        }
        // ---- the step's epilogue on the thread's items
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * 512 + tid;
            const long long off = ioff + (long long)item * 4;
            float4 v = val[it];
            if (L.mask) {
                const float4 k = *reinterpret_cast<const float4*>(L.mask + off);
                v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
            }
            if (L.resid) {                                 // (value selects: a pointer select between the two arrays would move them to scratch)
                const float4 ra = slot0[it], rb = slot1[it];
                const bool one = L.resid == 1;
                v.x += one ? ra.x : rb.x; v.y += one ? ra.y : rb.y; v.z += one ? ra.z : rb.z; v.w += one ? ra.w : rb.w;
            }
            if (L.drop) {
                const ChainDropSpec D = L.drop == 1 ? p.d[0] : p.d[1];      // (constant indices)
                const bool hi = img >= D.n_split;
                const unsigned sid = hi ? D.sid_hi : D.sid_lo;
                const long long rel = off - (hi ? (long long)D.n_split * (CH_PX * CH_C) : 0);
                uint32_t cc[4];
                ctgan_philox::draw4(p.seed, sid, step_ctr, (uint32_t)(rel >> 2), cc);
                const float inv = 1.f / D.keep;
                v.x *= inv * floorf(D.keep + ctgan_philox::u01(cc[0])); v.y *= inv * floorf(D.keep + ctgan_philox::u01(cc[1]));
                v.z *= inv * floorf(D.keep + ctgan_philox::u01(cc[2])); v.w *= inv * floorf(D.keep + ctgan_philox::u01(cc[3]));
            }
            {
                const bool s1 = L.save == 1, s2 = L.save == 2;
                slot0[it].x = s1 ? v.x : slot0[it].x; slot0[it].y = s1 ? v.y : slot0[it].y; slot0[it].z = s1 ? v.z : slot0[it].z; slot0[it].w = s1 ? v.w : slot0[it].w;
                slot1[it].x = s2 ? v.x : slot1[it].x; slot1[it].y = s2 ? v.y : slot1[it].y; slot1[it].z = s2 ? v.z : slot1[it].z; slot1[it].w = s2 ? v.w : slot1[it].w;
            }
            if (L.post_mask) {
                const float4 k = *reinterpret_cast<const float4*>(L.post_mask + off);
                v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
            }
            if (L.out) *reinterpret_cast<float4*>(L.out + off) = v;
            val[it] = v;
        }
        if (last) break;
        // ---- `val` becomes the next conv's halo patch: three bf16 planes per chunk, halo pixels zero
        for (int i = tid; i < 36 * 4 * CH_NP * 4; i += 512) {          // 36 halo pixels x 4 chunks x 3 planes x 4 x 16 B
            const int part16 = i & 3, r = i >> 2, pl = r % CH_NP, r2 = r / CH_NP, ck = r2 & 3, hp = r2 >> 2;
            const int ppx = hp < 10 ? hp : (hp < 20 ? 90 + (hp - 10) : (1 + ((hp - 20) >> 1)) * CH_PW + (((hp - 20) & 1) ? CH_PW - 1 : 0));
            *reinterpret_cast<u32x4*>(&smem[ck * CH_CHUNK + pl * CH_PPLANE + ppx * CH_LDS_K + part16 * 8]) = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * 512 + tid, px = item >> 5, c4 = item & 31;
            const int ck = c4 >> 3, q4 = c4 & 7;
            const int ppx = ((px >> 3) + 1) * CH_PW + (px & 7) + 1;
            const float4 v = val[it];
            unsigned o0[CH_NP], o1[CH_NP];
            split_pk<CH_MMA>(v.x, v.y, o0);
            split_pk<CH_MMA>(v.z, v.w, o1);
#pragma unroll
            for (int q = 0; q < CH_NP; ++q) {
                const u32x2 o = {o0[q], o1[q]};
                *reinterpret_cast<u32x2*>(&smem[ck * CH_CHUNK + q * CH_PPLANE + ppx * CH_LDS_K + q4 * 4]) = o;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int ctgan_conv2d16_chain8x8(const ctgan_chain8x8* c, ctgan_stream_t stream) {
    if (!c || !c->x || c->n_images <= 0 || c->n_convs < 1 || c->n_convs > CTGAN_CHAIN_MAX_CONVS)
        return ctgan_fail(CTGAN_E_BADARG, "conv2d16_chain8x8: bad argument");
    if (c->channels != CH_C || c->height != CH_HW || c->width != CH_HW)
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_chain8x8: 8x8 images of 128 channels only (got %d x %d x %d)", c->height, c->width, c->channels);
    ChainParams p{};
    p.x = c->x; p.n_steps = 1 + c->n_convs; p.seed = c->drop_seed; p.ctr = reinterpret_cast<const unsigned long long*>(c->drop_ctr);
    uintptr_t align = reinterpret_cast<uintptr_t>(c->x);
    const long long plane = 9LL * CH_C * CH_C;                     // 16-bit elements per plane of a 3x3x128x128 image
    for (int s = 0; s < p.n_steps; ++s) {
        const ctgan_chain_step& in = c->step[s];
        ChainStep& L = p.s[s];
        if (s == 0) {
            if (in.wp) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_chain8x8: step 0 is the pre step (no filter)");
        } else {
            if (!in.wp) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_chain8x8: step %d has no filter image", s);
            L.Wf = static_cast<const unsigned short*>(in.wp) + 3 * plane;      // the FRAG copy behind the three planes (ctgan_conv2d16_pack_filter)
            L.wf_bytes = (unsigned)(3 * plane * 2);
        }
        if (in.resid < 0 || in.resid > 2 || in.save < 0 || in.save > 2 || in.drop < 0 || in.drop > 2)
            return ctgan_fail(CTGAN_E_BADARG, "conv2d16_chain8x8: step %d: resid / save / drop must be 0, 1 or 2", s);
        L.mask = in.mask; L.post_mask = in.post_mask; L.out = in.out; L.resid = in.resid; L.save = in.save; L.drop = in.drop;
        align |= reinterpret_cast<uintptr_t>(in.mask) | reinterpret_cast<uintptr_t>(in.post_mask) | reinterpret_cast<uintptr_t>(in.out) | reinterpret_cast<uintptr_t>(in.wp);
        if (in.drop) {
            const ctgan_chain_drop& d = c->drop[in.drop - 1];
            if (!(d.keep > 0.f) || d.keep > 1.f) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_chain8x8: keep=%g not in (0,1]", d.keep);
        }
    }
    if (align & 15) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_chain8x8: operands must be 16-byte aligned");
    for (int i = 0; i < 2; ++i) { p.d[i].keep = c->drop[i].keep > 0.f ? c->drop[i].keep : 1.f; p.d[i].sid_lo = (unsigned)c->drop[i].stream_id_lo; p.d[i].sid_hi = (unsigned)c->drop[i].stream_id_hi; p.d[i].n_split = c->drop[i].n_split; }
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(chain8x8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_LDS_BYTES) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "conv2d16_chain8x8: cannot reserve %zu B of LDS", CH_LDS_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL(chain8x8_kernel, dim3((unsigned)c->n_images), dim3(512), CH_LDS_BYTES, static_cast<hipStream_t>(stream), p);
    ctgan_set_last_kernel("chain8x8<image/workgroup>");
    ctgan_set_last_symbol("chain8x8_kernel");
    return ctgan_check_launch("conv2d16_chain8x8");
}

}  // extern "C"
