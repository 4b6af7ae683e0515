// igemm16.hip - implicit-GEMM convolution family on the gfx950 16-bit matrix cores
// (v_mfma_f32_32x32x16_bf16 / _f16: 16x the rate of the fp32 MFMA the fp32 family in igemm.hip uses).
//
// Mixed precision as BASELINE.json configs[1] / [4] name it: operands are rounded to bf16 (or fp16) on their way into
// LDS, products are accumulated in fp32 by the MFMA, master weights / activations / gradients stay fp32 in HBM.  Filters are
// packed once per weight version into a 16-bit K-major image (ctgan_conv2d16_pack_filter), so the filter operand is
// streamed at 2 bytes per element; the pixel operand is read as fp32 and converted while it is staged (v_cvt_pk_*).
//
//   FWD   : D[pixel m][kout n] = sum_{k=(r,s,c)}  X(m; r,s,c) * Wp[n][k]
//   DGRAD : the same kernel on dy: a stride-1 data gradient is ONE correlation with the rotated filter, a stride-2 data
//           gradient is FOUR (one per parity of the dx pixel) with only the taps of that parity - the phase rides the M-tile
//           index, every phase has its own tap counts (5x5: 3x3, 3x2, 2x3, 2x2 - no zero taps), pads and packed filter.
//   WGRAD : dW[(r,s,c)][kout] = sum_pixels X(pixel; r,s,c) * dY[pixel][kout]: both operands are pixel-major in memory but the
//           MFMA wants 8 consecutive k (= pixels) per lane, so the staging pass transposes: a thread loads the same 4
//           channels of 4 consecutive pixels and packs (pixel, pixel+1) pairs with the conversion itself - the transpose
//           costs no extra instruction, only 8-byte instead of 16-byte LDS writes.  Split over the pixel axis into fp32
//           slabs + fixed-order reduction (deterministic, no float atomics).
//
// Orientation: the MFMA "A" operand is always the CHANNEL-indexed one (kout for FWD/DGRAD, c for WGRAD), so a lane's four
// consecutive accumulator registers are four consecutive channels of one pixel / one filter row.
// Tiling: 4 waves as 2x2, a wave owns TMxTN 32x32 accumulators, K slices of BK = 64 (or 32) elements, two LDS stages, one
// barrier per slice, register prefetch of the next slice, buffer loads with hardware range checks for the padding taps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

#include "common.h"
#include "philox.h"

#include "mma16.h"
#include "wgrad16c.h"

namespace {


// ---------------------------------------------------------------------------------------------- filter packing
// (the three planes of the split mode follow each other, `plane` 16-bit elements apart)
// FWD : wp[n][(r*S + s)*C + c]                       = w[r][s][c][n]                    (n = kout)
// DGRAD, phase (a,b): wp[off(a,b) + n*Kph + ((t*U + u)*Kout + k)] = w[r][s][n][k]      (n = c: the data gradient's output channel)
//        r = r0(a) + step*(T(a)-1-t), s = s0(b) + step*(U(b)-1-u)   (rotated: the gather then runs forward)
struct PhaseGeom { int T[2], U[2], r0[2], s0[2], pad_t[2], pad_l[2]; int nph, step; };

PhaseGeom phase_geom(const ctgan_conv_desc* d) {
    PhaseGeom g{};
    if (d->stride == 1) {
        g.nph = 1; g.step = 1;
        g.T[0] = d->R; g.U[0] = d->S; g.r0[0] = g.s0[0] = 0;
        g.pad_t[0] = d->R - 1 - d->pad_t; g.pad_l[0] = d->S - 1 - d->pad_l;
        return g;
    }
    g.nph = 4; g.step = 2;
    for (int a = 0; a < 2; ++a) {
        g.r0[a] = (a + d->pad_t) & 1;
        g.T[a] = (d->R - g.r0[a] + 1) / 2;
        g.pad_t[a] = g.T[a] - 1 - (a + d->pad_t - g.r0[a]) / 2;
        g.s0[a] = (a + d->pad_l) & 1;
        g.U[a] = (d->S - g.s0[a] + 1) / 2;
        g.pad_l[a] = g.U[a] - 1 - (a + d->pad_l - g.s0[a]) / 2;
    }
    return g;
}
long long phase_off(const PhaseGeom& g, const ctgan_conv_desc* d, int ph) {     // 16-bit elements before phase `ph`
    long long o = 0;
    for (int q = 0; q < ph; ++q) o += (long long)d->C * g.T[q >> 1] * g.U[q & 1] * d->K;
    return o;
}

// FRAG image of the split mode (stride-1 layers the halo-patch kernel takes; appended behind the three planes): the same 16-bit
// pieces in MFMA-FRAGMENT order, so that a wave reads the "A" fragment of its 32 output channels with ONE fully coalesced 1 KB load
// and streams through it linearly - the filter operand never touches LDS (conv16x3hf_kernel):
//   block (n/32, 32-channel chunk, tap, k step of 16, plane) -> 64 lanes x 16 B, lane = 32*h + n%32 holds k = 8*h .. 8*h+7 of that step
// u32 index of the pair (k, k+1), k even, of output channel n, reduction index (tap, c = 32*chunk + 16*ks + 8*h + k):
__device__ __forceinline__ long long frag_u32_index(int n, int tap, int c, int RS, int nch, int q) {
    const int nt = n >> 5, l31 = n & 31, ch = c >> 5, cc = c & 31, ks = cc >> 4, hh = (cc >> 3) & 1, e2 = (cc & 7) >> 1;
    const long long blk = ((((long long)nt * nch + ch) * RS + tap) * 2 + ks) * 3 + q;
    return blk * 256 + (hh * 32 + l31) * 4 + e2;
}

template <int MMA>
__global__ void pack_fwd_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int RS, int C, int K, long long plane, int frag) {
    // one thread per (n, tap, pair of c): reads are strided by K (the transpose), writes are 4-byte and coalesced
    const long long per = (long long)RS * C / 2, total = per * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / per);
        const long long e = (i - (long long)n * per) * 2;                      // (tap*C + c), c even
        const float a = w[e * K + n], b = w[(e + 1) * K + n];
        unsigned o[planes<MMA>()];
        split_pk<MMA>(a, b, o);
#pragma unroll
        for (int q = 0; q < planes<MMA>(); ++q) reinterpret_cast<unsigned*>(wp + q * plane)[i] = o[q];
        if constexpr (planes<MMA>() == 3) {
            if (frag) {
                const int tap = (int)(e / C), c = (int)(e - (long long)tap * C);
#pragma unroll
                for (int q = 0; q < 3; ++q) reinterpret_cast<unsigned*>(wp + 3 * plane)[frag_u32_index(n, tap, c, RS, C >> 5, q)] = o[q];
            }
        }
    }
}
struct PackPhases { int T[2], U[2], r0[2], s0[2]; int nph, step, R, S, C, K; int frag; int pad; };
// 16-bit elements of the packed data-gradient image before phase `ph` (phase_off of the host side; computed here so that a job is 88 bytes
// and 32 of them fit the kernel arguments: one pack launch per weight version instead of two on the headline's critic)
__host__ __device__ inline long long pack_phase_off(const PackPhases& pp, int ph) {
    long long o = 0;
    for (int q = 0; q < ph; ++q) o += (long long)pp.C * pp.T[q >> 1] * pp.U[q & 1] * pp.K;
    return o;
}
template <int MMA>
__global__ void pack_dgrad_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, const PackPhases pp, long long plane) {
    const int ph = blockIdx.y, a = ph >> 1, b = ph & 1;
    const int T = pp.T[a], U = pp.U[b];
    const long long kph = (long long)T * U * pp.K, per = kph / 2, total = per * pp.C;
    const long long ph_off = pack_phase_off(pp, ph);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / per);
        const long long e = (i - (long long)n * per) * 2;                      // (t*U + u)*K + k, k even
        const int k = (int)(e % pp.K), tu = (int)(e / pp.K), t = tu / U, u = tu - t * U;
        const int r = pp.r0[a] + pp.step * (T - 1 - t), s = pp.s0[b] + pp.step * (U - 1 - u);
        const float* src = w + (((long long)r * pp.S + s) * pp.C + n) * pp.K + k;
        unsigned o[planes<MMA>()];
        split_pk<MMA>(src[0], src[1], o);
#pragma unroll
        for (int q = 0; q < planes<MMA>(); ++q) reinterpret_cast<unsigned*>(wp + q * plane + ph_off)[i] = o[q];
        if constexpr (planes<MMA>() == 3) {
            if (pp.frag) {                                   // (the reduction channel is k, the output channel n = c; stride 2: step = 4 * phase + tap)
#pragma unroll
                for (int q = 0; q < 3; ++q) reinterpret_cast<unsigned*>(wp + 3 * plane)[frag_u32_index(n, ph * T * U + tu, k, pp.nph * T * U, pp.K >> 5, q)] = o[q];
            }
        }
    }
}

// all packed images of one weight version in ONE launch (a pack per filter and operator is ~5 us of launch floor each, 57 per
// iteration on the headline): blockIdx.y = job, grid-stride over the job's pairs; a data-gradient job walks its phases in turn
struct PackJob { const float* w; unsigned short* wp; PackPhases pp; long long plane; int op; int pad; };
#define CTGAN_PACK_BATCH 32
struct PackJobs { PackJob j[CTGAN_PACK_BATCH]; };
static_assert(sizeof(PackJobs) <= 3584, "kernel arguments");
template <int MMA>
__global__ void pack_batch_kernel(const PackJobs jobs) {
    const PackJob& jb = jobs.j[blockIdx.y];
    const PackPhases& pp = jb.pp;
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (jb.op == CTGAN_CONV_FWD) {
        const long long per = (long long)pp.R * pp.S * pp.C / 2, total = per * pp.K;
        if (per % 32 == 0 && pp.K % 32 == 0) {
            // The packed image is the TRANSPOSE of the HWIO filter ([kout][pairs along (r,s,c)] from [(r,s,c)][kout]): read straight,
            // every lane of a wave touched its own cache line (stride 2 K floats) - 21 us per launch for 15 MB, eight launches per
            // iteration.  32 x 32 (pair, kout) tiles through LDS: reads coalesced along kout, writes coalesced along the pairs.
            __shared__ unsigned tl[planes<MMA>()][32][33];
            const int tiles_p = (int)(per / 32), ntiles = tiles_p * (pp.K / 32);
            for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
                const int n0 = (tile / tiles_p) * 32;
                const long long p0 = (long long)(tile % tiles_p) * 32;
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = threadIdx.x + 256 * it, n_l = idx & 31, p_l = idx >> 5;
                    const long long e = (p0 + p_l) * 2;
                    unsigned o[planes<MMA>()];
                    split_pk<MMA>(jb.w[e * pp.K + n0 + n_l], jb.w[(e + 1) * pp.K + n0 + n_l], o);
#pragma unroll
                    for (int q = 0; q < planes<MMA>(); ++q) tl[q][p_l][n_l] = o[q];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = threadIdx.x + 256 * it, p_l = idx & 31, n_l = idx >> 5;
                    const int n = n0 + n_l;
                    const long long i = (long long)n * per + p0 + p_l;
#pragma unroll
                    for (int q = 0; q < planes<MMA>(); ++q) reinterpret_cast<unsigned*>(jb.wp + q * jb.plane)[i] = tl[q][p_l][n_l];
                }
                if constexpr (planes<MMA>() == 3) {
                    if (pp.frag) {
                        // FRAG image: a lane's unit = the four pairs (8 consecutive reduction channels) of one output channel - 16 bytes, and the 32
                        // output channels of the tile are 32 consecutive units of a fragment block: one 512-byte run per (wave half, plane) instead of
                        // a 4-byte store per pair.  (A group of four pairs never straddles a tap: C is a multiple of 32 where the image exists.)
                        const int n_l = threadIdx.x & 31, g4 = threadIdx.x >> 5;         // 32 output channels x 8 groups of four pairs
                        const long long e = (p0 + g4 * 4) * 2;
                        const int tap = (int)(e / pp.C), c = (int)(e - (long long)tap * pp.C);
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const u32x4 v = {tl[q][g4 * 4][n_l], tl[q][g4 * 4 + 1][n_l], tl[q][g4 * 4 + 2][n_l], tl[q][g4 * 4 + 3][n_l]};
                            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned*>(jb.wp + 3 * jb.plane) + frag_u32_index(n0 + n_l, tap, c, pp.R * pp.S, pp.C >> 5, q)) = v;
                        }
                    }
                }
            }
            return;
        }
        for (long long i = t0; i < total; i += stride) {
            const int n = (int)(i / per);
            const long long e = (i - (long long)n * per) * 2;
            unsigned o[planes<MMA>()];
            split_pk<MMA>(jb.w[e * pp.K + n], jb.w[(e + 1) * pp.K + n], o);
#pragma unroll
            for (int q = 0; q < planes<MMA>(); ++q) reinterpret_cast<unsigned*>(jb.wp + q * jb.plane)[i] = o[q];
            if constexpr (planes<MMA>() == 3) {
                if (pp.frag) {
                    const int tap = (int)(e / pp.C), c = (int)(e - (long long)tap * pp.C);
#pragma unroll
                    for (int q = 0; q < 3; ++q) reinterpret_cast<unsigned*>(jb.wp + 3 * jb.plane)[frag_u32_index(n, tap, c, pp.R * pp.S, pp.C >> 5, q)] = o[q];
                }
            }
        }
        return;
    }
    for (int ph = 0; ph < pp.nph; ++ph) {
        const int a = ph >> 1, b = ph & 1;
        const int T = pp.T[a], U = pp.U[b];
        const long long kph = (long long)T * U * pp.K, per = kph / 2, total = per * pp.C;
        const long long ph_off = pack_phase_off(pp, ph);
        for (long long i = t0; i < total; i += stride) {
            const int n = (int)(i / per);
            const long long e = (i - (long long)n * per) * 2;
            const int k = (int)(e % pp.K), tu = (int)(e / pp.K), t = tu / U, u = tu - t * U;
            const int r = pp.r0[a] + pp.step * (T - 1 - t), sx = pp.s0[b] + pp.step * (U - 1 - u);
            const float* src = jb.w + (((long long)r * pp.S + sx) * pp.C + n) * pp.K + k;
            unsigned o[planes<MMA>()];
            split_pk<MMA>(src[0], src[1], o);
#pragma unroll
            for (int q = 0; q < planes<MMA>(); ++q) reinterpret_cast<unsigned*>(jb.wp + q * jb.plane + ph_off)[i] = o[q];
            if constexpr (planes<MMA>() == 3) {
                if (pp.frag) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) reinterpret_cast<unsigned*>(jb.wp + 3 * jb.plane)[frag_u32_index(n, ph * T * U + tu, k, pp.nph * T * U, pp.K >> 5, q)] = o[q];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- FWD / DGRAD kernel
struct P16 {
    const float* X;                 // pixel operand (fp32, channel stride 1)
    const unsigned short* Wp;       // packed filter
    const float* bias; const float* mask; const float* resid;
    float* D;
    int H, W;                       // source extent of the gather
    int P, Q;                       // pixel grid of ONE phase (rows of the GEMM = N*P*Q)
    int C;                          // channels per tap of the pixel operand (GEMM K per tap)
    int stride;                     // gather stride (1 for data gradients)
    long long s_n, s_h, s_w;        // source strides (elements)
    int M, Ng;                      // rows per phase, output channels
    long long ds_n, ds_p, ds_q;     // D strides over (n, phase-grid row, phase-grid col); channel stride 1
    int relu, relu_in;
    int resid_up;                   // halo-patch kernel only: resid is the dense channels-last [N, P/2, Q/2, Ng] tensor, added through a nearest-2x upsample
    // halo-patch kernel only: tf.nn.dropout of the result inside the epilogue, exactly as the fp32 family's (igemm.hip FwdParams::drop):
    // y *= floor(keep + u) / keep, u = element (physical offset / 4) of the Philox stream (seed, sid, ctr[0]); up to three sample
    // ranges with their own keep / stream, draws indexed from the range's first element
    int drop; float drop_keep; unsigned long long drop_seed; unsigned drop_sid; const unsigned long long* drop_ctr;
    int drop_nr; int drop_mend[CTGAN_DROP_RANGES]; float drop_rkeep[CTGAN_DROP_RANGES]; unsigned drop_rsid[CTGAN_DROP_RANGES];
    long long drop_roff[CTGAN_DROP_RANGES];
    // slice kernels only (conv16_kernel, conv16_splitk_epilogue_kernel): the LeakyReLU + dropout pair of the DCGAN critics on the result
    // (ctgan_epilogue_ext::act) - D = r * (ref > 0 ? 1 : act_alpha) * floor(keep + u) / keep, ref = act_ref ? act_ref[offset] : r; keep /
    // stream from drop_keep / drop_sid or the sample range of the row (drop_nend: first sample past the range; any boundary, the choice
    // is per lane), draws indexed as above.  D is dense channels-last.
    int act; float act_alpha; const float* act_ref; int drop_nend[CTGAN_DROP_RANGES];
    // conv16x3hf_kernel<.., BN = true> only: batch norm of the pixel operand on load (ctgan_epilogue_ext::in_bn_*); bn_per = samples per statistic group
    const float* bn_mean; const float* bn_rstd; const float* bn_scale; const float* bn_offset; const int* bn_labels; int bn_per;
    unsigned x_bytes, w_bytes;      // w_bytes covers every plane
    const unsigned short* Wf;       // FRAG image of the filter (fragment order, see frag_u32_index) or null; wf_bytes its size
    unsigned wf_bytes;
    unsigned w_plane_bytes;         // split mode: byte distance between the filter's planes
    int nph, ph_tiles_m;
    int ph_T[2], ph_U[2], ph_pad_t[2], ph_pad_l[2];
    long long ph_w_off[4];          // packed-filter element offset of a phase
    long long ph_d_h, ph_d_w;       // D offset of phase (a,b) = a*ph_d_h + b*ph_d_w
    int ksplit;                     // > 1: blockIdx.y handles slices [y*nk/ksplit, (y+1)*nk/ksplit) and writes raw sums to slab[y][phase row][n]
    float* slab; size_t slab_bytes;
    int dbg;                        // perf-diagnosis bits (env CTGAN_DBG16): 1 no LDS store, 2 no global load, 4 no barrier (a branch around the MFMAs would move the accumulators out of the AGPRs)
};

// the fused LeakyReLU + dropout pair (P16::act) on four consecutive channels at physical offset `off` of sample n: the arithmetic and the
// draws of lrelu_dropout_rng_kernel (optim_rng.hip), so the fused and the separate launches agree bit for bit
__device__ __forceinline__ void conv16_act(const P16& p, float4& v, long long off, int n, unsigned long long step) {
    float keep = p.drop_keep;
    unsigned sid = p.drop_sid, off4 = 0;
    if (p.drop_nr) {
        const bool r1 = p.drop_nr > 1 && n >= p.drop_nend[0], r2 = p.drop_nr > 2 && n >= p.drop_nend[1];
        keep = r2 ? p.drop_rkeep[2] : (r1 ? p.drop_rkeep[1] : p.drop_rkeep[0]);
        sid = r2 ? p.drop_rsid[2] : (r1 ? p.drop_rsid[1] : p.drop_rsid[0]);
        off4 = (unsigned)((r2 ? p.drop_roff[2] : (r1 ? p.drop_roff[1] : p.drop_roff[0])) >> 2);
    }
    float4 r = v;
    if (p.act_ref) r = *reinterpret_cast<const float4*>(p.act_ref + off);
    const float a = p.act_alpha, inv = 1.f / keep;
    uint32_t c[4];
    ctgan_philox::draw4(p.drop_seed, sid, step, (uint32_t)(off >> 2) - off4, c);
    v.x = v.x * (r.x > 0.f ? 1.f : a) * inv * floorf(keep + ctgan_philox::u01(c[0]));
    v.y = v.y * (r.y > 0.f ? 1.f : a) * inv * floorf(keep + ctgan_philox::u01(c[1]));
    v.z = v.z * (r.z > 0.f ? 1.f : a) * inv * floorf(keep + ctgan_philox::u01(c[2]));
    v.w = v.w * (r.w > 0.f ? 1.f : a) * inv * floorf(keep + ctgan_philox::u01(c[3]));
}

// one workgroup per CU (the staging must be woven between the MFMAs): the 8-accumulator tiles
template <int MMA, int TM, int TN> constexpr bool conv16_one_wave() { return TM * TN >= 8; }
// ONE LDS stage (two barriers per slice), two workgroups per CU: the split mode's 128x128 tile (three planes per operand: 60 KB per
// stage).  A wave of the other workgroup multiplies while this one splits and stages.  Measured on the headline's layers
// (tools/conv16_bench.py f32x3 resnet): 149-178 TFLOP/s, against 126 for one double-buffered workgroup per CU (the scheduler does
// not weave the staging between the MFMAs), 120-143 for 4 producer + 4 consumer waves per workgroup and 128-150 for the same as a
// persistent kernel: VALU and LDS stores issued beside a saturated MFMA stream on the same SIMD are not free
// (tools/mfma_loop_probe.hip: 100 VALU + 18 stores per 48 MFMAs cost 35 % of the matrix rate), whichever wave issues them.
template <int MMA, int TM, int TN> constexpr bool conv16_single_stage() { return planes<MMA>() == 3 && TM * TN >= 2; }      // (2 x 1: the 128-kout x 64-pixel tile, 46 KB: three per CU)
// pixel sub-tiles per epilogue pass (the wide tiles take several passes: <= 70 KB of staging; the single-stage tile: 35 KB)
template <int MMA, int TM, int TN> constexpr int conv16_je() { return TM >= 4 ? 1 : (TN > 2 ? 2 : (conv16_single_stage<MMA, TM, TN>() ? 1 : TN)); }

template <int MMA, int TM, int TN, int BK, bool RELU_IN, bool SPLIT = conv16_one_wave<MMA, TM, TN>()>
__global__ __launch_bounds__(256) void conv16_kernel(const P16 p) {
    // TM: 32-wide kout sub-tiles per wave ("A" operand), TN: 32-wide pixel sub-tiles per wave ("B" operand)
    constexpr int NT = 256;
    constexpr int BMP = 2 * TN * 32;                    // pixels per block
    constexpr int BNC = 2 * TM * 32;                    // kout per block
    constexpr int LDS_K = BK + 8;                       // 16-bit elements per LDS row: 144 B (BK 64) / 80 B (BK 32): conflict-free b128 reads
    constexpr int XC = BK / 4;                          // 16-B fp32 chunks per pixel row per slice
    constexpr int X_PER = BMP * XC / NT;                // fp32 float4 loads per thread per slice
    constexpr int WC = BK / 8;                          // 16-B packed chunks per filter row per slice
    constexpr int W_PER = BNC * WC / NT;
    static_assert(BMP * XC % NT == 0 && BNC * WC % NT == 0 && X_PER >= 1 && W_PER >= 1, "tile / thread mismatch");
    constexpr int NP = planes<MMA>();                   // operand planes (split mode: h, m, l)
    constexpr int XPLANE = BMP * LDS_K, WPLANE = BNC * LDS_K;
    constexpr int STAGE = NP * (XPLANE + WPLANE);       // 16-bit elements: X planes, then W planes
    // SPLIT (the one-wave-per-SIMD tile): the two LDS stages are two DIFFERENT objects - a static array and the dynamic region -
    // so the compiler knows that the stores staging slice t+1 cannot alias the fragment reads of slice t and may weave them
    // between the MFMAs (with both stages inside one dynamic array every store has to stay behind every earlier read).
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    __shared__ __attribute__((aligned(16))) unsigned short stage0_static[SPLIT ? STAGE : 8];
    constexpr bool SINGLE = conv16_single_stage<MMA, TM, TN>();
    unsigned short* const S0 = SPLIT ? stage0_static : smem;
    unsigned short* const S1 = SPLIT ? smem : (SINGLE ? smem : smem + STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;            // wave's kout half / pixel half
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);       // neighbouring pixel tiles (shared halo rows) on one XCD
    const int tiles_n = (p.Ng + BNC - 1) / BNC;
    int tile_m = bid / tiles_n;
    const int tile_n = bid - tile_m * tiles_n;
    int pa = 0, pb = 0, ph = 0;
    if (p.nph > 1) { ph = tile_m / p.ph_tiles_m; tile_m -= ph * p.ph_tiles_m; pa = ph >> 1; pb = ph & 1; }
    const int T = p.ph_T[pa], U = p.ph_U[pb], pad_t = p.ph_pad_t[pa], pad_l = p.ph_pad_l[pb];
    const long long w_off = p.ph_w_off[ph];
    const long long d_off = pa * p.ph_d_h + pb * p.ph_d_w;
    const int m0 = tile_m * BMP, n0 = tile_n * BNC;
    const int cpt = p.C / BK;                            // slices per tap
    const int nk_all = T * U * cpt;
    const int k_first = p.ksplit > 1 ? (int)((long long)nk_all * blockIdx.y / p.ksplit) : 0;
    const int nk = (p.ksplit > 1 ? (int)((long long)nk_all * (blockIdx.y + 1) / p.ksplit) : nk_all) - k_first;
    const long long kph = (long long)T * U * p.C;        // packed-filter row length of this phase
    const int PQ = p.P * p.Q;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wp), 0, p.w_bytes, 0x00020000);

    // pixel-operand loader: thread -> (row, 16-B chunk); the row's byte offset at tap (0,0) is fixed for the whole kernel
    // BK 32 (80-B LDS rows): the two rows a store instruction's 16 lanes (b64) / 8 lanes (b128) cover are 4 rows apart (320 B = 64 mod
    // 128: disjoint banks); consecutive rows would overlap in 16 of their 64 bytes
    auto spread = [](int g) { return BK == 32 ? ((g & 1) * 4 + ((g >> 1) & 3) + (g >> 3) * 8) : g; };
    const int x_chunk = tid % XC, x_row0 = spread(tid / XC);
    unsigned x_voff[X_PER];
    int x_ih0[X_PER], x_iw0[X_PER];
    bool x_valid[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        const int m = m0 + x_row0 + i * (NT / XC);
        x_valid[i] = m < p.M;
        const int mm = x_valid[i] ? m : 0;
        const int n = mm / PQ, rem = mm - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
        x_ih0[i] = pp * p.stride - pad_t;
        x_iw0[i] = qq * p.stride - pad_l;
        const long long o = (long long)n * p.s_n + (long long)x_ih0[i] * p.s_h + (long long)x_iw0[i] * p.s_w + x_chunk * 4;
        x_voff[i] = (unsigned)(o * 4);                   // may be "negative": wraps consistently mod 2^32
    }
    const int w_chunk = tid % WC, w_row0 = spread(tid / WC);
    unsigned w_voff[W_PER];
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
        const int n = n0 + w_row0 + i * (NT / WC);
        w_voff[i] = n < p.Ng ? (unsigned)((w_off + (long long)n * kph + w_chunk * 8) * 2) : 0xFFFFFFFFu;
    }

    float4 rx[X_PER];
    u32x4 rw[NP][W_PER];
    // tap / channel chunk / linear slice index of the NEXT slice to load (a K split starts in the middle of the filter)
    int ld_k = k_first, ld_c = k_first % cpt, ld_u = (k_first / cpt) % U, ld_t = (k_first / cpt) / U;
    auto load_slice = [&]() {
        const unsigned xs = (unsigned)(((long long)ld_t * p.s_h + (long long)ld_u * p.s_w + ld_c * BK) * 4);
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            const bool ok = x_valid[i] & ((unsigned)(x_ih0[i] + ld_t) < (unsigned)p.H) & ((unsigned)(x_iw0[i] + ld_u) < (unsigned)p.W);
#if defined(C16_DBG) && (C16_DBG & 2)      // diagnosis build: no pixel-operand loads (results wrong by design)
            rx[i] = make_float4((float)ok, 0.f, 0.f, 0.f);
#elif defined(C16_DBG) && (C16_DBG & 8)    // diagnosis build: half the pixel-operand bytes (what 16-bit storage would fetch)
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, ok ? (x_voff[i] + xs) / 2 : 0xFFFFFFFFu, 0, 0);
            rx[i] = make_float4(__builtin_bit_cast(float, v[0]), __builtin_bit_cast(float, v[1]), 0.f, 0.f);
#else
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? x_voff[i] + xs : 0xFFFFFFFFu, 0, 0);
            rx[i] = __builtin_bit_cast(float4, v);
#endif
        }
        const unsigned ws = (unsigned)((long long)ld_k * BK * 2);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int i = 0; i < W_PER; ++i) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[i], ws + q * p.w_plane_bytes, 0);
                rw[q][i] = __builtin_bit_cast(u32x4, v);
            }
        ++ld_k;
        // branch-free advance (scalar selects): a branch here would split the slice's basic block and the scheduler could not
        // weave the staging instructions between the MFMAs
        ++ld_c;
        const bool wc = ld_c == cpt;
        ld_c = wc ? 0 : ld_c;
        ld_u += wc ? 1 : 0;
        const bool wu = ld_u == U;
        ld_u = wu ? 0 : ld_u;
        ld_t += wu ? 1 : 0;
    };
    auto store_slice = [&](unsigned short* st) {
        unsigned short* Xs = st;
        unsigned short* Ws = st + NP * XPLANE;
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            float4 v = rx[i];
            if (RELU_IN) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            unsigned o0[NP], o1[NP];
            split_pk<MMA>(v.x, v.y, o0);
            split_pk<MMA>(v.z, v.w, o1);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const u32x2 o = {o0[q], o1[q]};
                *reinterpret_cast<u32x2*>(&Xs[q * XPLANE + (x_row0 + i * (NT / XC)) * LDS_K + x_chunk * 4]) = o;
            }
        }
#if !(defined(C16_DBG) && (C16_DBG & 16))      // diagnosis build 16: the filter operand never touches LDS (no stores, no fragment reads)
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int i = 0; i < W_PER; ++i)
                *reinterpret_cast<u32x4*>(&Ws[q * WPLANE + (w_row0 + i * (NT / WC)) * LDS_K + w_chunk * 8]) = rw[q][i];
#endif
    };

    // split mode, one accumulator per wave: its six products alternate between two accumulators (no back-to-back dependent MFMAs)
    constexpr int NACC = (NP == 3 && TM * TN == 1) ? 2 : 1;
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.f;

    const int h = lane >> 5, l31 = lane & 31;
    auto mma_slice = [&](const unsigned short* Xs) {
        const unsigned short* Ws = Xs + NP * XPLANE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            u32x4 fw[NP][TM], fx[NP][TN];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#if defined(C16_DBG) && (C16_DBG & 16)
                    fw[q][i] = u32x4{(unsigned)(size_t)Ws, (unsigned)lane, (unsigned)(q + ks), (unsigned)i};
#else
                    fw[q][i] = *reinterpret_cast<const u32x4*>(&Ws[q * WPLANE + (wm * TM * 32 + i * 32 + l31) * LDS_K + ks * 16 + h * 8]);
#endif
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fx[q][j] = *reinterpret_cast<const u32x4*>(&Xs[q * XPLANE + (wn * TN * 32 + j * 32 + l31) * LDS_K + ks * 16 + h * 8]);
            }
            if constexpr (NP == 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[0][i][j] = Cvt<MMA>::mma(fw[0][i], fx[0][j], acc[0][i][j]);
            } else {
                // (filter piece, pixel piece), the small products first: l*h, h*l, m*m, m*h, h*m, h*h
                constexpr int QW[6] = {2, 0, 1, 1, 0, 0}, QX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[c % NACC][i][j] = Cvt<MMA>::mma(fw[QW[c]][i], fx[QX[c]][j], acc[c % NACC][i][j]);
            }
        }
    };
    load_slice();
    store_slice(S0);
    if (nk > 1) load_slice();
    __syncthreads();
    // steady state (slices kt+1 and kt+2 exist): ONE basic block per slice - stage slice kt+1 into the other LDS buffer, issue
    // the loads of slice kt+2, multiply slice kt.  SPLIT: scheduling groups ask for the staging instructions to be woven between
    // the MFMAs (one wave per SIMD: nothing else fills the 32-cycle MFMA gaps): first the fragments of the first k step, then
    // per MFMA a few VALU (conversions / addresses), one LDS read, one LDS write, one buffer load.
    auto slice = [&](unsigned short* wr, const unsigned short* rd) {
        store_slice(wr);                                   // the other stage: its readers passed the last barrier
        load_slice();
        mma_slice(rd);
        if (SPLIT) {
            __builtin_amdgcn_sched_group_barrier(0x100, NP * (TM + TN), 0);
#pragma unroll
            for (int g = 0; g < (NP == 3 ? 6 : 1) * TM * TN * (BK / 16); ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
            }
        }
        __syncthreads();
    };
    int kt = 0;
    if constexpr (SINGLE) {
        for (; kt < nk; ++kt) {
            mma_slice(S0);
            __syncthreads();                               // every wave has read slice kt
            if (kt + 1 < nk) {
                store_slice(S0);
                if (kt + 2 < nk) load_slice();
                __syncthreads();
            }
        }
    } else {
        for (; kt + 3 < nk; kt += 2) { slice(S1, S0); slice(S0, S1); }      // kt even: slice kt lives in S0
        for (; kt < nk; ++kt) {                                // the last slices
            if (kt + 1 < nk) store_slice((kt + 1) & 1 ? S1 : S0);
            if (kt + 2 < nk) load_slice();
            mma_slice(kt & 1 ? S1 : S0);
            __syncthreads();
        }
    }

    // epilogue through LDS: acc[i][j][4g + e] = D(pixel j*32 + l31, kout i*32 + 8g + 4h + e); every wave transposes its own
    // (TN*32 pixels) x (TM*32 kout) block so that a lane then owns 4 consecutive channels of a pixel and a wave instruction
    // stores whole rows (16-B per lane, mask / residual / bias operands as 16-B loads).
    if constexpr (NACC == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][0][e] += acc[1][0][0][e];
    }
    constexpr int LDE = TM * 32 + 4;          // the launcher sizes the dynamic LDS for max(stage(s), this staging area)
    constexpr int JE = conv16_je<MMA, TM, TN>();
    float* es = reinterpret_cast<float*>(smem) + wave * (JE * 32 * LDE);
    constexpr int C4 = TM * 8;                // float4 per pixel row of the wave's block
    constexpr int ROWS_PER = 64 / C4;
#pragma unroll
    for (int jh = 0; jh < TN; jh += JE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < JE; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = {acc[0][i][jh + j][4 * g], acc[0][i][jh + j][4 * g + 1], acc[0][i][jh + j][4 * g + 2], acc[0][i][jh + j][4 * g + 3]};
                    *reinterpret_cast<float4*>(&es[(j * 32 + l31) * LDE + i * 32 + 8 * g + 4 * h]) = v;
                }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the wave reads back what its own lanes wrote (no cross-wave traffic)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < JE * 32 / ROWS_PER; ++it) {
            const int row = it * ROWS_PER + lane / C4, c4 = lane % C4;
            const int m = m0 + wn * TN * 32 + jh * 32 + row, col = n0 + wm * TM * 32 + c4 * 4;
            if (m >= p.M || col >= p.Ng) continue;
            float4 v = *reinterpret_cast<const float4*>(&es[row * LDE + c4 * 4]);
            if (p.ksplit > 1) {               // partial sums: the reduction kernel applies the epilogue
                *reinterpret_cast<float4*>(p.slab + (((long long)blockIdx.y * p.nph + ph) * p.M + m) * p.Ng + col) = v;
                continue;
            }
            const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
            const long long off = d_off + n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col;
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.mask) {
                const float4 k = *reinterpret_cast<const float4*>(p.mask + off);
                v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
            }
            if (p.resid) { const float4 r = *reinterpret_cast<const float4*>(p.resid + off); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (p.act) conv16_act(p, v, off, n, p.drop_ctr ? p.drop_ctr[0] : 0);
            *reinterpret_cast<float4*>(p.D + off) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();      // the next pass overwrites the staging rows this pass has just read
    }
}

// ---------------------------------------------------------------------------------------------- split mode, halo-patch form
// conv16x3h: stride-1 R x S convolutions (forward, and the data gradient of a stride-1 conv) on 128 kout x 128 pixel tiles that are
// whole image rows (128 / Q rows of a Q-wide image) or whole images (two 8x8 images, each with its own halo block).  In the slice-per-(tap, channel chunk) kernel every pixel value is loaded,
// split into its three bf16 terms and written to LDS once PER TAP - R*S times - and that staging work (VALU + LDS stores issued
// beside the MFMAs; tools/mfma_loop_probe.hip: 100 VALU + 18 stores per 48 MFMAs cost 35 % of the matrix rate) is what holds it
// at half the matrix rate.  Here a workgroup stages, per 32-channel chunk, the tile's pixels PLUS THEIR HALO once - a
// (rows + R - 1) x (Q + S - 1) patch in split form - and runs all R*S taps from that patch by shifting the fragment row; per tap
// only the 24 KB filter slice moves.  Pixel-operand loads, split VALU and LDS stores drop by R*S * 128 / patch pixels (5.6x for
// 3x3 on 32-wide images).  Single LDS stage, two workgroups of 4 waves per CU (80 KB each) as conv16_single_stage: the other
// workgroup multiplies while this one stages.
// tile = TR rows of IMGS images each (IMGS > 1: whole images smaller than the tile); per image a (TR + R - 1) x PW patch block
struct PatchGeom { int TR, PW, NPX, n_it, IMGS, PIMG; };  // rows per image, patch width, patch pixels, float4 items per thread, images, patch pixels per image

template <bool RELU_IN>
__global__ __launch_bounds__(256) void conv16x3h_kernel(const P16 p, const PatchGeom pg) {
    constexpr int MMA = CTGAN_MMA_F32X3, NP = 3, TM = 2, TN = 2, BK = 32, NT = 256, MAXIT = 8;
    constexpr int LDS_K = BK + 8, WC = BK / 8, W_PER = 128 * WC / NT;
    constexpr int WPLANE = 128 * LDS_K;
    constexpr int LDE = TM * 32 + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int PPLANE = pg.NPX * LDS_K;
    unsigned short* const Ws = smem;                      // filter stage: 3 planes x 128 kout rows
    unsigned short* const Xs = smem + NP * WPLANE;        // patch: 3 planes x NPX pixel rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int R = p.ph_T[0], S = p.ph_U[0], RS = R * S;
    const int tiles_n = p.Ng / 128;
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);       // neighbouring pixel tiles (shared halo rows) on one XCD
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int nch = p.C / BK;
    const int PQ = p.P * p.Q;
    const int img = m0 / PQ, row0 = (m0 - img * PQ) / p.Q;        // first image / first row (0 when the tile holds whole images)

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wp), 0, p.w_bytes, 0x00020000);
    // ---- filter loader (as conv16_kernel: rows 4 apart per 8 lanes - disjoint LDS banks)
    auto spread = [](int g) { return (g & 1) * 4 + ((g >> 1) & 3) + (g >> 3) * 8; };
    const int w_chunk = tid % WC, w_row0 = spread(tid / WC);
    const long long kph = (long long)RS * p.C;
    unsigned w_voff[W_PER];
#pragma unroll
    for (int i = 0; i < W_PER; ++i) w_voff[i] = (unsigned)(((long long)(n0 + w_row0 + i * (NT / WC)) * kph + w_chunk * 8) * 2);
    u32x4 rw[NP][W_PER];
    int wl_tap = 0, wl_chunk = 0;                         // next filter slice to load
    auto load_w = [&]() __attribute__((always_inline)) {
        const unsigned ws = (unsigned)(((long long)wl_tap * p.C + wl_chunk * BK) * 2);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int i = 0; i < W_PER; ++i)
                rw[q][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[i], ws + q * p.w_plane_bytes, 0));
        if (++wl_tap == RS) { wl_tap = 0; ++wl_chunk; }
    };
    auto store_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int i = 0; i < W_PER; ++i)
                *reinterpret_cast<u32x4*>(&Ws[q * WPLANE + (w_row0 + i * (NT / WC)) * LDS_K + w_chunk * 8]) = rw[q][i];
    };
    // ---- patch loader: item = it * 256 + tid -> (patch pixel = item / 8, 4-channel group = item % 8): 8 lanes read one pixel's 128 B
    float4 rp[MAXIT];
    unsigned p_voff[MAXIT];                               // byte offset of the item at chunk 0 (0xFFFFFFFF: outside the image / the patch)
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int item = it * NT + tid, px = item >> 3;
        p_voff[it] = 0xFFFFFFFFu;
        if (it < pg.n_it && px < pg.NPX) {
            const int pi = px / pg.PIMG, pr = px - pi * pg.PIMG;      // image of the tile, pixel inside its patch block
            const int prow = pr / pg.PW, pcol = pr - prow * pg.PW;
            const int ih = row0 + prow - p.ph_pad_t[0], iw = pcol - p.ph_pad_l[0];
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                p_voff[it] = (unsigned)(((long long)(img + pi) * p.s_n + (long long)ih * p.s_h + (long long)iw * p.s_w + (item & 7) * 4) * 4);
        }
    }
    auto load_patch = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
            if (it < pg.n_it)
                rp[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, p_voff[it] == 0xFFFFFFFFu ? 0xFFFFFFFFu : p_voff[it] + chunk * (BK * 4), 0, 0));
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = it * NT + tid, px = item >> 3;
            if (it < pg.n_it && px < pg.NPX) {
                float4 v = rp[it];
                if (RELU_IN) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                unsigned o0[NP], o1[NP];
                split_pk<MMA>(v.x, v.y, o0);
                split_pk<MMA>(v.z, v.w, o1);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const u32x2 o = {o0[q], o1[q]};
                    *reinterpret_cast<u32x2*>(&Xs[q * PPLANE + px * LDS_K + (item & 7) * 4]) = o;
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;
    // patch index (tap (0,0)) of this lane's pixel in fragment j: tile pixel t = wn*64 + j*32 + l31 -> (image, row, column)
    int pix[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int tp = wn * 64 + j * 32 + l31, per = pg.TR * p.Q, ti = tp / per, tr = tp - ti * per;
        pix[j] = ti * pg.PIMG + (tr / p.Q) * pg.PW + (tr % p.Q);
    }
    constexpr int QW[6] = {2, 0, 1, 1, 0, 0}, QX[6] = {0, 2, 1, 0, 1, 0};      // (filter piece, pixel piece): l*h, h*l, m*m, m*h, h*m, h*h

    load_patch(0);
    load_w();
    for (int c = 0; c < nch; ++c) {
        store_patch();                                     // every wave passed the barrier behind the previous chunk's last MFMAs
        if (c + 1 < nch) load_patch(c + 1);                // in flight during this chunk's taps
        int tap_off = 0, s_cnt = 0;
        for (int tp = 0; tp < RS; ++tp) {
            store_w();
            if (tp + 1 < RS || c + 1 < nch) load_w();
            __syncthreads();
            // pinned order (scheduling barriers): products class by class - four MFMAs on four different accumulators - fragment reads in
            // the order the classes consume them, the second k step's reads between the MFMAs of the first
            u32x4 fw[2][NP][TM], fx[2][NP][TN];
            auto rd = [&](int ks, int qw, int qx) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fw[ks][qw][i] = *reinterpret_cast<const u32x4*>(&Ws[qw * WPLANE + (wm * 64 + i * 32 + l31) * LDS_K + ks * 16 + h * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fx[ks][qx][j] = *reinterpret_cast<const u32x4*>(&Xs[qx * PPLANE + (pix[j] + tap_off) * LDS_K + ks * 16 + h * 8]);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mm = [&](int ks, int cl) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = Cvt<MMA>::mma(fw[ks][QW[cl]][i], fx[ks][QX[cl]][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            };
            rd(0, 2, 0); rd(0, 0, 2);
            mm(0, 0); rd(0, 1, 1);
            mm(0, 1); rd(1, 2, 0);
            mm(0, 2); rd(1, 0, 2);
            mm(0, 3); rd(1, 1, 1);
            mm(0, 4); mm(0, 5);
            mm(1, 0); mm(1, 1); mm(1, 2); mm(1, 3); mm(1, 4); mm(1, 5);
            __syncthreads();                               // the filter stage (and, after the last tap, the patch) may be overwritten
            if (++s_cnt == S) { s_cnt = 0; tap_off += pg.PW - (S - 1); } else ++tap_off;      // next tap: one pixel right, or the next patch row
        }
    }

    // epilogue through LDS (as conv16_kernel, one 32-pixel sub-tile per pass)
    float* es = reinterpret_cast<float*>(smem) + wave * (32 * LDE);
    constexpr int C4 = TM * 8, ROWS_PER = 64 / C4;
    // dropout of the result: the row range is workgroup-uniform (range boundaries are multiples of the 128-pixel tile, checked by the host)
    const unsigned long long drop_step = p.drop ? (p.drop_ctr ? p.drop_ctr[0] : 0) : 0;
    float dkeep = p.drop_keep;
    unsigned dsid = p.drop_sid, doff4 = 0;
    if (p.drop_nr) {
        const bool r1 = p.drop_nr > 1 && m0 >= p.drop_mend[0], r2 = p.drop_nr > 2 && m0 >= p.drop_mend[1];
        dkeep = r2 ? p.drop_rkeep[2] : (r1 ? p.drop_rkeep[1] : p.drop_rkeep[0]);
        dsid = r2 ? p.drop_rsid[2] : (r1 ? p.drop_rsid[1] : p.drop_rsid[0]);
        doff4 = (unsigned)((r2 ? p.drop_roff[2] : (r1 ? p.drop_roff[1] : p.drop_roff[0])) >> 2);
    }
    const bool do_drop = p.drop && dkeep < 1.f;
#pragma unroll
    for (int jh = 0; jh < TN; ++jh) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = {acc[i][jh][4 * q], acc[i][jh][4 * q + 1], acc[i][jh][4 * q + 2], acc[i][jh][4 * q + 3]};
                *reinterpret_cast<float4*>(&es[l31 * LDE + i * 32 + 8 * q + 4 * h]) = v;
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / ROWS_PER; ++it) {
            const int row = it * ROWS_PER + lane / C4, c4 = lane % C4;
            const int m = m0 + wn * TN * 32 + jh * 32 + row, col = n0 + wm * TM * 32 + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(&es[row * LDE + c4 * 4]);
            const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
            const long long off = n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col;
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.mask) {
                const float4 k = *reinterpret_cast<const float4*>(p.mask + off);
                v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
            }
            if (p.resid) {
                const long long ro = p.resid_up ? ((((long long)n * (p.P >> 1) + (pp >> 1)) * (p.Q >> 1) + (qq >> 1)) * p.Ng + col) : off;
                const float4 r = *reinterpret_cast<const float4*>(p.resid + ro);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (do_drop) {
                uint32_t c[4];
                ctgan_philox::draw4(p.drop_seed, dsid, drop_step, (uint32_t)(off >> 2) - doff4, c);
                const float inv = 1.f / dkeep;
                v.x *= inv * floorf(dkeep + ctgan_philox::u01(c[0])); v.y *= inv * floorf(dkeep + ctgan_philox::u01(c[1]));
                v.z *= inv * floorf(dkeep + ctgan_philox::u01(c[2])); v.w *= inv * floorf(dkeep + ctgan_philox::u01(c[3]));
            }
            *reinterpret_cast<float4*>(p.D + off) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------- halo patch, filter from L2
// conv16x3hf: the halo-patch kernel with the FILTER operand streamed straight from L2 into registers in MFMA-fragment order (FRAG image,
// frag_u32_index) instead of through an LDS stage.  What changes against conv16x3h_kernel:
//   * waves split the 128 output channels FOUR ways (a wave: 32 kout x 128 pixels, 1 x 4 accumulators), so every filter fragment is
//     loaded by exactly one wave (no duplicate L2 traffic): 6 coalesced 1 KB loads per tap per wave, one tap ahead of its MFMAs
//     (two register sets);
//   * no filter stage in LDS: no ds_write of filters (a quarter of the LDS port traffic of the old kernel), and the two barriers per
//     TAP (48 MFMAs) become two per 32-channel CHUNK (R*S*48 MFMAs) - between them the four waves run free;
//   * LDS holds the patch only (49 KB): fragment reads stay at 0.5 ds_read_b128 per MFMA (pixel fragments are not shared between
//     accumulators of a 1 x 4 tile; filter fragments are, in registers).
// Same arithmetic in the same order per accumulator element (chunk-major, tap, k step, the six products small-first): bit-identical
// results to conv16x3h_kernel (tests/test_gpu_kernels16.py).
// TN = 32-pixel sub-tiles per workgroup tile (4: 128 pixels; 2 / 1: the 64- / 32-pixel tiles of launches whose 128-pixel tiles
// could not fill the chip - the 16x16 / 8x8 layers at 64-192 rows; the filter stream per workgroup is the same, so they trade L2
// bytes per MFMA for workgroups).
// BN (round 5): training-mode batch norm of the INPUT applied while the patch is staged - x' = (x - mean[g][c]) * rstd[g][c] * scale[lab][c] +
// offset[lab][c] for the pixels inside the image (the SAME zero padding stays zero), then RELU_IN: the generator's Conv2 layers under no_grad
// (P16::bn_*; tiles inside one image: the coefficients of a thread's channel quad are four float4 per chunk).
template <bool RELU_IN, int TN, bool BN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TN <= 2 ? 3 : 2))) void conv16x3hf_kernel(const P16 p, const PatchGeom pg) {
    // (MAXIT = float4 patch items per thread the registers hold: conv16x3hf_maxit on the host side)
    constexpr int MMA = CTGAN_MMA_F32X3, NP = 3, BK = 32, NT = 256, MAXIT = TN == 4 ? 8 : (TN == 2 ? 6 : 4), BMP = TN * 32;
    constexpr int LDS_K = BK + 8;
    constexpr int LDE = 32 + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int PPLANE = pg.NPX * LDS_K;
    unsigned short* const Xs = smem;                      // patch: 3 planes x NPX pixel rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the filter stream's offset lives in SGPRs (no waterfall loop around its loads)
    const int R = p.ph_T[0], S = p.ph_U[0], RS = R * S;
    const int tiles_n = p.Ng / 128;
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);       // neighbouring pixel tiles (shared halo rows) on one XCD
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BMP, n0 = tile_n * 128;
    const int nch = p.C / BK;
    const int PQ = p.P * p.Q;
    const int img = m0 / PQ, row0 = (m0 - img * PQ) / p.Q;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wf), 0, p.wf_bytes, 0x00020000);
    // ---- filter fragment stream of this wave: output channels n0 + 32*wave .. +31; 6 KB per (chunk, tap) step, steps contiguous
    const unsigned a_voff = (unsigned)lane * 16u;
    unsigned a_soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((long long)((n0 >> 5) + wave) * nch * RS * 6144));
    u32x4 fa[2][2][NP];                                   // [register set][k step][plane]
    auto loadA = [&](auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < NP; ++q)
#if defined(HF_DBG) && (HF_DBG & 1)      // diagnosis build: the filter stream reads the same 6 KB again and again (L1 hits: what a free L2 stream would give)
                fa[SET][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, (unsigned)((ks * NP + q) * 1024), 0));
#else
                fa[SET][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, a_soff + (unsigned)((ks * NP + q) * 1024), 0));
#endif
        a_soff += 6144u;
    };
    // ---- patch loader (as conv16x3h_kernel)
    float4 rp[MAXIT];
    unsigned p_voff[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int item = it * NT + tid, px = item >> 3;
        p_voff[it] = 0xFFFFFFFFu;
        if (it < pg.n_it && px < pg.NPX) {
            const int pi = px / pg.PIMG, pr = px - pi * pg.PIMG;
            const int prow = pr / pg.PW, pcol = pr - prow * pg.PW;
            const int ih = row0 + prow - p.ph_pad_t[0], iw = pcol - p.ph_pad_l[0];
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                p_voff[it] = (unsigned)(((long long)(img + pi) * p.s_n + (long long)ih * p.s_h + (long long)iw * p.s_w + (item & 7) * 4) * 4);
        }
    }
    // BN: the chunk's coefficients of this thread's channel quad travel with its patch loads (a thread's quad is the same for all its items:
    // NT % 8 == 0) - in flight during the previous chunk's taps like the patch itself
    float4 bn_mu = make_float4(0.f, 0.f, 0.f, 0.f), bn_rs = bn_mu, bn_ga = bn_mu, bn_be = bn_mu;
    auto load_patch = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
            if (it < pg.n_it)
                rp[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, p_voff[it] == 0xFFFFFFFFu ? 0xFFFFFFFFu : p_voff[it] + chunk * (BK * 4), 0, 0));
        if constexpr (BN) {
            const int cc = chunk * BK + (tid & 7) * 4, gq = img / p.bn_per, lab = p.bn_labels ? p.bn_labels[img] : 0;
            bn_mu = *reinterpret_cast<const float4*>(p.bn_mean + (long long)gq * p.C + cc); bn_rs = *reinterpret_cast<const float4*>(p.bn_rstd + (long long)gq * p.C + cc);
            bn_ga = *reinterpret_cast<const float4*>(p.bn_scale + (long long)lab * p.C + cc); bn_be = *reinterpret_cast<const float4*>(p.bn_offset + (long long)lab * p.C + cc);
        }
    };
    auto store_patch = [&](int) __attribute__((always_inline)) {
        const float4 mu = bn_mu, rs = bn_rs, ga = bn_ga, be = bn_be;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = it * NT + tid, px = item >> 3;
            if (it < pg.n_it && px < pg.NPX) {
                float4 v = rp[it];
                if constexpr (BN) {
                    if (p_voff[it] != 0xFFFFFFFFu) {      // bn_apply_vec_kernel's operation order: ((x - mean) * rstd) * scale + offset
                        v.x = (v.x - mu.x) * rs.x * ga.x + be.x; v.y = (v.y - mu.y) * rs.y * ga.y + be.y;
                        v.z = (v.z - mu.z) * rs.z * ga.z + be.z; v.w = (v.w - mu.w) * rs.w * ga.w + be.w;
                    }
                }
                if (RELU_IN) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                unsigned o0[NP], o1[NP];
                split_pk<MMA>(v.x, v.y, o0);
                split_pk<MMA>(v.z, v.w, o1);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const u32x2 o = {o0[q], o1[q]};
                    *reinterpret_cast<u32x2*>(&Xs[q * PPLANE + px * LDS_K + (item & 7) * 4]) = o;
                }
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;
    int pix[TN];                                           // patch index (tap (0,0)) of this lane's pixel in fragment j: tile pixel 32*j + l31
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int tp = j * 32 + l31, per = pg.TR * p.Q, ti = tp / per, tr = tp - ti * per;
        pix[j] = (ti * pg.PIMG + (tr / p.Q) * pg.PW + (tr % p.Q)) * LDS_K + h * 8;
    }
    constexpr int QW[6] = {2, 0, 1, 1, 0, 0}, QX[6] = {0, 2, 1, 0, 1, 0};      // (filter piece, pixel piece): l*h, h*l, m*m, m*h, h*m, h*h

    const int T = nch * RS;
    int c = 0, tp = 0, tap_off = 0, s_cnt = 0;
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    load_patch(0);
    loadA(set0{});
    auto step = [&](auto curc, auto nxtc, bool has_next) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value;
        if (tp == 0) {                                     // chunk prologue (uniform): every wave is past the previous chunk's reads
#if defined(HF_DBG) && (HF_DBG & 2)          // diagnosis build: the patch is staged for the first chunk only (results wrong by design)
            if (c == 0)
#endif
            store_patch(c);
            if (c + 1 < nch) load_patch(c + 1);            // in flight during this chunk's taps
            __syncthreads();
            tap_off = 0; s_cnt = 0;
        }
        if (has_next) loadA(nxtc);                         // the next step's filter fragments, one step ahead
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fx[NP][TN];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fx[q][j] = *reinterpret_cast<const u32x4*>(&Xs[q * PPLANE + pix[j] + tap_off * LDS_K + ks * 16]);
#pragma unroll
            for (int cl = 0; cl < 6; ++cl)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = Cvt<MMA>::mma(fa[CUR][ks][QW[cl]], fx[QX[cl]][j], acc[j]);
        }
        if (++s_cnt == S) { s_cnt = 0; tap_off += pg.PW - (S - 1); } else ++tap_off;
        if (++tp == RS) { tp = 0; ++c; __syncthreads(); }  // the patch may be overwritten
    };
    int st = 0;
    for (; st + 1 < T; st += 2) { step(set0{}, set1{}, true); step(set1{}, set0{}, st + 2 < T); }
    if (st < T) step(set0{}, set1{}, false);

    // epilogue through LDS: a wave's 32 kout x 32 pixels per pass, transposed so that a lane stores 4 consecutive channels of one pixel
    float* es = reinterpret_cast<float*>(smem) + wave * (32 * LDE);
    constexpr int C4 = 8, ROWS_PER = 64 / C4;
    const unsigned long long drop_step = p.drop ? (p.drop_ctr ? p.drop_ctr[0] : 0) : 0;
    float dkeep = p.drop_keep;
    unsigned dsid = p.drop_sid, doff4 = 0;
    if (p.drop_nr) {
        const bool r1 = p.drop_nr > 1 && m0 >= p.drop_mend[0], r2 = p.drop_nr > 2 && m0 >= p.drop_mend[1];
        dkeep = r2 ? p.drop_rkeep[2] : (r1 ? p.drop_rkeep[1] : p.drop_rkeep[0]);
        dsid = r2 ? p.drop_rsid[2] : (r1 ? p.drop_rsid[1] : p.drop_rsid[0]);
        doff4 = (unsigned)((r2 ? p.drop_roff[2] : (r1 ? p.drop_roff[1] : p.drop_roff[0])) >> 2);
    }
    const bool do_drop = p.drop && dkeep < 1.f;
#pragma unroll
    for (int jh = 0; jh < TN; ++jh) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = {acc[jh][4 * q], acc[jh][4 * q + 1], acc[jh][4 * q + 2], acc[jh][4 * q + 3]};
            *reinterpret_cast<float4*>(&es[l31 * LDE + 8 * q + 4 * h]) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / ROWS_PER; ++it) {
            const int row = it * ROWS_PER + lane / C4, c4 = lane % C4;
            const int m = m0 + jh * 32 + row, col = n0 + wave * 32 + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(&es[row * LDE + c4 * 4]);
            const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
            const long long off = n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col;
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.mask) {
                const float4 k = *reinterpret_cast<const float4*>(p.mask + off);
                v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
            }
            if (p.resid) {
                const long long ro = p.resid_up ? ((((long long)n * (p.P >> 1) + (pp >> 1)) * (p.Q >> 1) + (qq >> 1)) * p.Ng + col) : off;
                const float4 r = *reinterpret_cast<const float4*>(p.resid + ro);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (do_drop) {
                uint32_t cc[4];
                ctgan_philox::draw4(p.drop_seed, dsid, drop_step, (uint32_t)(off >> 2) - doff4, cc);
                const float inv = 1.f / dkeep;
                v.x *= inv * floorf(dkeep + ctgan_philox::u01(cc[0])); v.y *= inv * floorf(dkeep + ctgan_philox::u01(cc[1]));
                v.z *= inv * floorf(dkeep + ctgan_philox::u01(cc[2])); v.w *= inv * floorf(dkeep + ctgan_philox::u01(cc[3]));
            }
            *reinterpret_cast<float4*>(p.D + off) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------- halo patch, one channel chunk per wave
// conv16x3hk (round 6): the halo-patch form for launches that CANNOT fill the chip with pixel tiles - the 8x8 layers of the critic at
// 64-384 rows and the 16x16 layers at 64 rows (the penalty's double backward): 16 K pixels x 128 output channels is 512 tiles of
// conv16x3hf_kernel<.,1> (32 pixels x 128 kout), every one of which streams the WHOLE filter (885 KB of fragments) out of L2 - 453 MB
// per launch, 14.6 TB/s at the measured 31 us: those launches are bound by the L2 -> CU filter stream (0.33 of 2500/6, the worst of the
// family; profiles/r05_steady_state_resnet.txt), and the 64-row ones fall back to the fp32 pipe (igemm_fwd_pipe<32x64,k4>, 18 us for
// 1.2 GFLOP).  Bigger pixel tiles halve the stream but leave 128-256 workgroups of 72 dependent (tap, chunk) steps each.
// Here the K axis is split INSIDE the workgroup instead:
//   * workgroup = 64 pixels (whole rows of one image: an 8x8 image, 4 rows of a 16-wide one) x 64 output channels; wave w owns CHANNEL
//     CHUNK w (C = 128 = 4 chunks of 32): it stages ITS chunk of the halo patch into its own LDS region (no workgroup barrier: only
//     wave-local ordering), streams the filter fragments of (2 kout blocks, chunk w, taps 0..8) from L2 - 12 KB per tap, one tap ahead -
//     and runs 9 taps x 2 k-steps x (2 x 2 accumulators) x 6 MFMAs: a dependent chain of 18 steps instead of 72;
//   * the four partial sums meet in LDS (each wave parks its 64 x 64 block in its own region), ONE barrier, then every thread sums the
//     four partials of its four float4 in the fixed order chunk 0..3 (deterministic) and applies conv16x3hf's epilogue.
// Per launch the filter stream is (M / 64) x (Ng / 64) x 442 KB = 226 MB at 16 K pixels - half of TN = 1's - from twice as many
// independent waves per pixel.  LDS: 4 x 3 planes x NPX x 80 B <= 104 KB: one workgroup (one wave per SIMD) per CU.
template <bool RELU_IN>
__global__ __launch_bounds__(256) void conv16x3hk_kernel(const P16 p, const PatchGeom pg) {
    constexpr int MMA = CTGAN_MMA_F32X3, NP = 3, BK = 32, BMP = 64, BNK = 64, MAXIT = 14;
    constexpr int LDS_K = BK + 8;
    constexpr int LDE = BNK + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int PPLANE = pg.NPX * LDS_K;
    const int WREG = NP * PPLANE;                           // 16-bit elements per wave region (>= 64 * LDE * 2: the partial sums fit)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = this wave's channel chunk
    unsigned short* const Xs = smem + wave * WREG;
    const int R = p.ph_T[0], S = p.ph_U[0], RS = R * S;
    const int tiles_n = p.Ng / BNK;
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);       // the kout halves of a pixel tile (same patch) on one XCD
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BMP, n0 = tile_n * BNK;
    const int nch = p.C / BK;                               // == 4 (the launcher checks)
    const int PQ = p.P * p.Q;
    const int img = m0 / PQ, row0 = (m0 - img * PQ) / p.Q;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wf), 0, p.wf_bytes, 0x00020000);
    // ---- filter fragment streams of this wave: kout blocks (n0 / 32) and (n0 / 32 + 1), chunk `wave`, taps contiguous (6 KB each)
    const unsigned a_voff = (unsigned)lane * 16u;
    unsigned a_soff0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)(n0 >> 5) * nch + wave) * RS) * 6144));
    unsigned a_soff1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)((n0 >> 5) + 1) * nch + wave) * RS) * 6144));
    u32x4 fa[2][2][2][NP];                                // [register set][kout block][k step][plane]
    auto loadA = [&](auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                fa[SET][0][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, a_soff0 + (unsigned)((ks * NP + q) * 1024), 0));
                fa[SET][1][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, a_soff1 + (unsigned)((ks * NP + q) * 1024), 0));
            }
        a_soff0 += 6144u; a_soff1 += 6144u;
    };
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    loadA(set0{});                                        // tap 0's fragments fly while the patch is staged
    // ---- this wave's chunk of the halo patch: NPX pixels x 8 channel quads over 64 lanes
    {
        float4 rp[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = it * 64 + lane, px = item >> 3;
            unsigned voff = 0xFFFFFFFFu;
            if (px < pg.NPX) {
                const int prow = px / pg.PW, pcol = px - prow * pg.PW;
                const int ih = row0 + prow - p.ph_pad_t[0], iw = pcol - p.ph_pad_l[0];
                if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                    voff = (unsigned)(((long long)img * p.s_n + (long long)ih * p.s_h + (long long)iw * p.s_w + wave * BK + (item & 7) * 4) * 4);
            }
            rp[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, voff, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = it * 64 + lane, px = item >> 3;
            if (px < pg.NPX) {
                float4 v = rp[it];
                if (RELU_IN) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                unsigned o0[NP], o1[NP];
                split_pk<MMA>(v.x, v.y, o0);
                split_pk<MMA>(v.z, v.w, o1);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const u32x2 o = {o0[q], o1[q]};
                    *reinterpret_cast<u32x2*>(&Xs[q * PPLANE + px * LDS_K + (item & 7) * 4]) = o;
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): the wave reads what its own lanes wrote - no cross-wave traffic yet
    __builtin_amdgcn_wave_barrier();

    f32x16 acc[2][2];                                     // [kout block][pixel block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;
    int pix[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tp = j * 32 + l31;
        pix[j] = ((tp / p.Q) * pg.PW + (tp % p.Q)) * LDS_K + h * 8;
    }
    constexpr int QW[6] = {2, 0, 1, 1, 0, 0}, QX[6] = {0, 2, 1, 0, 1, 0};      // (filter piece, pixel piece): l*h, h*l, m*m, m*h, h*m, h*h
    int tap_off = 0, s_cnt = 0;
    auto step = [&](auto curc, auto nxtc, bool has_next) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value;
        if (has_next) loadA(nxtc);                         // the next tap's filter fragments, one tap ahead
        u32x4 fx[2][NP][2];                                // [k step][plane][pixel block]: both k steps' pixel fragments read up front
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fx[ks][q][j] = *reinterpret_cast<const u32x4*>(&Xs[q * PPLANE + pix[j] + tap_off * LDS_K + ks * 16]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int cl = 0; cl < 6; ++cl)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = Cvt<MMA>::mma(fa[CUR][i][ks][QW[cl]], fx[ks][QX[cl]][j], acc[i][j]);
        if (++s_cnt == S) { s_cnt = 0; tap_off += pg.PW - (S - 1); } else ++tap_off;
    };
    int st = 0;
    for (; st + 1 < RS; st += 2) { step(set0{}, set1{}, true); step(set1{}, set0{}, st + 2 < RS); }
    if (st < RS) step(set0{}, set1{}, false);

    // ---- the four chunks' partial sums meet in LDS: each wave parks its 64 pixels x 64 kout (fp32) in its OWN region (it alone read it)
    float* const part = reinterpret_cast<float*>(smem);
    const int PREG = WREG / 2;                             // floats per wave region
    {
        float* es = part + wave * PREG;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    *reinterpret_cast<float4*>(&es[(j * 32 + l31) * LDE + i * 32 + 8 * q + 4 * h]) = v;
                }
    }
    __syncthreads();
    const unsigned long long drop_step = p.drop ? (p.drop_ctr ? p.drop_ctr[0] : 0) : 0;
    float dkeep = p.drop_keep;
    unsigned dsid = p.drop_sid, doff4 = 0;
    if (p.drop_nr) {
        const bool r1 = p.drop_nr > 1 && m0 >= p.drop_mend[0], r2 = p.drop_nr > 2 && m0 >= p.drop_mend[1];
        dkeep = r2 ? p.drop_rkeep[2] : (r1 ? p.drop_rkeep[1] : p.drop_rkeep[0]);
        dsid = r2 ? p.drop_rsid[2] : (r1 ? p.drop_rsid[1] : p.drop_rsid[0]);
        doff4 = (unsigned)((r2 ? p.drop_roff[2] : (r1 ? p.drop_roff[1] : p.drop_roff[0])) >> 2);
    }
    const bool do_drop = p.drop && dkeep < 1.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 16 + (tid >> 4), c4 = tid & 15;
        const int m = m0 + row, col = n0 + c4 * 4;
        float4 v = *reinterpret_cast<const float4*>(&part[row * LDE + c4 * 4]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {                      // chunk order 0, 1, 2, 3: the same sum whatever the waves' timing
            const float4 a = *reinterpret_cast<const float4*>(&part[w * PREG + row * LDE + c4 * 4]);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
        const long long off = n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col;
        if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (p.mask) {
            const float4 k = *reinterpret_cast<const float4*>(p.mask + off);
            v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
        }
        if (p.resid) {
            const long long ro = p.resid_up ? ((((long long)n * (p.P >> 1) + (pp >> 1)) * (p.Q >> 1) + (qq >> 1)) * p.Ng + col) : off;
            const float4 r = *reinterpret_cast<const float4*>(p.resid + ro);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (do_drop) {
            uint32_t cc[4];
            ctgan_philox::draw4(p.drop_seed, dsid, drop_step, (uint32_t)(off >> 2) - doff4, cc);
            const float inv = 1.f / dkeep;
            v.x *= inv * floorf(dkeep + ctgan_philox::u01(cc[0])); v.y *= inv * floorf(dkeep + ctgan_philox::u01(cc[1]));
            v.z *= inv * floorf(dkeep + ctgan_philox::u01(cc[2])); v.w *= inv * floorf(dkeep + ctgan_philox::u01(cc[3]));
        }
        *reinterpret_cast<float4*>(p.D + off) = v;
    }
}

// ---------------------------------------------------------------------------------------------- slice staging, filter from L2
// conv16x3sf (round 5): the split-mode forward of the launches the halo-patch kernels do not take - the folded ConvMeanPool / MeanPoolConv
// filters (4x4 / 2x2, stride 2; TF/CT_gan_cifar_resnet.py:89-98) - with the filter operand streamed from L2 in fragment order, as
// conv16x3hf does.  The slice kernel (conv16_kernel<3,...>) stages pixels AND filter through LDS and every wave reads three filter
// fragments per two accumulators: per 12 MFMAs (384 cycles) a wave moves 9 KB of fragments out of LDS and its workgroup 18 KB of operands
// into it - at three workgroups per CU more LDS cycles than MFMA cycles (0.37 of 2500/6; a diagnosis build with the filter operand out of
// LDS ran 193 -> 135 us on the 192-row 32x32 -> 16x16 layer).  Here LDS holds the pixel operand only:
//   * a workgroup = TN*32 output positions x 128 output channels; wave w owns channels 32w .. 32w+31 of ALL positions (1 x TN accumulators),
//     so every filter fragment is loaded by exactly one wave - 6 KB per (32-channel chunk, tap) step, one step ahead of its MFMAs (two
//     register sets) - and fragment reads from LDS are 0.5 ds_read_b128 per MFMA;
//   * the pixel operand of a step (TN*32 positions x 32 channels, gathered with the conv's stride, split into its three bf16 terms) goes
//     through TWO LDS stages: the loads of step s+2 are issued and step s+1 is stored while step s is multiplied, one barrier per step;
//   * steps run chunk-major, taps inside - the order of the FRAG image (frag_u32_index), so the filter stream is linear.
// Epilogue as conv16x3hf's (bias, mask, residual, relu).  Another summation order than the slice kernel: results agree to fp32 rounding.
template <bool RELU_IN, int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TN <= 2 ? 3 : 2))) void conv16x3sf_kernel(const P16 p) {
    constexpr int MMA = CTGAN_MMA_F32X3, NP = 3, BK = 32, NT = 256, BMP = TN * 32;
    constexpr int LDS_K = BK + 8, XPLANE = BMP * LDS_K, STAGE = NP * XPLANE;
    constexpr int X_PER = BMP * 8 / NT;                   // float4 items (4 channels of a position) per thread and step
    constexpr int LDE = 32 + 4;
    static_assert(X_PER >= 1, "tile / thread mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* const S0 = smem;
    unsigned short* const S1 = smem + STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = p.Ng / 128;
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);       // neighbouring position tiles (shared input rows) on one XCD
    int tile_m = bid / tiles_n;
    const int tile_n = bid - tile_m * tiles_n;
    // four-phase data gradient of a stride-2 conv (nph == 4; the folded 4x4 filters: every phase 2 x 2 taps): the workgroup's phase, its
    // taps / pads, its offset in the result and - the FRAG image of these filters is in (chunk, phase, tap) order - in the filter stream
    int ph = 0;
    if (p.nph > 1) { ph = tile_m / p.ph_tiles_m; tile_m -= ph * p.ph_tiles_m; }
    const int pa = ph >> 1, pb = ph & 1;
    const int R = p.ph_T[pa], S = p.ph_U[pb], RS = R * S;
    const int RS_all = p.nph > 1 ? p.nph * RS : RS;
    const int pad_t = p.ph_pad_t[pa], pad_l = p.ph_pad_l[pb];
    const long long d_off = pa * p.ph_d_h + pb * p.ph_d_w;
    const int m0 = tile_m * BMP, n0 = tile_n * 128;
    const int nch = p.C / BK;
    // K split (launches whose tiles cannot fill the chip): blockIdx.y handles the chunks [c0, c1) and leaves raw sums in its slab
    const int c0 = p.ksplit > 1 ? (int)((long long)nch * blockIdx.y / p.ksplit) : 0;
    const int c1 = p.ksplit > 1 ? (int)((long long)nch * (blockIdx.y + 1) / p.ksplit) : nch;
    const int PQ = p.P * p.Q;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wf), 0, p.wf_bytes, 0x00020000);
    // ---- filter fragment stream of this wave (conv16x3hf_kernel): 6 KB per step; the steps of a chunk contiguous, chunks RS_all steps apart
    const unsigned a_voff = (unsigned)lane * 16u;
    unsigned a_soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)((n0 >> 5) + wave) * nch + c0) * RS_all + ph * RS) * 6144));
    const unsigned a_skip = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((RS_all - RS) * 6144));
    int a_tap = 0;
    u32x4 fa[2][2][NP];                                   // [register set][k step][plane]
    auto loadA = [&](auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < NP; ++q)
#if defined(SF_DBG) && (SF_DBG & 1)      // diagnosis build: the filter stream re-reads one 6 KB block (results wrong by design)
                fa[SET][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, (unsigned)((ks * NP + q) * 1024), 0));
#else
                fa[SET][ks][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rsrc, a_voff, a_soff + (unsigned)((ks * NP + q) * 1024), 0));
#endif
        ++a_tap;
        const bool wrap = a_tap == RS;
        a_tap = wrap ? 0 : a_tap;
        a_soff += 6144u + (wrap ? a_skip : 0u);
    };
    // ---- pixel-operand loader: item -> (position, 4-channel group); the position's byte offset at tap (0,0), chunk 0 is fixed
    unsigned x_voff[X_PER];
    int x_ih0[X_PER], x_iw0[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        const int item = i * NT + tid, px = item >> 3;
        const int m = m0 + px;
        const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
        x_ih0[i] = pp * p.stride - pad_t;
        x_iw0[i] = qq * p.stride - pad_l;
        const long long o = (long long)n * p.s_n + (long long)x_ih0[i] * p.s_h + (long long)x_iw0[i] * p.s_w + (item & 7) * 4;
        x_voff[i] = (unsigned)(o * 4);                    // may be "negative": wraps consistently mod 2^32
    }
    float4 rx[X_PER];
    int ld_c = c0, ld_t = 0, ld_u = 0;                    // chunk / tap of the NEXT step to load (taps inside a chunk)
    auto load_x = [&]() __attribute__((always_inline)) {
        const unsigned xs = (unsigned)(((long long)ld_t * p.s_h + (long long)ld_u * p.s_w + ld_c * BK) * 4);
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            const bool ok = ((unsigned)(x_ih0[i] + ld_t) < (unsigned)p.H) & ((unsigned)(x_iw0[i] + ld_u) < (unsigned)p.W);
            rx[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? x_voff[i] + xs : 0xFFFFFFFFu, 0, 0));
        }
        ++ld_u;                                           // branch-free advance (a branch would split the step's basic block)
        const bool wu = ld_u == S;
        ld_u = wu ? 0 : ld_u;
        ld_t += wu ? 1 : 0;
        const bool wt = ld_t == R;
        ld_t = wt ? 0 : ld_t;
        ld_c += wt ? 1 : 0;
    };
    auto store_x = [&](unsigned short* Xs) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            const int item = i * NT + tid, px = item >> 3;
            float4 v = rx[i];
            if (RELU_IN) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            unsigned o0[NP], o1[NP];
            split_pk<MMA>(v.x, v.y, o0);
            split_pk<MMA>(v.z, v.w, o1);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const u32x2 o = {o0[q], o1[q]};
                *reinterpret_cast<u32x2*>(&Xs[q * XPLANE + px * LDS_K + (item & 7) * 4]) = o;
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;
    constexpr int QW[6] = {2, 0, 1, 1, 0, 0}, QX[6] = {0, 2, 1, 0, 1, 0};      // (filter piece, pixel piece): l*h, h*l, m*m, m*h, h*m, h*h

    const int T = (c1 - c0) * RS;
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    load_x();
    loadA(set0{});
    store_x(S0);
    __syncthreads();
    if (T > 1) load_x();
    auto step = [&](auto curc, auto nxtc, const unsigned short* rd, unsigned short* wr, int st) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value;
        if (st + 1 < T) loadA(nxtc);                       // the next step's filter fragments, one step ahead
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fx[NP][TN];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fx[q][j] = *reinterpret_cast<const u32x4*>(&rd[q * XPLANE + (j * 32 + l31) * LDS_K + ks * 16 + h * 8]);
#pragma unroll
            for (int cl = 0; cl < 6; ++cl)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = Cvt<MMA>::mma(fa[CUR][ks][QW[cl]], fx[QX[cl]][j], acc[j]);
        }
        if (st + 1 < T) {
#if defined(SF_DBG) && (SF_DBG & 2)          // diagnosis build: no split / LDS stores after the first two steps
            if (st < 1)
#endif
            store_x(wr);                                   // step st+1 (its readers of two steps ago passed the last barrier)
#if defined(SF_DBG) && (SF_DBG & 4)          // diagnosis build: no pixel-operand loads after the first steps
            if (st < 1)
#endif
            if (st + 2 < T) load_x();
        }
#if !(defined(SF_DBG) && (SF_DBG & 8))       // diagnosis build 8: no barrier per step
        __syncthreads();
#endif
    };
    int st = 0;
    for (; st + 1 < T; st += 2) { step(set0{}, set1{}, S0, S1, st); step(set1{}, set0{}, S1, S0, st + 1); }
    if (st < T) step(set0{}, set1{}, S0, S1, st);

    // epilogue through LDS (conv16x3hf_kernel): a wave's 32 kout x 32 positions per pass
    float* es = reinterpret_cast<float*>(smem) + wave * (32 * LDE);
    constexpr int C4 = 8, ROWS_PER = 64 / C4;
#pragma unroll
    for (int jh = 0; jh < TN; ++jh) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = {acc[jh][4 * q], acc[jh][4 * q + 1], acc[jh][4 * q + 2], acc[jh][4 * q + 3]};
            *reinterpret_cast<float4*>(&es[l31 * LDE + 8 * q + 4 * h]) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / ROWS_PER; ++it) {
            const int row = it * ROWS_PER + lane / C4, c4 = lane % C4;
            const int m = m0 + jh * 32 + row, col = n0 + wave * 32 + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(&es[row * LDE + c4 * 4]);
            if (p.ksplit > 1) {               // partial sums: conv16_splitk_epilogue_kernel applies the epilogue
                *reinterpret_cast<float4*>(p.slab + (((long long)blockIdx.y * p.nph + ph) * p.M + m) * p.Ng + col) = v;
                continue;
            }
            const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
            const long long off = d_off + n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col;
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.mask) {
                const float4 k = *reinterpret_cast<const float4*>(p.mask + off);
                v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
            }
            if (p.resid) { const float4 r = *reinterpret_cast<const float4*>(p.resid + off); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(p.D + off) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------- WGRAD kernel
struct W16 {
    const float* X; const float* DY;
    float* OUT;                      // [splits][Mtot][Ng] fp32 slabs (the final dw when splits == 1)
    int H, W, P, Q, C, R, S, stride, pad_t, pad_l;
    long long s_n, s_h, s_w;         // x strides (channel stride 1)
    int Mtot, Ng, Kg;                // R*S*C, kout, N*P*Q
    int chunk;                       // pixels per split (multiple of 64)
    int relu_x;
    unsigned x_bytes, dy_bytes;
    int dbg;
    int pq_shift, q_shift;           // log2(P*Q), log2(Q) when both are powers of two, else -1: pixel -> (n,p,q) without integer division
    int with_bias;                   // slab row Mtot receives the column sums of dy (bias gradient), summed by the workgroups of the first M tile
};

// one workgroup per CU, staging woven between the MFMAs: the 256x128 tile
template <int MMA, int TM, int TN> constexpr bool wgrad16_one_wave() { return TM * TN == 8; }
// one LDS stage, two workgroups per CU: the 128x128 tile of the split mode (see conv16_single_stage)
template <int MMA, int TM, int TN> constexpr bool wgrad16_single_stage() { return planes<MMA>() == 3 && TM * TN == 4; }

// RMODE: relu on the x operand - 0 never, 1 always, 2 per problem (p.relu_x; the grouped launch)
template <int MMA, int TM, int TN, int RMODE, bool SPLIT = wgrad16_one_wave<MMA, TM, TN>()>
__device__ __forceinline__ void wgrad16_body(const W16& p, const int bx, const int by) {
    // block tile: (2*TM*32) channels of ONE tap  x  (2*TN*32) kout, K slices of 64 pixels (32 in the split mode: three planes per operand)
    constexpr int NP = planes<MMA>();
    constexpr int NT = 256, BKP = NP == 3 ? 32 : 64;
    constexpr int PGS = BKP / 4;                        // groups of 4 pixels per slice
    constexpr int BMC = 2 * TM * 32, BNK = 2 * TN * 32;
    constexpr int LDS_K = BKP + 8;
    constexpr int XPLANE = BMC * LDS_K, YPLANE = BNK * LDS_K;
    constexpr int STAGE = NP * (XPLANE + YPLANE);       // X planes, then dY planes
    constexpr int XG = BMC / 4;                         // 4-channel groups of the x tile
    constexpr int X_PER = XG * PGS / NT;                // (4 pixels x 4 channels) blocks per thread per slice
    constexpr int YG = BNK / 4;
    constexpr int XPG = XG >= 32 ? 8 : 16, YPG = YG >= 32 ? 8 : 16;      // pixel groups a wave's load instruction spans (see the staging map)
    constexpr int Y_PER = YG * PGS / NT;
    static_assert(XG * PGS % NT == 0 && YG * PGS % NT == 0 && X_PER >= 1 && Y_PER >= 1 && XPG <= PGS && YPG <= PGS, "tile / thread mismatch");
    // SPLIT (the one-wave-per-SIMD tile): two DIFFERENT LDS objects for the two stages, see conv16_kernel
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    __shared__ __attribute__((aligned(16))) unsigned short stage0_static[SPLIT ? STAGE : 8];
    constexpr bool SINGLE = wgrad16_single_stage<MMA, TM, TN>();
    unsigned short* const S0 = SPLIT ? stage0_static : smem;
    unsigned short* const S1 = SPLIT ? smem : (SINGLE ? smem : smem + STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.Ng + BNK - 1) / BNK;
    const int tile_m = bx / tiles_n, tile_n = bx - tile_m * tiles_n;
    const int cblocks = p.C / BMC;                       // M tiles per tap
    const int tap = tile_m / cblocks, c0 = (tile_m - tap * cblocks) * BMC;
    const int r = tap / p.S, s = tap - r * p.S;
    const int n0 = tile_n * BNK;
    const int k_begin = by * p.chunk;
    const int k_end = min(k_begin + p.chunk, p.Kg);
    const int nk = (k_end - k_begin + BKP - 1) / BKP;
    const int PQ = p.P * p.Q;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, p.dy_bytes, 0x00020000);

    // staging map: thread -> (pixel group of 4 = tid % 8 [+ 8 for odd b], channel group of 4 = tid / 8 [+ 32 per pair of b]):
    // a wave load instruction touches 8 pixels x 128 B - whole cache lines (16 pixel groups x 64 B per wave moves half-used
    // lines over the L2 -> L1 path); its LDS writes land 2-way = the minimum for 512 B.  64-wide tiles keep the 16 x 64 B map.
    // operand registers: the wide tile (one wave per SIMD, 512 registers) keeps TWO slices in flight.  Measured neutral (339 vs 337
    // TFLOP/s at 1024 channels): the waves' 61 % wait share (tools/pmc_run16b.sh) is not a latency x bytes-in-flight limit; kept
    // because it costs nothing at one wave per SIMD and removes the load latency from the list of suspects.
    constexpr int NSET = SPLIT ? 2 : 1;
    float4 rxv[NSET][X_PER][4], ryv[NSET][Y_PER][4];
    const bool bias_wg = p.with_bias && tile_m == 0;
    float4 bsum[Y_PER];
#pragma unroll
    for (int b = 0; b < Y_PER; ++b) bsum[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned x_wstep = (unsigned)p.s_w * 4u, x_estep = (unsigned)p.stride * x_wstep;
    auto load_slice = [&](int kt, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int kbase = k_begin + kt * BKP;
#pragma unroll
        for (int b = 0; b < X_PER; ++b) {
            const int pg = tid % XPG + XPG * (b % (PGS / XPG)), cg = tid / XPG + (NT / XPG) * (b / (PGS / XPG));      // 16 pixel groups, XG channel groups
            const int pix = kbase + pg * 4;              // 4 consecutive pixels: same image row (Q % 4 == 0)
            const bool inr = pix < k_end;
            const int pc = inr ? pix : 0;
            int n, pp, qq;
            if (SPLIT || p.pq_shift >= 0) {  // (the wide tile is only planned for power-of-two grids: no branch inside its slice)
                n = pc >> p.pq_shift;
                const int rem = pc & (PQ - 1);
                pp = rem >> p.q_shift; qq = rem & (p.Q - 1);
            } else {
                n = pc / PQ;
                const int rem = pc - n * PQ;
                pp = rem / p.Q; qq = rem - pp * p.Q;
            }
            const int ih = pp * p.stride - p.pad_t + r;
            const bool rowok = inr & ((unsigned)ih < (unsigned)p.H);
            // 32-bit byte offsets (the launcher checks that the tensor spans < 4 GiB)
            const unsigned base = ((unsigned)n * (unsigned)p.s_n + (unsigned)ih * (unsigned)p.s_h + (unsigned)(c0 + cg * 4)) * 4u;
            const int iw0 = qq * p.stride - p.pad_l + s;
            const unsigned w0 = base + (unsigned)iw0 * x_wstep;          // one multiply per block; the 4 pixels step by a kernel constant
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = rowok & ((unsigned)(iw0 + e * p.stride) < (unsigned)p.W);
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? w0 + e * x_estep : 0xFFFFFFFFu, 0, 0);
                rxv[SET][b][e] = __builtin_bit_cast(float4, v);
            }
        }
#pragma unroll
        for (int b = 0; b < Y_PER; ++b) {
            const int pg = tid % YPG + YPG * (b % (PGS / YPG)), cg = tid / YPG + (NT / YPG) * (b / (PGS / YPG));
            const int pix = kbase + pg * 4;
            const bool colok = (n0 + cg * 4) < p.Ng;
            const unsigned ybase = ((unsigned)pix * (unsigned)p.Ng + (unsigned)(n0 + cg * 4)) * 4u;
            const unsigned ystep = (unsigned)p.Ng * 4u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = colok & (pix + e < k_end);
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, ok ? ybase + e * ystep : 0xFFFFFFFFu, 0, 0);
                ryv[SET][b][e] = __builtin_bit_cast(float4, v);
            }
        }
    };
    auto store_slice = [&](unsigned short* st, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        unsigned short* Xs = st;
        unsigned short* Ys = st + NP * XPLANE;
        // transpose by register naming: channel j of pixels 0..3 -> two packed dwords (per plane)
        auto put = [&](unsigned short* dst, int plane, float a0, float a1, float a2, float a3) {
            unsigned o0[NP], o1[NP];
            split_pk<MMA>(a0, a1, o0);
            split_pk<MMA>(a2, a3, o1);
#pragma unroll
            for (int q = 0; q < NP; ++q) { const u32x2 o = {o0[q], o1[q]}; *reinterpret_cast<u32x2*>(dst + q * plane) = o; }
        };
#pragma unroll
        for (int b = 0; b < X_PER; ++b) {
            const int pg = tid % XPG + XPG * (b % (PGS / XPG)), cg = tid / XPG + (NT / XPG) * (b / (PGS / XPG));
            float4 (&v)[4] = rxv[SET][b];
            if (RMODE == 1 || (RMODE == 2 && p.relu_x)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e].x = fmaxf(v[e].x, 0.f); v[e].y = fmaxf(v[e].y, 0.f); v[e].z = fmaxf(v[e].z, 0.f); v[e].w = fmaxf(v[e].w, 0.f); }
            }
            unsigned short* dst = &Xs[(cg * 4) * LDS_K + pg * 4];
            put(dst, XPLANE, v[0].x, v[1].x, v[2].x, v[3].x);
            put(dst + LDS_K, XPLANE, v[0].y, v[1].y, v[2].y, v[3].y);
            put(dst + 2 * LDS_K, XPLANE, v[0].z, v[1].z, v[2].z, v[3].z);
            put(dst + 3 * LDS_K, XPLANE, v[0].w, v[1].w, v[2].w, v[3].w);
        }
#pragma unroll
        for (int b = 0; b < Y_PER; ++b) {
            const int pg = tid % YPG + YPG * (b % (PGS / YPG)), cg = tid / YPG + (NT / YPG) * (b / (PGS / YPG));
            const float4 (&v)[4] = ryv[SET][b];
            if (bias_wg && p.with_bias == 1) {      // workgroup-uniform: the bias gradient rides the staging of dy (rows past the chunk were loaded as zeros); with_bias == 2: the row stays zero
                bsum[b].x += (v[0].x + v[1].x) + (v[2].x + v[3].x); bsum[b].y += (v[0].y + v[1].y) + (v[2].y + v[3].y);
                bsum[b].z += (v[0].z + v[1].z) + (v[2].z + v[3].z); bsum[b].w += (v[0].w + v[1].w) + (v[2].w + v[3].w);
            }
            unsigned short* dst = &Ys[(cg * 4) * LDS_K + pg * 4];
            put(dst, YPLANE, v[0].x, v[1].x, v[2].x, v[3].x);
            put(dst + LDS_K, YPLANE, v[0].y, v[1].y, v[2].y, v[3].y);
            put(dst + 2 * LDS_K, YPLANE, v[0].z, v[1].z, v[2].z, v[3].z);
            put(dst + 3 * LDS_K, YPLANE, v[0].w, v[1].w, v[2].w, v[3].w);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int h = lane >> 5, l31 = lane & 31;
    auto mma_slice = [&](const unsigned short* Xs) {
        const unsigned short* Ys = Xs + NP * XPLANE;
#pragma unroll
        for (int ks = 0; ks < BKP / 16; ++ks) {
            u32x4 fa[NP][TM], fb[NP][TN];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[q][i] = *reinterpret_cast<const u32x4*>(&Xs[q * XPLANE + (wm * TM * 32 + i * 32 + l31) * LDS_K + ks * 16 + h * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[q][j] = *reinterpret_cast<const u32x4*>(&Ys[q * YPLANE + (wn * TN * 32 + j * 32 + l31) * LDS_K + ks * 16 + h * 8]);
            }
            if constexpr (NP == 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Cvt<MMA>::mma(fa[0][i], fb[0][j], acc[i][j]);
            } else {
                constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};      // small products first (see conv16_kernel)
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = Cvt<MMA>::mma(fa[QA[c]][i], fb[QB[c]][j], acc[i][j]);
            }
        }
    };
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, NSET - 1>;
    int kt = 0;
    if (SPLIT) {
        // Two register sets: slice k's operands live in set (k & 1).  Steady state per slice k: stage slice k+1 from its set
        // into the other LDS object, refill that set with slice k+3 (two slices of latency slack), multiply slice k.
        if (nk > 0) { load_slice(0, set0{}); store_slice(S0, set0{}); }
        if (nk > 1) load_slice(1, set1{});
        if (nk > 2) load_slice(2, set0{});
        __syncthreads();
        auto slice = [&](unsigned short* wr, const unsigned short* rd, int k, auto set_c) {
            store_slice(wr, set_c);                       // slice k+1
            load_slice(k + 3, set_c);                     // slice k+3 into the registers just drained (out-of-range rows read as zeros)
            mma_slice(rd);
            __builtin_amdgcn_sched_group_barrier(0x100, NP * (TM + TN), 0);
#pragma unroll
            for (int g = 0; g < (NP == 3 ? 6 : 1) * TM * TN * (BKP / 16); ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);      // VALU (conversions, addresses, halo checks)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
            }
            __syncthreads();
        };
        // slice k+1 is staged while slice k is multiplied; both must exist.  Loads past the chunk return zeros and are never staged.
        for (; kt + 2 < nk; kt += 2) { slice(S1, S0, kt, set1{}); slice(S0, S1, kt + 1, set0{}); }
        // here slices kt (in S0) and, if it exists, kt+1 (registers of set 1) remain
        if (kt + 1 < nk) store_slice(S1, set1{});
        if (kt < nk) mma_slice(S0);
        __syncthreads();
        if (kt + 1 < nk) mma_slice(S1);
    } else {
        if (nk > 0) {
            load_slice(0, set0{});
            store_slice(S0, set0{});
            if (nk > 1) load_slice(1, set0{});
        }
        __syncthreads();
        if constexpr (SINGLE) {
            for (; kt < nk; ++kt) {
                mma_slice(S0);
                if (kt + 1 < nk) {
                    __syncthreads();                       // every wave has read slice kt
                    store_slice(S0, set0{});
                    if (kt + 2 < nk) load_slice(kt + 2, set0{});
                    __syncthreads();
                }
            }
        } else {
            for (; kt < nk; ++kt) {
                if (kt + 1 < nk && !(p.dbg & 1)) store_slice((kt + 1) & 1 ? S1 : S0, set0{});
                if (kt + 2 < nk && !(p.dbg & 2)) load_slice(kt + 2, set0{});
                mma_slice(kt & 1 ? S1 : S0);
                if (!(p.dbg & 4)) __syncthreads();
            }
        }
    }
    // acc[i][j][4g + e] = dW(channel c0 + wm*TM*32 + i*32 + 8g + 4h + e, kout n0 + wn*TN*32 + j*32 + l31): 32 lanes = 128 B rows
    float* out = p.OUT + (long long)by * (p.Mtot + (p.with_bias ? 1 : 0)) * p.Ng;
    if (bias_wg) {
        // column sums of this chunk's dy tile: a thread holds 4 channels of its pixel groups; fold the pixel groups through LDS in a
        // fixed order (deterministic), one thread per channel writes slab row Mtot
        __syncthreads();
        float* red = reinterpret_cast<float*>(S0);
#pragma unroll
        for (int b = 0; b < Y_PER; ++b) {
            const int pg = tid % YPG + YPG * (b % (PGS / YPG)), cg = tid / YPG + (NT / YPG) * (b / (PGS / YPG));
            *reinterpret_cast<float4*>(&red[pg * BNK + cg * 4]) = bsum[b];
        }
        __syncthreads();
        if (tid < BNK && n0 + tid < p.Ng) {
            float t = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < PGS; ++g2) t += red[g2 * BNK + tid];
            out[(long long)p.Mtot * p.Ng + n0 + tid] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + l31;
            if (col >= p.Ng) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = c0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                out[((long long)tap * p.C + c) * p.Ng + col] = acc[i][j][e];
            }
        }
}

template <int MMA, int TM, int TN, bool RELU_X>
__global__ __launch_bounds__(256) void wgrad16_kernel(const W16 p) {
    wgrad16_body<MMA, TM, TN, RELU_X ? 1 : 0>(p, (int)blockIdx.x, (int)blockIdx.y);
}

// Several weight-gradient problems (different filters, or several (x, dy) uses of one filter) in ONE launch: the small launches of a
// step - 8x8 layers, the 64-row gradient-penalty segments - each fill a fraction of the chip and pay a launch of their own; together
// they fill it.  Workgroups [first[j], first[j+1]) belong to problem j, tile-major inside a split (the splits of one tile are far apart).
constexpr int W16_GROUP_MAX = 20;
struct W16Group { int n; int first[W16_GROUP_MAX + 1]; int tiles[W16_GROUP_MAX]; W16 j[W16_GROUP_MAX]; };
template <int MMA, int TM, int TN>
__global__ __launch_bounds__(256) void wgrad16_group_kernel(const W16Group g) {
    // (An XCD-aware order - the tap tiles of one split on one XCD's L2, logical index = the XCD's contiguous range - was measured: 14.246
    // vs 14.253 ms per iteration, no effect, although the launch misses L2 on 70 % of its requests (profiles/r03_pmc_traffic_x3.json):
    // the misses are served by the memory-side cache at a rate the kernel does not wait for.  Not kept.)
    int job = 0;
    while (job + 1 < g.n && (int)blockIdx.x >= g.first[job + 1]) ++job;
    job = __builtin_amdgcn_readfirstlane(job);
    const int local = (int)blockIdx.x - g.first[job];
    const int tiles = g.tiles[job];
    const int by = local / tiles;
    wgrad16_body<MMA, TM, TN, 2>(g.j[job], local - by * tiles, by);
}

// the reductions of all problems of a grouped launch in one launch: job = blockIdx.y; out = sum of the slabs [+ add] (fixed order)
constexpr int R16_BATCH = 16;
struct R16Job { const float* part; float* out; float* out2; const float* add; const float* add2; long long n, n_main; int splits, pad; };
struct R16Jobs { int n; int pad; R16Job j[R16_BATCH]; };
__global__ void reduce16_batch_kernel(const R16Jobs jobs) {
    const R16Job& jb = jobs.j[blockIdx.y];
    const long long n = jb.n, n_main = jb.n_main;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float* part = jb.part;
    const int splits = jb.splits;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(part + (long long)(k + 0) * n + i);
        const float4 v1 = *reinterpret_cast<const float4*>(part + (long long)(k + 1) * n + i);
        const float4 v2 = *reinterpret_cast<const float4*>(part + (long long)(k + 2) * n + i);
        const float4 v3 = *reinterpret_cast<const float4*>(part + (long long)(k + 3) * n + i);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; k < splits; ++k) {
        const float4 v0 = *reinterpret_cast<const float4*>(part + (long long)k * n + i);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    float4 r;
    r.x = (a0.x + a1.x) + (a2.x + a3.x); r.y = (a0.y + a1.y) + (a2.y + a3.y);
    r.z = (a0.z + a1.z) + (a2.z + a3.z); r.w = (a0.w + a1.w) + (a2.w + a3.w);
    if (i < n_main) {
        if (jb.add) { const float4 v = *reinterpret_cast<const float4*>(jb.add + i); r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        *reinterpret_cast<float4*>(jb.out + i) = r;
    } else {
        if (jb.add2) { const float4 v = *reinterpret_cast<const float4*>(jb.add2 + (i - n_main)); r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        *reinterpret_cast<float4*>(jb.out2 + (i - n_main)) = r;
    }
}

// elements [0, n_main) go to `out`, the trailing n - n_main (the bias row of the slabs) to `out2`; n, n_main multiples of 4
__global__ void reduce16_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ out2, long long n, long long n_main, int splits) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int k = 0;
    for (; k + 2 <= splits; k += 2) {                                  // fixed order: deterministic
        const float4 v0 = *reinterpret_cast<const float4*>(part + (long long)k * n + i);
        const float4 v1 = *reinterpret_cast<const float4*>(part + (long long)(k + 1) * n + i);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    }
    if (k < splits) {
        const float4 v0 = *reinterpret_cast<const float4*>(part + (long long)k * n + i);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    const float4 r = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    if (i < n_main) *reinterpret_cast<float4*>(out + i) = r;
    else *reinterpret_cast<float4*>(out2 + (i - n_main)) = r;
}

// ---------------------------------------------------------------------------------------------- host side
bool mma_ok(int mma) { return mma == CTGAN_MMA_BF16 || mma == CTGAN_MMA_F16 || mma == CTGAN_MMA_F32X3; }
// Does the split-mode packed image of (d, op) carry the FRAG copy (frag_u32_index) behind its planes?  A function of the filter's
// shape only (never of N, H, W: the image is cached per filter and operator): stride 1, several taps, 32-channel chunks on the
// reduction side, 128-channel tiles on the output side - the shapes conv16x3hf_kernel takes.
// ... and the data gradient of the folded ConvMeanPool / UpsampleConv filters (4x4, stride 2, pad 1) for the stride-2 halo kernel
// (conv16s2.h): its image in (phase, tap) step order.
bool s2_halo_shape(const ctgan_conv_desc* d) { return d->stride == 2 && d->R == 4 && d->S == 4 && d->pad_t == 1 && d->pad_l == 1 && !d->x_up; }
bool frag_image_shape(const ctgan_conv_desc* d, int op) {
    const int nout = op == CTGAN_CONV_FWD ? d->K : d->C, cred = op == CTGAN_CONV_FWD ? d->C : d->K;
    if (!(op == CTGAN_CONV_FWD || op == CTGAN_CONV_DGRAD) || nout % 128 != 0 || cred % 32 != 0) return false;
    if (d->stride == 1) return d->R * d->S >= 2;
    if (op == CTGAN_CONV_FWD) return d->stride == 2 && d->R * d->S >= 2 && !d->x_up;      // conv16x3sf_kernel (the folded 4x4 / 2x2 stride-2 filters)
    return s2_halo_shape(d);
}
int mma_planes(int mma) { return mma == CTGAN_MMA_F32X3 ? 3 : 1; }
int dbg16() { static const int v = [] { const char* e = getenv("CTGAN_DBG16"); return e ? atoi(e) : 0; }(); return v; }

bool shape_ok_fwd(const ctgan_conv_desc* d) {
    return !d->x_up && d->C % 32 == 0 && d->K % 4 == 0 && d->xs[1] == 1 && d->ys[1] == 1 &&
           d->xs[0] % 4 == 0 && d->xs[2] % 4 == 0 && d->xs[3] % 4 == 0 && d->ys[0] % 4 == 0 && d->ys[2] % 4 == 0 && d->ys[3] % 4 == 0;
}
bool shape_ok_dgrad(const ctgan_conv_desc* d) {
    if (d->x_up || d->K % 32 != 0 || d->C % 4 != 0 || d->xs[1] != 1 || d->ys[1] != 1) return false;
    if (d->xs[0] % 4 || d->xs[2] % 4 || d->xs[3] % 4 || d->ys[0] % 4 || d->ys[2] % 4 || d->ys[3] % 4) return false;
    if (d->stride == 2) return !(d->H & 1) && !(d->W & 1) && d->P * 2 == d->H && d->Q * 2 == d->W;
    return d->stride == 1;
}
bool shape_ok_wgrad(const ctgan_conv_desc* d) {
    return !d->x_up && d->C % 64 == 0 && d->K % 4 == 0 && d->Q % 4 == 0 && d->xs[1] == 1 && d->xs[0] % 4 == 0 && d->xs[2] % 4 == 0 &&
           d->xs[3] % 4 == 0 && d->ys[1] == 1 && d->ys[3] == d->K && d->ys[2] == (int64_t)d->Q * d->K && d->ys[0] == (int64_t)d->P * d->Q * d->K;
}

// D = epilogue(sum over K splits of slab[s][phase row][n]): fixed order, 16-B accesses
__global__ __launch_bounds__(256) void conv16_splitk_epilogue_kernel(const P16 p) {
    const long long rows = (long long)p.nph * p.M;
    const int n4 = p.Ng / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n4) return;
    const long long rowi = i / n4;
    const int col = (int)(i - rowi * n4) * 4;
    const int ph = (int)(rowi / p.M), m = (int)(rowi - (long long)ph * p.M);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < p.ksplit; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(p.slab + ((long long)s * rows + rowi) * p.Ng + col);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    const int PQ = p.P * p.Q;
    const int n = m / PQ, rem = m - n * PQ, pp = rem / p.Q, qq = rem - pp * p.Q;
    const long long off = (ph >> 1) * p.ph_d_h + (ph & 1) * p.ph_d_w + n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col;
    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (p.mask) {
        const float4 k = *reinterpret_cast<const float4*>(p.mask + off);
        v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
    }
    if (p.resid) { const float4 r = *reinterpret_cast<const float4*>(p.resid + off); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (p.act) conv16_act(p, v, off, n, p.drop_ctr ? p.drop_ctr[0] : 0);
    *reinterpret_cast<float4*>(p.D + off) = v;
}

// K split of a forward / data-gradient launch whose pixel x kout tiles cannot fill the chip (8x8 / 4x4 layers at batch 64:
// 1024 pixels x 512 kout = 128 tiles of 64x64 for K = 6400).  Returns the split count for `blocks` tiles and `nk` slices.
int conv16_ksplit(long long blocks, int nk) {
    if (blocks >= 256 || nk < 32) return 1;
    int s = (int)(512 / blocks);
    if (s > 8) s = 8;
    while (s > 1 && nk / s < 12) --s;
    return s < 2 ? 1 : s;
}

template <int MMA, int TM, int TN, int BK>
int launch_conv16(const P16& p, hipStream_t st, const char* name) {
    constexpr int BMP = 2 * TN * 32, BNC = 2 * TM * 32;
    constexpr bool split = conv16_one_wave<MMA, TM, TN>();            // one stage is a static array (see the kernel)
    constexpr size_t lds_stages = (size_t)((split || conv16_single_stage<MMA, TM, TN>()) ? 1 : 2) * planes<MMA>() * (BMP + BNC) * (BK + 8) * 2;
    constexpr size_t lds_epi = (size_t)4 * conv16_je<MMA, TM, TN>() * 32 * (TM * 32 + 4) * 4;
    constexpr size_t lds = lds_stages > lds_epi ? lds_stages : lds_epi;
    auto kern = p.relu_in ? conv16_kernel<MMA, TM, TN, BK, true> : conv16_kernel<MMA, TM, TN, BK, false>;
    static bool attr[2] = {false, false};
    if (!attr[p.relu_in ? 1 : 0]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "conv16: cannot reserve %zu B of LDS", lds);
        attr[p.relu_in ? 1 : 0] = true;
    }
    const int tiles_m = (p.M + BMP - 1) / BMP, tiles_n = (p.Ng + BNC - 1) / BNC;
    P16 q = p;
    q.ph_tiles_m = tiles_m;
    q.dbg = dbg16();
    if (!(q.ksplit > 1 && q.slab)) { q.ksplit = 1; q.slab = nullptr; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(q.nph * tiles_m * tiles_n), (unsigned)q.ksplit), dim3(256), lds, st, q);
    ctgan_set_last_kernel(name);
    ctgan_set_last_symbol("conv16_kernel<%d, %d, %d, %d, %s, %s>", MMA, TM, TN, BK, p.relu_in ? "true" : "false", conv16_one_wave<MMA, TM, TN>() ? "true" : "false");
    int rc = ctgan_check_launch(name);
    if (rc || q.ksplit == 1) return rc;
    const long long n = (long long)q.nph * q.M * (q.Ng / 4);
    hipLaunchKernelGGL(conv16_splitk_epilogue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q);
    return ctgan_check_launch("conv16_splitk_epilogue");
}

// the halo-patch form takes: one phase, gather stride 1, whole-row tiles inside one image (128 % Q == 0, P*Q % 128 == 0), kout a multiple of
// 128, 32-channel chunks, more than one tap, a patch of at most 256 pixels (8 float4 items per thread)
size_t conv16x3h_lds(const PatchGeom& g) {
    const size_t stages = (size_t)3 * (128 + g.NPX) * 40 * 2, epi = (size_t)4 * 32 * 68 * 4;
    return stages > epi ? stages : epi;
}
bool conv16x3h_ok(const P16& p, PatchGeom* out, int bmp = 128) {
    if (p.nph != 1 || p.stride != 1 || p.Ng % 128 || p.C % 32 || p.Q <= 0 || p.M % bmp) return false;
    const int PQ = p.P * p.Q;
    if (!(PQ % bmp == 0 && bmp % p.Q == 0) && !(PQ < bmp && bmp % PQ == 0)) return false;      // whole rows of one image, or whole images
    const int R = p.ph_T[0], S = p.ph_U[0];
    if (R * S < 2) return false;
    PatchGeom g;
    g.IMGS = PQ < bmp ? bmp / PQ : 1;
    g.TR = PQ < bmp ? p.P : bmp / p.Q;
    g.PW = p.Q + S - 1; g.PIMG = (g.TR + R - 1) * g.PW; g.NPX = g.IMGS * g.PIMG;
    g.n_it = (g.NPX * 8 + 255) / 256;
    if (g.n_it > (bmp == 128 ? 8 : (bmp == 64 ? 6 : 4)) || conv16x3h_lds(g) > 80 * 1024) return false;      // the kernels' patch registers (MAXIT); two workgroups per CU
    if (out) *out = g;
    return true;
}
// Pixels per workgroup tile of the fragment-streaming halo kernel for this launch.  Preference order by image size, from the tile sweep
// (profiles/r03_hf_tile_sweep_*.txt; TFLOP/s forward at 128 / 64 / 32 pixels): 32x32 images 196-221 / 173-195 / 155-184; 16x16 images
// 177 / 186-188 / 169-175 (64 rows: 95 / 144 / 150); 8x8 images 50-76 / 83-117 / 108-127 - the first that qualifies with >= 192 tiles,
// else the first that qualifies.  0: no tile shape qualifies.
int conv16x3hf_tile(const P16& p) {
    static const int pref_big[3] = {128, 64, 32}, pref_16[3] = {64, 128, 32}, pref_8[3] = {32, 64, 128};
    const int PQ = p.P * p.Q;
    const int* pref = PQ >= 1024 ? pref_big : (PQ >= 256 ? pref_16 : pref_8);
    int first = 0;
    for (int k = 0; k < 3; ++k) {
        const int bmp = pref[k];
        if (!conv16x3h_ok(p, nullptr, bmp)) continue;
        if (!first) first = bmp;
        if ((long long)(p.M / bmp) * (p.Ng / 128) >= 192) return bmp;
    }
    return first;
}
// the 8x8 layers in the hybrid (fp32-mode) routing: 1 (default) = 32-pixel x 128-kout tiles of the fragment-streaming kernel from 192 rows up
// (768 workgroups of 15 KB LDS and 159 registers for the 384-row shared forward: three per CU, all co-resident, three waves per SIMD -
// where 128-pixel tiles gave 192 workgroups for 256 CUs); 0 = round 2: only the 384-row launches, on the LDS-staged halo kernel.
// Measured on one box: 15.69 vs 16.08 ms per iteration (64-pixel x 64-kout tiles with waves 2 x 2: the same 15.69 - dropped).
int x3_8x8_mode() { return 1; }
// Which launches the hybrid routing hands to the fragment-streaming kernel (tools/conv16_bench.py on the headline's layers,
// profiles/r03_conv_bench_*.txt, profiles/r03_hf_tile_sweep_*.txt; fp32 family for comparison: 112-129 on 32x32 / 16x16 images at 128-320 rows,
// 103 at (64, 16x16), 69 / 89 / 94 / 104 at 8x8 images of 64 / 128 / 192 / 384 rows): every launch whose preferred tile yields >= 192 workgroups on
// images of >= 256 pixels; 8x8 images per x3_8x8_mode() from 128 rows up (>= 256 tiles of 32 pixels; 64 rows: 53 against the fp32 family's 69 - stays there).
bool conv16x3hf_wins(const P16& p) {
    const int bmp = conv16x3hf_tile(p);
    if (!bmp) return false;
    const long long tiles = (long long)(p.M / bmp) * (p.Ng / 128);
    const int PQ = p.P * p.Q;
    if (PQ >= 256) return tiles >= 192;
    const int min_tiles = 256;     // 256 = from 128 rows: 15.50 vs 15.65 ms per iteration on one box (384: round-3 mid-state; 128 = 64 rows: 15.79)
    return PQ == 64 && x3_8x8_mode() == 1 && bmp == 32 && tiles >= min_tiles;
}
// CTGAN_X3_HALO_V=1: the filter through an LDS stage (conv16x3h_kernel, 128-pixel tiles only); default 2: filter fragments streamed
// from L2 (conv16x3hf_kernel, 128- / 64- / 32-pixel tiles)
int g_halo_version_override = 0;      // tests: ctgan_debug_x3_halo_version
int halo_version() {
    return g_halo_version_override ? g_halo_version_override : 2;
}
bool conv16x3hf_usable(const P16& p) { return halo_version() != 1 && p.Wf != nullptr && conv16x3hf_tile(p) > 0; }

template <bool RELU_IN, int TN, bool BN = false>
int launch_conv16x3hf_t(const P16& p, const PatchGeom& pg, hipStream_t st) {
    const size_t epi = (size_t)4 * 32 * 36 * 4, stage = (size_t)3 * pg.NPX * 40 * 2;
    const size_t lds = stage > epi ? stage : epi;
    static size_t have = 0;
    if (have < lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv16x3hf_kernel<RELU_IN, TN, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "conv16x3hf: cannot reserve %zu B of LDS", lds);
        have = lds;
    }
    P16 q = p;
    q.ph_tiles_m = p.M / (TN * 32);
    q.dbg = dbg16();
    q.ksplit = 1; q.slab = nullptr;
    hipLaunchKernelGGL((conv16x3hf_kernel<RELU_IN, TN, BN>), dim3((unsigned)(q.ph_tiles_m * (p.Ng / 128))), dim3(256), lds, st, q, pg);
    ctgan_set_last_kernel(BN ? (TN == 4 ? "conv16x3hf<128x128,k32,bn>" : (TN == 2 ? "conv16x3hf<64x128,k32,bn>" : "conv16x3hf<32x128,k32,bn>"))
                             : (TN == 4 ? "conv16x3hf<128x128,k32>" : (TN == 2 ? "conv16x3hf<64x128,k32>" : "conv16x3hf<32x128,k32>")));
    ctgan_set_last_symbol("conv16x3hf_kernel<%s, %d, %s>", RELU_IN ? "true" : "false", TN, BN ? "true" : "false");      // (as rocprofv3 prints it: every template argument)
    return ctgan_check_launch("conv16x3hf");
}

// conv16x3hk_kernel (one channel chunk per wave, 64-pixel x 64-kout tiles): which launches it takes.  The halo-patch shapes with C = 128 whose
// 64-pixel tiles lie inside one image, without the input batch norm, when the launch is ONE round of one workgroup per CU (<= 256
// workgroups: 8x8 images up to 128 rows - the penalty's double backward and the generator step's critic pass - 16x16 up to 32).  Measured
// kernel-only (tools/hk_prof.sh, profiles/r06_hk_prof.txt): 8x8 at 64 rows 17.5 us against 20.6 (conv16x3hf<.,1>) and 18.8 (fp32 pipe, where
// these launches ran), 128 rows 18.9 against 22.2 / 29.5; from 192 rows on the pixel-tiled kernels are level or ahead (30.1 / 28.2, 44.8 / 41.2 at
// 384 rows; 16x16 at 128 rows 57.4 / 49.1): a workgroup's 432 MFMAs per wave are 6 us of a 17 us round - the rest is launch ramp, patch staging
// and the reduction, which more rounds do not amortise.  g_hk: 0 = never (tests / A-B: ctgan_debug_x3_hk), 1 = by the rule, 2 = every launch
// that qualifies whatever its size.
int g_hk = 1;
int g_hk_max_wgs = 256;      // one round of one workgroup per CU: beyond it the pixel-tiled kernels are level or ahead (profiles/r06_hk_prof.txt)
bool conv16x3hk_takes(const P16& p, PatchGeom* out) {
    if (!g_hk || p.nph != 1 || p.stride != 1 || !p.Wf || p.bn_mean || p.act || p.C != 128 || p.Ng % 128 || p.M % 64) return false;
    PatchGeom g;
    if (!conv16x3h_ok(p, &g, 64) || g.IMGS != 1 || g.NPX > 14 * 8 || g.NPX * 60 < 64 * 68) return false;      // (14 patch items per lane; a wave region holds its 64 x 68 partial sums)
    const long long wgs = (long long)(p.M / 64) * (p.Ng / 64);
    if (g_hk == 1 && wgs > g_hk_max_wgs) return false;
    if (out) *out = g;
    return true;
}
template <bool RELU_IN>
int launch_conv16x3hk_t(const P16& p, const PatchGeom& pg, hipStream_t st) {
    const size_t bytes = (size_t)4 * 3 * pg.NPX * 40 * 2;    // four wave regions of three planes (each >= the 64 x 68 floats of its partial sums: conv16x3hk_takes)
    static size_t have = 0;
    if (have < bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv16x3hk_kernel<RELU_IN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "conv16x3hk: cannot reserve %zu B of LDS", bytes);
        have = bytes;
    }
    P16 q = p;
    q.ph_tiles_m = p.M / 64;
    q.ksplit = 1; q.slab = nullptr;
    hipLaunchKernelGGL((conv16x3hk_kernel<RELU_IN>), dim3((unsigned)(q.ph_tiles_m * (p.Ng / 64))), dim3(256), bytes, st, q, pg);
    ctgan_set_last_kernel("conv16x3hk<64x64,chunk/wave>");
    ctgan_set_last_symbol("conv16x3hk_kernel<%s>", RELU_IN ? "true" : "false");
    return ctgan_check_launch("conv16x3hk");
}

// conv16x3sf_kernel: strided forward launches with a FRAG image; tile = 128 positions from 512 tiles up, 64 from 384, 32 (2x2 filters) from 192.  0: the launch stays on the slice kernel (no image, ragged tiles, too few workgroups for a kernel without K split)
int g_s2fwd = 1;                      // tests / A-B: ctgan_debug_x3_s2fwd(0) puts the strided forward launches back on the slice kernel
int g_sf_ksplit = 1;                  // A/B: ctgan_debug_x3_s2fwd(2) = no K split on conv16x3sf_kernel
int conv16x3sf_ksplit(const P16& p) {
    if (!g_sf_ksplit || p.nph != 1 || p.M % 64 || p.C < 64 || !p.slab || p.ph_T[0] * p.ph_U[0] <= 4) return 1;
    const long long tiles = (long long)(p.M / 64) * (p.Ng / 128);
    if (tiles < 160 || tiles >= 384) return 1;      // (tools/conv16_bench.py f32x3 s2: 128 tiles 37 -> 39 us, 192 tiles 57 -> 50, 256 tiles 64 -> 56, 320 tiles 81 -> 78)
    return (size_t)2 * p.M * p.Ng * sizeof(float) <= p.slab_bytes ? 2 : 1;
}
// The four-phase data gradients of the folded 4x4 filters: one phase per workgroup on conv16x3sf_kernel at 64-position tiles (three workgroups
// of four waves per CU instead of conv16x3p_kernel's one of eight; bit-identical results - the same accumulation order).  tools/sf_dgrad_check.py:
// 8x8 dy grids 84.3 -> 71.5 us at 320 rows, 88.3 -> 79.4 at 192 rows x 256 channels; 16x16 dy grids level in the micro-benchmark (155 / 157 us at
// 192 rows) and ahead in the step: iteration 13.01 -> 12.93 ms with every such launch here (128-position tiles: 13.07).
int g_s2dgrad_sf = 0;                 // tests / A-B: ctgan_debug_x3_s2dgrad_sf(-1) = none of them (conv16x3p_kernel)
int conv16x3sf_tile(const P16& p) {
    if (p.nph == 4) {
        // (the FRAG image of these filters exists for the 4x4 / stride-2 / pad-1 shape only: every phase 2 x 2 taps)
        if (g_s2dgrad_sf < 0 || !p.Wf || p.Ng % 128 || p.C % 32 || p.drop || p.act || p.resid_up || p.relu_in || p.ph_T[0] != 2 || p.ph_T[1] != 2 || p.ph_U[0] != 2 || p.ph_U[1] != 2) return 0;
        const long long kt = 4LL * (p.Ng / 128);
        return (p.M % 64 == 0 && (p.M / 64) * kt >= 768) ? 64 : 0;
    }
    if (!g_s2fwd || p.nph != 1 || p.stride != 2 || !p.Wf || p.Ng % 128 || p.C % 32 || p.drop || p.act || p.resid_up || p.M % 32) return 0;
    // one accumulator per output element and no K split: reductions up to 2,304 terms (4x4x128, 3x3x256: 864 accumulation steps) - the chain of the
    // halo-patch kernels at 2x their length; longer ones (the DCGAN 5x5x128 layers in the fp32 mode) keep the slice kernel's two accumulators / K split
    if ((long long)p.ph_T[0] * p.ph_U[0] * p.C > 2304) return 0;
    const long long kt = p.Ng / 128;
#ifdef SF_TUNE      // (A/B builds: thresholds from the environment)
    static const int t128 = [] { const char* e = getenv("CTGAN_SF_T128"); return e ? atoi(e) : 512; }();
    static const int t64 = [] { const char* e = getenv("CTGAN_SF_T64"); return e ? atoi(e) : 384; }();
    static const int t32 = [] { const char* e = getenv("CTGAN_SF_T32"); return e ? atoi(e) : 192; }();
#else
    constexpr int t128 = 512, t64 = 384, t32 = 192;      // (64 rows of 32x32 -> 16x16 = 256 tiles of 64: 74.4 us in the step against the slice kernel's 63.8)
#endif
    if (p.M % 128 == 0 && (p.M / 128) * kt >= t128) return 128;
    if (p.M % 64 == 0 && (p.M / 64) * kt >= t64) return 64;
    // 160 .. 383 tiles of 64 positions (4x4 layers at 192 rows of 16x16 / 64 rows of 32x32): two workgroups per tile, each half the chunks,
    // raw sums to slabs + conv16_splitk_epilogue_kernel (conv16x3sf_ksplit)
    if (conv16x3sf_ksplit(p) > 1) return 64;
    // 32-position tiles (a wave streams 6 KB of filter per 12 MFMAs): only the short reductions of the 2x2 shortcut filters gain (in the step:
    // 22.2 -> 18.8 us); the 4x4 layers at 192 rows of 16x16 / 64 rows of 32x32 measured 61.0 / 64.1 us against the slice kernel's 57.1 / 63.8
    return (p.ph_T[0] * p.ph_U[0] <= 4 && (p.M / 32) * kt >= t32) ? 32 : 0;
}
template <bool RELU_IN, int TN>
int launch_conv16x3sf_t(const P16& p, hipStream_t st) {
    const size_t epi = (size_t)4 * 32 * 36 * 4, stages = (size_t)2 * 3 * (TN * 32) * 40 * 2;
    const size_t lds = stages > epi ? stages : epi;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv16x3sf_kernel<RELU_IN, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "conv16x3sf: cannot reserve %zu B of LDS", lds);
        attr = true;
    }
    P16 q = p;
    q.ksplit = (TN == 2 && (long long)(p.M / 64) * (p.Ng / 128) < 384) ? conv16x3sf_ksplit(p) : 1;
    if (q.ksplit == 1) q.slab = nullptr;
    q.ph_tiles_m = p.M / (TN * 32);
    hipLaunchKernelGGL((conv16x3sf_kernel<RELU_IN, TN>), dim3((unsigned)(p.nph * q.ph_tiles_m * (p.Ng / 128)), (unsigned)q.ksplit), dim3(256), lds, st, q);
    ctgan_set_last_kernel(q.ksplit > 1 ? "conv16x3sf<64x128,k32,ksplit>" : (TN == 4 ? "conv16x3sf<128x128,k32>" : (TN == 2 ? "conv16x3sf<64x128,k32>" : "conv16x3sf<32x128,k32>")));
    ctgan_set_last_symbol("conv16x3sf_kernel<%s, %d>", RELU_IN ? "true" : "false", TN);
    int rc = ctgan_check_launch("conv16x3sf");
    if (rc || q.ksplit == 1) return rc;
    const long long n = (long long)q.nph * q.M * (q.Ng / 4);
    hipLaunchKernelGGL(conv16_splitk_epilogue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q);
    return ctgan_check_launch("conv16_splitk_epilogue");
}
int launch_conv16x3sf(const P16& p, int bmp, hipStream_t st) {
    if (p.relu_in) return bmp == 128 ? launch_conv16x3sf_t<true, 4>(p, st) : (bmp == 64 ? launch_conv16x3sf_t<true, 2>(p, st) : launch_conv16x3sf_t<true, 1>(p, st));
    return bmp == 128 ? launch_conv16x3sf_t<false, 4>(p, st) : (bmp == 64 ? launch_conv16x3sf_t<false, 2>(p, st) : launch_conv16x3sf_t<false, 1>(p, st));
}

int launch_conv16x3h(const P16& p, hipStream_t st) {
    {
        PatchGeom pk;                 // launches that cannot fill the chip with pixel tiles: one channel chunk per wave (conv16x3hk_kernel)
        if (conv16x3hk_takes(p, &pk)) return p.relu_in ? launch_conv16x3hk_t<true>(p, pk, st) : launch_conv16x3hk_t<false>(p, pk, st);
    }
    // whole small images per 128-pixel tile (8x8): the LDS-staged kernel is the faster one when it applies
    const bool prefer_v1 = !g_halo_version_override && x3_8x8_mode() == 0 && p.P * p.Q < 128 && conv16x3h_ok(p, nullptr) && (long long)(p.M / 128) * (p.Ng / 128) >= 192;
    if (p.bn_mean) {                  // batch norm of the input on load: the fragment-streaming kernel, tiles inside one image, ReLU behind the norm
        PatchGeom pg;
        const int bmp = conv16x3hf_usable(p) ? conv16x3hf_tile(p) : 0;
        if (!bmp || !conv16x3h_ok(p, &pg, bmp) || pg.IMGS != 1 || !p.relu_in)
            return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv16x3hf: input batch norm needs the fragment-streaming halo kernel, tiles inside one image and relu_in");
        if (bmp == 128) return launch_conv16x3hf_t<true, 4, true>(p, pg, st);
        if (bmp == 64) return launch_conv16x3hf_t<true, 2, true>(p, pg, st);
        return launch_conv16x3hf_t<true, 1, true>(p, pg, st);
    }
    if (conv16x3hf_usable(p) && !prefer_v1) {
        const int bmp = conv16x3hf_tile(p);
        PatchGeom pg;
        conv16x3h_ok(p, &pg, bmp);
        if (bmp == 128) return p.relu_in ? launch_conv16x3hf_t<true, 4>(p, pg, st) : launch_conv16x3hf_t<false, 4>(p, pg, st);
        if (bmp == 64) return p.relu_in ? launch_conv16x3hf_t<true, 2>(p, pg, st) : launch_conv16x3hf_t<false, 2>(p, pg, st);
        return p.relu_in ? launch_conv16x3hf_t<true, 1>(p, pg, st) : launch_conv16x3hf_t<false, 1>(p, pg, st);
    }
    PatchGeom pg;
    if (!conv16x3h_ok(p, &pg)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv16x3h: shape outside the halo-patch form");
    const size_t lds = conv16x3h_lds(pg);
    auto kern = p.relu_in ? conv16x3h_kernel<true> : conv16x3h_kernel<false>;
    static size_t reserved[2] = {0, 0};
    size_t& have = reserved[p.relu_in ? 1 : 0];
    if (have < lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "conv16x3h: cannot reserve %zu B of LDS", lds);
        have = lds;
    }
    P16 q = p;
    q.ph_tiles_m = p.M / 128;
    q.dbg = dbg16();
    q.ksplit = 1; q.slab = nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)(q.ph_tiles_m * (p.Ng / 128))), dim3(256), lds, st, q, pg);
    ctgan_set_last_kernel("conv16x3h<128x128,k32>");
    ctgan_set_last_symbol("conv16x3h_kernel<%s>", p.relu_in ? "true" : "false");
    return ctgan_check_launch("conv16x3h<128x128,k32>");
}
// can launch_conv16x3h take this launch (either kernel)?
bool halo_takes(const P16& p) { return conv16x3hf_usable(p) || conv16x3h_ok(p, nullptr); }

#include "conv16s2.h"
int g_s2halo = 1;                     // tests / A-B: ctgan_debug_x3_s2halo(0) puts the stride-2 data gradients back on the slice kernel
int x3_s2halo() { return g_s2halo; }

template <int MMA>
int dispatch_conv16_tiles(const P16& p, bool small, hipStream_t st);

template <int MMA>
int dispatch_conv16(const P16& p, hipStream_t st) {
    // small pixel grids (8x8 / 4x4 layers at batch 64): 64x64 tiles expose 4x the workgroups
    const long long big_tiles = (long long)p.nph * ((p.M + 127) / 128) * ((p.Ng + 127) / 128);
    const bool small = big_tiles < 192 || p.Ng % 128 != 0;
    if constexpr (planes<MMA>() == 3) {
        if (conv16x3hk_takes(p, nullptr)) return launch_conv16x3h(p, st);
        if (p.nph == 4 && x3_s2halo() && conv16x3p_tile(p) && big_tiles >= 96) {
            if (const int bmp = conv16x3sf_tile(p)) return launch_conv16x3sf(p, bmp, st);
            return launch_conv16x3p(p, st);
        }
        if (p.nph == 1)
            if (const int bmp = conv16x3sf_tile(p)) return launch_conv16x3sf(p, bmp, st);
    }
    P16 q = p;
    q.ksplit = 1;
    if (small && p.slab) {
        const int bk = (p.C % 64 == 0 && planes<MMA>() == 1) ? 64 : 32;
        int nk_min = 1 << 30;
        for (int ph = 0; ph < p.nph; ++ph) { const int nk = p.ph_T[ph >> 1] * p.ph_U[ph & 1] * (p.C / bk); if (nk < nk_min) nk_min = nk; }
        const long long blocks = (long long)p.nph * ((p.M + 63) / 64) * ((p.Ng + 63) / 64);
        q.ksplit = conv16_ksplit(blocks, nk_min);
        if ((size_t)q.ksplit * p.nph * p.M * p.Ng * sizeof(float) > p.slab_bytes) q.ksplit = 1;
    }
    if (q.ksplit == 1) q.slab = nullptr;
    return dispatch_conv16_tiles<MMA>(q, small, st);
}

template <int MMA>
int dispatch_conv16_tiles(const P16& p, bool small, hipStream_t st) {
    if constexpr (planes<MMA>() == 3) {
        // split mode: stride-1 whole-row tiles go to the halo-patch kernel; everything else to the slice kernels with three planes per
        // operand in LDS - 32-deep slices, the 128x128 tile with ONE 60 KB stage (two workgroups per CU)
        const int halo = 1;
        if (halo_takes(p)) {
            const int bmp = conv16x3hf_usable(p) ? conv16x3hf_tile(p) : 128;
            if (!small || halo == 2 || (bmp < 128 && (long long)(p.M / bmp) * (p.Ng / 128) >= 96)) return launch_conv16x3h(p, st);
        }
        if (small) return launch_conv16<MMA, 1, 1, 32>(p, st, p.ksplit > 1 ? "conv16x3<64x64,k32,ksplit>" : "conv16x3<64x64,k32>");
        {
            // launches whose 128x128 tiles leave workgroup slots empty (< 512: the stride-2 layers at 128-192 rows have 256-384) run on
            // 128-kout x 64-pixel tiles, 46 KB of LDS and 112 registers: three workgroups per CU.  Measured (one box): (192, 32x32, 4x4 s2)
            // forward 144 -> 152, (192, 16x16) data gradient 125 -> 135 TFLOP/s, iteration 16.13 -> 16.02 ms.  CTGAN_X3_SLICE64=0: off, 2: always.
            const int s64 = 1;
            const long long t128 = (long long)p.nph * ((p.M + 127) / 128) * (p.Ng / 128);
            if (s64 && p.Ng % 128 == 0 && (s64 == 2 || t128 < 512)) return launch_conv16<MMA, 2, 1, 32>(p, st, "conv16x3<128x64,k32>");
        }
        return launch_conv16<MMA, 2, 2, 32>(p, st, "conv16x3<128x128,k32>");
    } else
    if (p.C % 64 == 0) {
        if (small) return launch_conv16<MMA, 1, 1, 64>(p, st, p.ksplit > 1 ? "conv16<64x64,k64,ksplit>" : "conv16<64x64,k64>");
        // (wider forward tiles - 128 kout x 256 pixels, 256 kout x 128 pixels, one wave per SIMD - measured slower / neutral against this
        // tile at two waves per SIMD and removed: DESIGN_HISTORY 4.3)
        {
            // launches whose 128x128 tiles leave workgroup slots empty (< 512) on 128-kout x 64-pixel tiles: config[1] 7.90 -> 7.82 ms per
            // iteration, config[4] unchanged (170.9 vs 170.4-171.1).  CTGAN_CONV16_TILE64=0: off.
            const int t64 = 1;
            const long long t128 = (long long)p.nph * ((p.M + 127) / 128) * (p.Ng / 128);
            if (t64 && p.Ng % 128 == 0 && t128 < 512) return launch_conv16<MMA, 2, 1, 64>(p, st, "conv16<128x64,k64>");
        }
        return launch_conv16<MMA, 2, 2, 64>(p, st, "conv16<128x128,k64>");
    }
    if (small) return launch_conv16<MMA, 1, 1, 32>(p, st, p.ksplit > 1 ? "conv16<64x64,k32,ksplit>" : "conv16<64x64,k32>");
    return launch_conv16<MMA, 2, 2, 32>(p, st, "conv16<128x128,k32>");
}

int run_conv16(int mma, const P16& p, hipStream_t st) {
    if (mma == CTGAN_MMA_BF16) return dispatch_conv16<CTGAN_MMA_BF16>(p, st);
    if (mma == CTGAN_MMA_F16) return dispatch_conv16<CTGAN_MMA_F16>(p, st);
    return dispatch_conv16<CTGAN_MMA_F32X3>(p, st);
}

template <int MMA, int TM, int TN>
int launch_wgrad16(const W16& p, int splits, hipStream_t st, const char* name) {
    constexpr int BMC = 2 * TM * 32, BNK = 2 * TN * 32;
    constexpr int bkp = planes<MMA>() == 3 ? 32 : 64;
    constexpr size_t lds = (size_t)((wgrad16_one_wave<MMA, TM, TN>() || wgrad16_single_stage<MMA, TM, TN>()) ? 1 : 2) * planes<MMA>() * (BMC + BNK) * (bkp + 8) * 2;      // the one-wave tiles keep one stage in a static array
    auto kern = p.relu_x ? wgrad16_kernel<MMA, TM, TN, true> : wgrad16_kernel<MMA, TM, TN, false>;
    static bool attrs[2] = {false, false};
    bool& attr = attrs[p.relu_x ? 1 : 0];
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "wgrad16: cannot reserve %zu B of LDS", lds);
        attr = true;
    }
    const int tiles_m = p.R * p.S * (p.C / BMC), tiles_n = (p.Ng + BNK - 1) / BNK;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n), (unsigned)splits), dim3(256), lds, st, p);
    ctgan_set_last_kernel(name);
    ctgan_set_last_symbol("wgrad16_kernel<%d, %d, %d, %s>", MMA, TM, TN, p.relu_x ? "true" : "false");
    return ctgan_check_launch(name);
}

struct WPlan16 { int bmc, bnk, tiles, splits, chunk; };
bool pow2_grid(const ctgan_conv_desc* d) { const int pq = d->P * d->Q; return !(pq & (pq - 1)) && !(d->Q & (d->Q - 1)); }
// the split mode has the 128x128 tile only (three planes per operand: one 60 KB stage), for power-of-two pixel grids
bool shape_ok_wgrad_x3(const ctgan_conv_desc* d) { return d->C % 128 == 0 && d->K % 128 == 0 && pow2_grid(d); }

WPlan16 wgrad16_plan(const ctgan_conv_desc* d, int mma) {
    WPlan16 w;
    w.bmc = d->C % 128 == 0 ? 128 : 64;
    w.bnk = d->K % 128 == 0 ? 128 : 64;
    const int wide_env = 2;
    const int wide = mma == CTGAN_MMA_F32X3 ? 0 : wide_env;
    const int pq = d->P * d->Q;
    // 4x2 accumulators per wave (256 x 128): 0.75 operand bytes per MFMA of the 128x128 tile
    if (wide && d->C % 256 == 0 && d->K % 128 == 0 && !(pq & (pq - 1)) && !(d->Q & (d->Q - 1))) w.bmc = 256;
    // 4x4 accumulators per wave (256 x 256 per workgroup): the kernel is bound by the bytes each CU can pull through its L1 miss
    // path (~35 GB/s per CU, ~9 TB/s chip-wide, whatever the tile): 15 B per kFLOP instead of 23 (256x128) / 30 (128x128)
    if (wide >= 2 && w.bmc == 256 && d->K % 256 == 0) w.bnk = 256;
    w.tiles = d->R * d->S * (d->C / w.bmc) * ((d->K + w.bnk - 1) / w.bnk);
    const int Kg = d->N * d->P * d->Q;
    // 512 workgroups are resident at a time (2 per CU: 73 KB of LDS each).  Pick the split count whose LAST round of workgroups
    // is the fullest (576 tiles unsplit would run as 512 + 64: 44 % of the second round idle), preferring fewer splits (less
    // slab traffic) among equals; at least 4 slices (256 pixels) per split.
    const int max_s = (Kg + 255) / 256;
    int s = 1;
    double best = -1.;
    const int resident = w.bmc == 256 ? 256 : 512;      // the 110 KB wide tile: one workgroup per CU
    for (int cand = 1; cand <= max_s && cand <= 64 && (long long)cand * w.tiles <= 8 * resident; ++cand) {
        const long long blocks = (long long)cand * w.tiles;
        const long long rounds = (blocks + resident - 1) / resident;
        const double eff = (double)blocks / (double)(rounds * resident);
        if (eff > best + 0.03) { best = eff; s = cand; }
    }
    int ch = (Kg + s - 1) / s;
    ch = ((ch + 63) / 64) * 64;
    w.splits = (Kg + ch - 1) / ch;
    w.chunk = ch;
    return w;
}

}  // namespace

// The kernels address their operands with 32-bit byte offsets through buffer descriptors: every operand a launch gathers from must
// span < 4 GiB.  Part of the routing predicates (not only of the launchers), so that a caller that asks "does the 16-bit family take
// this launch" never gets a yes followed by CTGAN_E_UNSUPPORTED.
static bool extents_ok(const ctgan_conv_desc* d, int op, int mma) {
    const long long lim = 1LL << 32;
    const long long x_extent = (long long)(d->N - 1) * d->xs[0] + (long long)(d->H - 1) * d->xs[2] + (long long)(d->W - 1) * d->xs[3] + d->C;
    const long long y_extent = (long long)(d->N - 1) * d->ys[0] + (long long)(d->P - 1) * d->ys[2] + (long long)(d->Q - 1) * d->ys[3] + d->K;
    const long long w_bytes = (long long)d->R * d->S * d->C * d->K * 2 * mma_planes(mma);
    if (op == CTGAN_CONV_FWD) return x_extent * 4 < lim && w_bytes < lim;
    if (op == CTGAN_CONV_DGRAD) return y_extent * 4 < lim && w_bytes < lim;
    return x_extent * 4 < lim && (long long)d->N * d->P * d->Q * d->K * 4 < lim;
}

extern "C" {

void ctgan_debug_x3_halo_version(int version) { g_halo_version_override = version; }
void ctgan_debug_x3_s2halo(int on) { g_s2halo = on ? 1 : 0; }
void ctgan_debug_x3_s2fwd(int on) { g_s2fwd = on ? 1 : 0; g_sf_ksplit = on == 2 ? 0 : 1; }
void ctgan_debug_x3_s2dgrad_sf(int on) { g_s2dgrad_sf = on; }
void ctgan_debug_x3_hk(int mode, int max_wgs) { g_hk = mode; if (max_wgs > 0) g_hk_max_wgs = max_wgs; }
static thread_local int g_last_group_kinds = 0;
static thread_local unsigned g_last_group_col_mask = 0;
int ctgan_debug_last_wgrad_group_kinds(void) { return g_last_group_kinds; }
unsigned ctgan_debug_last_wgrad_group_col_mask(void) { return g_last_group_col_mask; }

int ctgan_conv2d16_supported(const ctgan_conv_desc* d, int op, int mma) {
    if (!d || !mma_ok(mma)) return 0;
    if (op != CTGAN_CONV_FWD && op != CTGAN_CONV_DGRAD && op != CTGAN_CONV_WGRAD) return 0;
    if (!extents_ok(d, op, mma)) return 0;
    if (op == CTGAN_CONV_FWD) return shape_ok_fwd(d) ? 1 : 0;
    if (op == CTGAN_CONV_DGRAD) return shape_ok_dgrad(d) ? 1 : 0;
    return (shape_ok_wgrad(d) && (mma != CTGAN_MMA_F32X3 || shape_ok_wgrad_x3(d))) ? 1 : 0;
}

int ctgan_conv2d16_wgrad_col_takes(const ctgan_conv_desc* d, int mma, int32_t rows) {
    return (d && rows > 0 && mma_ok(mma) && ctgan_wgrad16c_takes(d, mma, rows)) ? 1 : 0;
}

static long long x3_wgrad_min_pixels() { return 32768LL; }      // routing threshold of weight gradients launched at request time

int ctgan_conv2d16_x3_prefers(const ctgan_conv_desc* d, int op) {
    // 1 for the launches on which CTGAN_MMA_F32X3 is measured faster than the fp32 MFMA family (tools/conv16_bench.py): stride-1 layers
    // the halo-patch kernel takes (170-200 vs 110-125 TFLOP/s), stride-2 layers on the slice kernel, large weight gradients - each only
    // when the launch fills the chip
    if (!d) return 0;
    if (op == CTGAN_CONV_WGRAD) {
        // weight gradient (wgrad16x3<128x128>): 145-167 vs 108-122 TFLOP/s from 32k pixels up; below that the fp32 family's grouped
        // multi-segment launch keeps the layer
        return (ctgan_conv2d16_supported(d, op, CTGAN_MMA_F32X3) && (long long)d->N * d->P * d->Q >= x3_wgrad_min_pixels()) ? 1 : 0;
    }
    if (op != CTGAN_CONV_FWD && op != CTGAN_CONV_DGRAD) return 0;
    if (op == CTGAN_CONV_FWD ? !shape_ok_fwd(d) : !shape_ok_dgrad(d)) return 0;
    if (!extents_ok(d, op, CTGAN_MMA_F32X3)) return 0;
    if (d->stride == 2) {
        // stride-2 layers (the folded ConvMeanPool / UpsampleConv filters) on the single-stage slice kernel: 145 / 160 vs 125 / 113
        // TFLOP/s (forward / four-phase data gradient) when the launch has >= 192 tiles of 128x128
        const long long M = op == CTGAN_CONV_FWD ? (long long)d->N * d->P * d->Q : (long long)d->N * (d->H / 2) * (d->W / 2);
        const int Ng = op == CTGAN_CONV_FWD ? d->K : d->C, Cg = op == CTGAN_CONV_FWD ? d->C : d->K;
        const int nph = op == CTGAN_CONV_FWD ? 1 : 4;
        if (Ng % 128 != 0 || Cg % 32 != 0) return 0;
        const long long t128 = nph * ((M + 127) / 128) * (Ng / 128);
        if (s2_halo_shape(d) && x3_s2halo()) {
            // the stride-2 halo kernel (conv16s2.h; tools/conv16_bench.py f32x3 s2 against f32): the four-phase data gradient on 8- / 16-wide
            // dy grids (155-196 against 88-117 TFLOP/s from 128 rows up)
            // (from 192 tiles of 128x128, as the slice kernel before it: 96 measured 13.61 against 13.58 ms per iteration)
            const int pq = (d->H / 2) * (d->W / 2), q = d->W / 2;
            if (op == CTGAN_CONV_DGRAD && pq % 32 == 0 && (q == 8 || q == 16) && t128 >= 192) return 1;
        }
        // the slice kernel: forward 130-146 against 101-125 TFLOP/s from 128 tiles of 128x128, four-phase data gradient from 192
        // (forward from 96 tiles: the 64-row launches of the penalty's double backward on 32x32 inputs - 13.60 against 13.67 ms per iteration
        // with 192 - and the 192-row trunk forward on 16x16 inputs, 113 against 96-102 TFLOP/s)
        return t128 >= (op == CTGAN_CONV_FWD ? 96 : 192) ? 1 : 0;
    }
    P16 p{};
    p.nph = 1; p.stride = d->stride;
    p.ph_T[0] = d->R; p.ph_U[0] = d->S;
    if (op == CTGAN_CONV_FWD) { p.Ng = d->K; p.C = d->C; p.P = d->P; p.Q = d->Q; p.M = d->N * d->P * d->Q; }
    else { p.Ng = d->C; p.C = d->K; p.P = d->H; p.Q = d->W; p.M = d->N * d->H * d->W; }
    if (halo_version() != 1 && frag_image_shape(d, op)) {
        // the fragment-streaming halo kernel has 64- and 32-pixel tiles: the 16x16 / 8x8 layers at 64-192 rows qualify too
        P16 q = p;
        q.Wf = reinterpret_cast<const unsigned short*>(d);      // (any non-null value: only the shape matters here)
        if (conv16x3hk_takes(q, nullptr)) return 1;          // (incl. the 64-row launches of the penalty's double backward, which the pixel-tiled kernels leave to the fp32 pipe)
        if (conv16x3hf_wins(q)) return 1;
    }
    if (!conv16x3h_ok(p, nullptr)) return 0;
    return (long long)(p.M / 128) * (p.Ng / 128) >= 192 ? 1 : 0;
}

static bool frag_image(const ctgan_conv_desc* d, int op, int mma) { return mma == CTGAN_MMA_F32X3 && frag_image_shape(d, op); }

size_t ctgan_conv2d16_filter_elems(const ctgan_conv_desc* d, int op, int mma) {
    if (!d || !mma_ok(mma)) return 0;
    // both layouts hold every tap exactly once (no zero-padded phases); the FRAG copy doubles the split-mode image of the shapes that have one
    return (size_t)mma_planes(mma) * d->R * d->S * d->C * d->K * (frag_image(d, op, mma) ? 2 : 1);
}

int ctgan_conv2d16_pack_filter(const ctgan_conv_desc* d, int op, int mma, const float* w, void* wp, ctgan_stream_t stream) {
    if (!d || !w || !wp || !mma_ok(mma)) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_pack_filter: bad argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned short* out = (unsigned short*)wp;
    const long long plane = (long long)d->R * d->S * d->C * d->K;
    if (op == CTGAN_CONV_FWD) {
        if (d->C % 2) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_pack_filter: odd channel count");
        const long long total = (long long)d->R * d->S * d->C / 2 * d->K;
        const int frag = frag_image(d, op, mma) ? 1 : 0;
        if (((long long)d->R * d->S * d->C / 2) % 32 == 0 && d->K % 32 == 0) {
            // the transposing pack through 32 x 32 LDS tiles (pack_batch_kernel's forward branch) as a batch of one: the straight kernel below
            // reads one cache line per lane - 19.5 us for config[1]'s 5x5x256x512 filter (13 MB in, 6.5 MB out), once per weight version
            PackJobs jobs{};
            PackJob& jb = jobs.j[0];
            jb.w = w; jb.wp = out; jb.op = CTGAN_CONV_FWD; jb.plane = plane;
            jb.pp.R = d->R; jb.pp.S = d->S; jb.pp.C = d->C; jb.pp.K = d->K; jb.pp.nph = 1; jb.pp.frag = frag;
            const dim3 grid(ctgan_blocks(total / 4, 256, 2048), 1);
            if (mma == CTGAN_MMA_BF16) hipLaunchKernelGGL(pack_batch_kernel<CTGAN_MMA_BF16>, grid, dim3(256), 0, st, jobs);
            else if (mma == CTGAN_MMA_F16) hipLaunchKernelGGL(pack_batch_kernel<CTGAN_MMA_F16>, grid, dim3(256), 0, st, jobs);
            else hipLaunchKernelGGL(pack_batch_kernel<CTGAN_MMA_F32X3>, grid, dim3(256), 0, st, jobs);
            return ctgan_check_launch("pack16_fwd");
        }
        if (mma == CTGAN_MMA_BF16) hipLaunchKernelGGL(pack_fwd_kernel<CTGAN_MMA_BF16>, dim3(ctgan_blocks(total, 256)), dim3(256), 0, st, w, out, d->R * d->S, d->C, d->K, plane, 0);
        else if (mma == CTGAN_MMA_F16) hipLaunchKernelGGL(pack_fwd_kernel<CTGAN_MMA_F16>, dim3(ctgan_blocks(total, 256)), dim3(256), 0, st, w, out, d->R * d->S, d->C, d->K, plane, 0);
        else hipLaunchKernelGGL(pack_fwd_kernel<CTGAN_MMA_F32X3>, dim3(ctgan_blocks(total, 256)), dim3(256), 0, st, w, out, d->R * d->S, d->C, d->K, plane, frag);
        return ctgan_check_launch("pack16_fwd");
    }
    if (op == CTGAN_CONV_DGRAD) {
        if (!shape_ok_dgrad(d)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_pack_filter: data-gradient shape outside the 16-bit family");
        const PhaseGeom g = phase_geom(d);
        PackPhases pp{};
        for (int a = 0; a < 2; ++a) { pp.T[a] = g.T[a]; pp.U[a] = g.U[a]; pp.r0[a] = g.r0[a]; pp.s0[a] = g.s0[a]; }
        pp.nph = g.nph; pp.step = g.step; pp.R = d->R; pp.S = d->S; pp.C = d->C; pp.K = d->K;
        pp.frag = frag_image(d, op, mma) ? 1 : 0;
        long long most = 0;
        for (int ph = 0; ph < g.nph; ++ph) {
            const long long n = (long long)d->C * g.T[ph >> 1] * g.U[ph & 1] * d->K / 2;
            if (n > most) most = n;
        }
        if (mma == CTGAN_MMA_BF16) hipLaunchKernelGGL(pack_dgrad_kernel<CTGAN_MMA_BF16>, dim3(ctgan_blocks(most, 256), g.nph), dim3(256), 0, st, w, out, pp, plane);
        else if (mma == CTGAN_MMA_F16) hipLaunchKernelGGL(pack_dgrad_kernel<CTGAN_MMA_F16>, dim3(ctgan_blocks(most, 256), g.nph), dim3(256), 0, st, w, out, pp, plane);
        else hipLaunchKernelGGL(pack_dgrad_kernel<CTGAN_MMA_F32X3>, dim3(ctgan_blocks(most, 256), g.nph), dim3(256), 0, st, w, out, pp, plane);
        return ctgan_check_launch("pack16_dgrad");
    }
    return ctgan_fail(CTGAN_E_BADARG, "conv2d16_pack_filter: op %d", op);
}

int ctgan_conv2d16_pack_batch(const ctgan_conv_desc* descs, const int32_t* ops, int32_t n, int mma, const float* const* ws, void* const* wps,
                              ctgan_stream_t stream) {
    if (!descs || !ops || !ws || !wps || n <= 0 || !mma_ok(mma)) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_pack_batch: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += CTGAN_PACK_BATCH) {
        PackJobs jobs{};
        const int cnt = n - base < CTGAN_PACK_BATCH ? n - base : CTGAN_PACK_BATCH;
        long long most = 0;
        for (int i = 0; i < cnt; ++i) {
            const ctgan_conv_desc* d = descs + base + i;
            PackJob& jb = jobs.j[i];
            jb.w = ws[base + i]; jb.wp = (unsigned short*)wps[base + i]; jb.op = ops[base + i];
            if (!jb.w || !jb.wp) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_pack_batch: null filter");
            jb.plane = (long long)d->R * d->S * d->C * d->K;
            PackPhases& pp = jb.pp;
            pp.R = d->R; pp.S = d->S; pp.C = d->C; pp.K = d->K;
            pp.frag = frag_image(d, jb.op, mma) ? 1 : 0;
            long long work;
            if (jb.op == CTGAN_CONV_FWD) {
                if (d->C % 2) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_pack_batch: odd channel count");
                pp.nph = 1; work = jb.plane / 2;
            } else if (jb.op == CTGAN_CONV_DGRAD) {
                if (!shape_ok_dgrad(d)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_pack_batch: data-gradient shape outside the 16-bit family");
                const PhaseGeom g = phase_geom(d);
                for (int a = 0; a < 2; ++a) { pp.T[a] = g.T[a]; pp.U[a] = g.U[a]; pp.r0[a] = g.r0[a]; pp.s0[a] = g.s0[a]; }
                pp.nph = g.nph; pp.step = g.step;
                work = 0;
                for (int ph = 0; ph < g.nph; ++ph) {
                    const long long m = (long long)d->C * g.T[ph >> 1] * g.U[ph & 1] * d->K / 2;
                    if (m > work) work = m;
                }
            } else {
                return ctgan_fail(CTGAN_E_BADARG, "conv2d16_pack_batch: op %d", jb.op);
            }
            if (work > most) most = work;
        }
        const dim3 grid(ctgan_blocks(most, 256, 64), (unsigned)cnt);
        if (mma == CTGAN_MMA_BF16) hipLaunchKernelGGL(pack_batch_kernel<CTGAN_MMA_BF16>, grid, dim3(256), 0, st, jobs);
        else if (mma == CTGAN_MMA_F16) hipLaunchKernelGGL(pack_batch_kernel<CTGAN_MMA_F16>, grid, dim3(256), 0, st, jobs);
        else hipLaunchKernelGGL(pack_batch_kernel<CTGAN_MMA_F32X3>, grid, dim3(256), 0, st, jobs);
        const int rc = ctgan_check_launch("pack16_batch");
        if (rc) return rc;
    }
    return 0;
}

size_t ctgan_conv2d16_workspace_bytes(const ctgan_conv_desc* d, int op) {
    // K-split slabs of the forward / data gradient (only launches whose tiles cannot fill the chip use them): 8 splits at most
    if (!d || (op != CTGAN_CONV_FWD && op != CTGAN_CONV_DGRAD)) return 0;
    const long long rows = op == CTGAN_CONV_FWD ? (long long)d->N * d->P * d->Q : (long long)d->N * d->H * d->W;
    const long long cols = op == CTGAN_CONV_FWD ? d->K : d->C;
    if (rows * cols > (1LL << 22)) return 0;                 // >= 256 tiles of 128x128: never split
    return (size_t)8 * rows * cols * sizeof(float);
}

// ctgan_epilogue_ext::act on the slice kernels: fills the act / dropout fields of p for a dense channels-last result of `sample_elems`
// elements per sample.  Only the bf16 / fp16 modes (the split mode routes launches to the halo-patch kernels, which have no such epilogue).
static int conv16_set_act(P16& p, const ctgan_epilogue_ext* ext, int mma, bool dense, long long sample_elems, const char* who) {
    if (mma == CTGAN_MMA_F32X3 || !dense || ext->out_mask)
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "%s: fused LeakyReLU + dropout needs a 16-bit mode and a dense channels-last result", who);
    if (ext->n_ranges < 0 || ext->n_ranges > CTGAN_DROP_RANGES) return ctgan_fail(CTGAN_E_UNSUPPORTED, "%s: more than %d sample ranges", who, CTGAN_DROP_RANGES);
    if (ext->act_ref && (reinterpret_cast<uintptr_t>(ext->act_ref) & 15)) return ctgan_fail(CTGAN_E_BADARG, "%s: act_ref must be 16-byte aligned", who);
    p.act = 1; p.act_alpha = ext->act_alpha; p.act_ref = ext->act_ref;
    p.drop_seed = ext->drop_seed; p.drop_ctr = reinterpret_cast<const unsigned long long*>(ext->drop_ctr);
    p.drop_keep = 1.f; p.drop_sid = 0; p.drop_nr = ext->n_ranges;
    for (int i = 0; i < CTGAN_DROP_RANGES; ++i) { p.drop_nend[i] = 0x7fffffff; p.drop_rkeep[i] = 1.f; p.drop_rsid[i] = 0; p.drop_roff[i] = 0; }
    if (ext->n_ranges > 0) {
        long long start = 0;
        for (int i = 0; i < ext->n_ranges; ++i) {
            if (ext->range_end[i] < start) return ctgan_fail(CTGAN_E_BADARG, "%s: sample ranges must ascend", who);
            p.drop_nend[i] = ext->range_end[i];
            p.drop_rkeep[i] = (ext->range_keep[i] > 0.f && ext->range_keep[i] < 1.f) ? ext->range_keep[i] : 1.f;
            p.drop_rsid[i] = (unsigned)ext->range_stream_id[i];
            p.drop_roff[i] = start * sample_elems;
            start = ext->range_end[i];
        }
    } else if (ext->drop_keep > 0.f && ext->drop_keep < 1.f) {
        p.drop_keep = ext->drop_keep; p.drop_sid = (unsigned)ext->drop_stream_id;
    }
    return CTGAN_OK;
}

static int conv2d16_fwd_impl(const ctgan_conv_desc* d, int mma, const float* x, const void* wp, const float* bias, const float* resid,
                             float* y, int flags, const ctgan_epilogue_ext* ext, void* ws, size_t ws_bytes, ctgan_stream_t stream) {
    if (!d || !x || !wp || !y || !mma_ok(mma)) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_fwd: bad argument");
    if (!shape_ok_fwd(d)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd: shape outside the 16-bit family");
    if (ext && ext->out_tanh) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd_ex: the tanh epilogue exists in the many -> few pixel kernel only");
    const long long x_extent = (long long)(d->N - 1) * d->xs[0] + (long long)(d->H - 1) * d->xs[2] + (long long)(d->W - 1) * d->xs[3] + d->C;
    const long long w_plane = (long long)d->R * d->S * d->C * d->K * 2;
    if (x_extent * 4 >= (1LL << 32) || w_plane * mma_planes(mma) >= (1LL << 32))
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd: operand exceeds the 4 GiB buffer range");
    P16 p{};
    p.X = x; p.Wp = (const unsigned short*)wp; p.bias = bias; p.resid = resid; p.mask = ext ? ext->out_mask : nullptr; p.D = y;
    p.H = d->H; p.W = d->W; p.P = d->P; p.Q = d->Q; p.C = d->C; p.stride = d->stride;
    p.s_n = d->xs[0]; p.s_h = d->xs[2]; p.s_w = d->xs[3];
    p.M = d->N * d->P * d->Q; p.Ng = d->K;
    p.ds_n = d->ys[0]; p.ds_p = d->ys[2]; p.ds_q = d->ys[3];
    p.relu = (flags & CTGAN_EPI_RELU) ? 1 : 0; p.relu_in = (flags & CTGAN_IN_RELU) ? 1 : 0;
    p.resid_up = (resid && (flags & CTGAN_RESID_UP)) ? 1 : 0;
    p.x_bytes = (unsigned)(x_extent * 4); p.w_plane_bytes = (unsigned)w_plane; p.w_bytes = (unsigned)(w_plane * mma_planes(mma));
    if (frag_image(d, CTGAN_CONV_FWD, mma)) { p.Wf = p.Wp + 3 * (w_plane / 2); p.wf_bytes = (unsigned)(3 * w_plane); }
    p.nph = 1;
    p.slab = (float*)ws; p.slab_bytes = ws ? ws_bytes : 0;
    p.ph_T[0] = d->R; p.ph_U[0] = d->S; p.ph_pad_t[0] = d->pad_t; p.ph_pad_l[0] = d->pad_l;
    p.ph_T[1] = d->R; p.ph_U[1] = d->S; p.ph_pad_t[1] = d->pad_t; p.ph_pad_l[1] = d->pad_l;
    hipStream_t st = (hipStream_t)stream;
    if (ext && ext->in_bn_mean) {                            // batch norm of the input while the halo patch is staged (split mode, stride 1)
        if (mma != CTGAN_MMA_F32X3 || d->stride != 1 || !ext->in_bn_rstd || !ext->in_bn_scale || !ext->in_bn_offset || ext->in_bn_groups <= 0 ||
            d->N % ext->in_bn_groups || ext->act || ext->out_mask || ext->n_ranges > 0 || (ext->drop_keep > 0.f && ext->drop_keep < 1.f) ||
            ((reinterpret_cast<uintptr_t>(ext->in_bn_mean) | reinterpret_cast<uintptr_t>(ext->in_bn_rstd) | reinterpret_cast<uintptr_t>(ext->in_bn_scale) |
              reinterpret_cast<uintptr_t>(ext->in_bn_offset)) & 15))
            return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd_ex: input batch norm on load: split mode, stride 1, no other epilogue extension");
        p.bn_mean = ext->in_bn_mean; p.bn_rstd = ext->in_bn_rstd; p.bn_scale = ext->in_bn_scale; p.bn_offset = ext->in_bn_offset;
        p.bn_labels = ext->in_bn_labels; p.bn_per = d->N / ext->in_bn_groups;
        return launch_conv16x3h(p, st);
    }
    if (ext && ext->act) {                                   // the LeakyReLU + dropout pair in the slice kernels' epilogue
        const bool dense = d->ys[1] == 1 && d->ys[3] == d->K && d->ys[2] == (int64_t)d->Q * d->K && d->ys[0] == (int64_t)d->P * d->Q * d->K;
        if (resid || p.resid_up) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd_ex: fused LeakyReLU + dropout with a residual");
        if (const int rc = conv16_set_act(p, ext, mma, dense, (long long)d->P * d->Q * d->K, "conv2d16_fwd_ex")) return rc;
        return run_conv16(mma, p, st);
    }
    const bool ranged = ext && ext->n_ranges > 0;
    const bool want_drop = ext && (ranged || (ext->drop_keep > 0.f && ext->drop_keep < 1.f));
    if (want_drop) {                                         // only the halo-patch kernel has the dropout epilogue
        const bool dense = d->ys[1] == 1 && d->ys[3] == d->K && d->ys[2] == (int64_t)d->Q * d->K && d->ys[0] == (int64_t)d->P * d->Q * d->K;
        if (mma != CTGAN_MMA_F32X3 || !halo_takes(p) || (ranged && (!dense || ext->n_ranges > CTGAN_DROP_RANGES)))
            return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd_ex: epilogue dropout outside the halo-patch form");
        p.drop = 1; p.drop_keep = 1.f; p.drop_seed = ext->drop_seed; p.drop_sid = 0;
        p.drop_ctr = reinterpret_cast<const unsigned long long*>(ext->drop_ctr);
        for (int i = 0; i < CTGAN_DROP_RANGES; ++i) { p.drop_mend[i] = 0x7fffffff; p.drop_rkeep[i] = 1.f; p.drop_rsid[i] = 0; p.drop_roff[i] = 0; }
        if (ranged) {
            p.drop_nr = ext->n_ranges;
            long long start = 0;
            for (int i = 0; i < p.drop_nr; ++i) {
                if (((long long)ext->range_end[i] * d->P * d->Q) % 128)
                    return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd_ex: dropout row ranges must start at multiples of 128 output pixels");
                p.drop_mend[i] = (int)((long long)ext->range_end[i] * d->P * d->Q);
                p.drop_rkeep[i] = (ext->range_keep[i] > 0.f && ext->range_keep[i] < 1.f) ? ext->range_keep[i] : 1.f;
                p.drop_rsid[i] = (unsigned)ext->range_stream_id[i];
                p.drop_roff[i] = start * (long long)d->P * d->Q * d->K;
                start = ext->range_end[i];
            }
        } else {
            p.drop_keep = ext->drop_keep; p.drop_sid = (unsigned)ext->drop_stream_id;
        }
        return launch_conv16x3h(p, st);
    }
    if (p.resid_up) {                                        // only the halo-patch kernel reads the residual through the upsample
        if (mma != CTGAN_MMA_F32X3 || ((d->P | d->Q) & 1) || !halo_takes(p))
            return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_fwd: CTGAN_RESID_UP outside the halo-patch form");
        return launch_conv16x3h(p, st);
    }
    return run_conv16(mma, p, st);
}

int ctgan_conv2d16_fwd(const ctgan_conv_desc* d, int mma, const float* x, const void* wp, const float* bias, const float* resid,
                       float* y, int flags, void* ws, size_t ws_bytes, ctgan_stream_t stream) {
    return conv2d16_fwd_impl(d, mma, x, wp, bias, resid, y, flags, nullptr, ws, ws_bytes, stream);
}
int ctgan_conv2d16_fwd_ex(const ctgan_conv_desc* d, int mma, const float* x, const void* wp, const float* bias, const float* resid,
                          float* y, int flags, const ctgan_epilogue_ext* ext, void* ws, size_t ws_bytes, ctgan_stream_t stream) {
    return conv2d16_fwd_impl(d, mma, x, wp, bias, resid, y, flags, ext, ws, ws_bytes, stream);
}

int ctgan_conv2d16_dgrad(const ctgan_conv_desc* d, int mma, const float* dy, const void* wp, const float* bias, const float* mask,
                         const float* resid, float* dx, int flags, void* ws, size_t ws_bytes, ctgan_stream_t stream) {
    return ctgan_conv2d16_dgrad_ex(d, mma, dy, wp, bias, mask, resid, dx, flags, nullptr, ws, ws_bytes, stream);
}

int ctgan_conv2d16_dgrad_ex(const ctgan_conv_desc* d, int mma, const float* dy, const void* wp, const float* bias, const float* mask,
                            const float* resid, float* dx, int flags, const ctgan_epilogue_ext* ext, void* ws, size_t ws_bytes,
                            ctgan_stream_t stream) {
    if (!d || !dy || !wp || !dx || !mma_ok(mma)) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_dgrad: bad argument");
    if (!shape_ok_dgrad(d)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_dgrad: shape outside the 16-bit family");
    const long long y_extent = (long long)(d->N - 1) * d->ys[0] + (long long)(d->P - 1) * d->ys[2] + (long long)(d->Q - 1) * d->ys[3] + d->K;
    const long long w_plane = (long long)d->R * d->S * d->C * d->K * 2;
    if (y_extent * 4 >= (1LL << 32) || w_plane * mma_planes(mma) >= (1LL << 32))
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_dgrad: operand exceeds the 4 GiB buffer range");
    const PhaseGeom g = phase_geom(d);
    P16 p{};
    p.X = dy; p.Wp = (const unsigned short*)wp; p.bias = bias; p.mask = mask; p.resid = resid; p.D = dx;
    p.H = d->P; p.W = d->Q;                                  // the gather runs over dy
    p.C = d->K; p.stride = 1;
    p.s_n = d->ys[0]; p.s_h = d->ys[2]; p.s_w = d->ys[3];
    p.Ng = d->C;
    p.relu = (flags & CTGAN_EPI_RELU) ? 1 : 0; p.relu_in = 0;
    p.x_bytes = (unsigned)(y_extent * 4); p.w_plane_bytes = (unsigned)w_plane; p.w_bytes = (unsigned)(w_plane * mma_planes(mma));
    if (frag_image(d, CTGAN_CONV_DGRAD, mma)) { p.Wf = p.Wp + 3 * (w_plane / 2); p.wf_bytes = (unsigned)(3 * w_plane); }
    p.nph = g.nph;
    p.slab = (float*)ws; p.slab_bytes = ws ? ws_bytes : 0;
    for (int a = 0; a < 2; ++a) { p.ph_T[a] = g.T[a]; p.ph_U[a] = g.U[a]; p.ph_pad_t[a] = g.pad_t[a]; p.ph_pad_l[a] = g.pad_l[a]; }
    for (int ph = 0; ph < 4; ++ph) p.ph_w_off[ph] = ph < g.nph ? phase_off(g, d, ph) : 0;
    if (g.nph == 4) {
        p.P = d->H / 2; p.Q = d->W / 2;
        p.ds_n = d->xs[0]; p.ds_p = 2 * d->xs[2]; p.ds_q = 2 * d->xs[3];
        p.ph_d_h = d->xs[2]; p.ph_d_w = d->xs[3];
    } else {
        p.P = d->H; p.Q = d->W;
        p.ds_n = d->xs[0]; p.ds_p = d->xs[2]; p.ds_q = d->xs[3];
        p.ph_T[1] = p.ph_T[0]; p.ph_U[1] = p.ph_U[0]; p.ph_pad_t[1] = p.ph_pad_t[0]; p.ph_pad_l[1] = p.ph_pad_l[0];
    }
    p.M = d->N * p.P * p.Q;
    hipStream_t st = (hipStream_t)stream;
    if (ext && ext->act) {                                   // the pair's backward (act_ref = the forward result) in the slice kernels' epilogue
        const bool dense = d->xs[1] == 1 && d->xs[3] == d->C && d->xs[2] == (int64_t)d->W * d->C && d->xs[0] == (int64_t)d->H * d->W * d->C;
        if (const int rc = conv16_set_act(p, ext, mma, dense, (long long)d->H * d->W * d->C, "conv2d16_dgrad_ex")) return rc;
        return run_conv16(mma, p, st);
    }
    const bool ranged = ext && ext->n_ranges > 0;
    bool range_drop = false;
    if (ranged)
        for (int i = 0; i < ext->n_ranges && i < CTGAN_DROP_RANGES; ++i) range_drop = range_drop || (ext->range_keep[i] > 0.f && ext->range_keep[i] < 1.f);
    if (range_drop || (ext && !ranged && ext->drop_keep > 0.f && ext->drop_keep < 1.f)) {
        // the data gradient multiplied by a dropout mask (the mask of the dropout whose result the forward conv consumed), as
        // ctgan_conv2d_dgrad_ex: only the halo-patch kernels have the dropout epilogue, on a dense channels-last dx
        const bool dense = d->xs[1] == 1 && d->xs[3] == d->C && d->xs[2] == (int64_t)d->W * d->C && d->xs[0] == (int64_t)d->H * d->W * d->C;
        if (mma != CTGAN_MMA_F32X3 || g.nph != 1 || !dense || !halo_takes(p))
            return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_dgrad_ex: epilogue dropout outside the halo-patch form");
        p.drop = 1; p.drop_keep = ranged ? 1.f : ext->drop_keep; p.drop_seed = ext->drop_seed; p.drop_sid = ranged ? 0u : (unsigned)ext->drop_stream_id;
        p.drop_ctr = reinterpret_cast<const unsigned long long*>(ext->drop_ctr);
        p.drop_nr = 0;
        for (int i = 0; i < CTGAN_DROP_RANGES; ++i) { p.drop_mend[i] = 0x7fffffff; p.drop_rkeep[i] = 1.f; p.drop_rsid[i] = 0; p.drop_roff[i] = 0; }
        if (ranged) {
            // sample ranges of dx, each with the mask of its own forward dropout (the merged backward of a critic step, round 5): as in the
            // forward, boundaries on 128-pixel tiles, Philox indices relative to the range's first element
            if (ext->n_ranges > CTGAN_DROP_RANGES) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_dgrad_ex: more than %d sample ranges", CTGAN_DROP_RANGES);
            p.drop_nr = ext->n_ranges;
            long long start = 0;
            for (int i = 0; i < p.drop_nr; ++i) {
                if (i + 1 < p.drop_nr && ((long long)ext->range_end[i] * d->H * d->W) % 128)
                    return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_dgrad_ex: dropout row ranges must start at multiples of 128 pixels");
                p.drop_mend[i] = (int)((long long)ext->range_end[i] * d->H * d->W);
                p.drop_rkeep[i] = (ext->range_keep[i] > 0.f && ext->range_keep[i] < 1.f) ? ext->range_keep[i] : 1.f;
                p.drop_rsid[i] = (unsigned)ext->range_stream_id[i];
                p.drop_roff[i] = start * (long long)d->H * d->W * d->C;
                start = ext->range_end[i];
            }
        }
        return launch_conv16x3h(p, st);
    }
    return run_conv16(mma, p, st);
}

// A single weight gradient on the filter-column kernel (wgrad16c.hip) where it takes the geometry: its pixels per split, 0 otherwise.
static int wgrad16_col_chunk(const ctgan_conv_desc* d, int mma) {
    if (!ctgan_wgrad16c_takes(d, mma, d->N)) return 0;
    ctgan_wc_problem w{};
    w.d = d; w.N = d->N;
    int chunk = 0;
    ctgan_wgrad16c_plan(&w, 1, mma, &chunk);
    return chunk;
}

size_t ctgan_conv2d16_wgrad_workspace_bytes(const ctgan_conv_desc* d, int mma) {
    // (sized for the bias row as well: with db the partial sums always go through the slabs, also when the plan has one split)
    if (!d || !ctgan_conv2d16_supported(d, CTGAN_CONV_WGRAD, mma)) return 0;
    if (const int chunk = wgrad16_col_chunk(d, mma)) {
        const int Kg = d->N * d->P * d->Q;
        return (size_t)((Kg + chunk - 1) / chunk) * ((size_t)d->R * d->S * d->C + 1) * d->K * sizeof(float);
    }
    const WPlan16 w = wgrad16_plan(d, mma);
    return (size_t)w.splits * ((size_t)d->R * d->S * d->C + 1) * d->K * sizeof(float);
}

int ctgan_conv2d16_wgrad(const ctgan_conv_desc* d, int mma, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                         int flags, ctgan_stream_t stream) {
    return ctgan_conv2d16_wgrad_bias(d, mma, x, dy, dw, nullptr, ws, ws_bytes, flags, stream);
}

int ctgan_conv2d16_wgrad_bias(const ctgan_conv_desc* d, int mma, const float* x, const float* dy, float* dw, float* db, void* ws,
                              size_t ws_bytes, int flags, ctgan_stream_t stream) {
    if (!d || !x || !dy || !dw || !mma_ok(mma)) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_wgrad: bad argument");
    if (db && (d->K % 4 || (reinterpret_cast<uintptr_t>(db) & 15)))
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_wgrad: the fused bias gradient needs K %% 4 == 0 and a 16-byte aligned db");
    if (!ctgan_conv2d16_supported(d, CTGAN_CONV_WGRAD, mma)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_wgrad: shape outside the 16-bit family");
    const long long x_extent = (long long)(d->N - 1) * d->xs[0] + (long long)(d->H - 1) * d->xs[2] + (long long)(d->W - 1) * d->xs[3] + d->C;
    const long long y_extent = (long long)d->N * d->P * d->Q * d->K;
    if (x_extent * 4 >= (1LL << 32) || y_extent * 4 >= (1LL << 32))
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_wgrad: operand exceeds the 4 GiB buffer range");
    hipStream_t st = (hipStream_t)stream;
    if (const int chunk = wgrad16_col_chunk(d, mma)) {
        // the filter-column kernel: always through slabs ([splits][R*S*C (+1)][K]) and the fixed-order reduction
        const int Kg = d->N * d->P * d->Q, splits = (Kg + chunk - 1) / chunk;
        const size_t rows = (size_t)d->R * d->S * d->C + (db ? 1 : 0);
        const size_t need = (size_t)splits * rows * d->K * sizeof(float);
        if (need > ws_bytes || !ws) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_wgrad: workspace %zu B < %zu B", ws_bytes, need);
        if (!(flags & CTGAN_WGRAD16_REDUCE_ONLY)) {
            ctgan_wc_problem w{};
            w.d = d; w.x = x; w.dy = dy; w.out = (float*)ws; w.N = d->N; w.relu_x = (flags & CTGAN_IN_RELU) ? 1 : 0; w.with_bias = db ? 1 : 0; w.chunk = chunk;
            const int rc = ctgan_wgrad16c_launch(&w, 1, mma, st);
            if (rc) return rc;
        }
        if (!(flags & CTGAN_WGRAD16_GEMM_ONLY)) {
            const long long n_main = (long long)d->R * d->S * d->C * d->K, n = n_main + (db ? d->K : 0);
            hipLaunchKernelGGL(reduce16_kernel, dim3(ctgan_blocks(n / 4, 256, 1 << 20)), dim3(256), 0, st, (const float*)ws, dw, db ? db : dw, n, n_main, splits);
            return ctgan_check_launch("reduce16");
        }
        return 0;
    }
    const WPlan16 w = wgrad16_plan(d, mma);
    const bool slabs = w.splits > 1 || db;
    const size_t need = slabs ? (size_t)w.splits * ((size_t)d->R * d->S * d->C + (db ? 1 : 0)) * d->K * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_wgrad: workspace %zu B < %zu B", ws_bytes, need);
    W16 p{};
    p.X = x; p.DY = dy; p.OUT = slabs ? (float*)ws : dw;
    p.with_bias = db ? 1 : 0;
    p.H = d->H; p.W = d->W; p.P = d->P; p.Q = d->Q; p.C = d->C; p.R = d->R; p.S = d->S; p.stride = d->stride;
    p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.s_n = d->xs[0]; p.s_h = d->xs[2]; p.s_w = d->xs[3];
    p.Mtot = d->R * d->S * d->C; p.Ng = d->K; p.Kg = d->N * d->P * d->Q;
    p.chunk = w.chunk; p.relu_x = (flags & CTGAN_IN_RELU) ? 1 : 0; p.dbg = dbg16();
    {
        const int pq = d->P * d->Q;
        const bool pow2 = !(pq & (pq - 1)) && !(d->Q & (d->Q - 1));
        p.pq_shift = pow2 ? __builtin_ctz(pq) : -1;
        p.q_shift = pow2 ? __builtin_ctz(d->Q) : -1;
    }
    p.x_bytes = (unsigned)(x_extent * 4); p.dy_bytes = (unsigned)(y_extent * 4);
    int rc = 0;
    const bool bf = mma == CTGAN_MMA_BF16;
    if (flags & CTGAN_WGRAD16_REDUCE_ONLY) { /* the slabs are already in ws (bench.py times the two launches apart) */ }
    else
    if (mma == CTGAN_MMA_F32X3) rc = launch_wgrad16<CTGAN_MMA_F32X3, 2, 2>(p, w.splits, st, "wgrad16x3<128x128>");
    else if (w.bmc == 256 && w.bnk == 256) rc = bf ? launch_wgrad16<CTGAN_MMA_BF16, 4, 4>(p, w.splits, st, "wgrad16<256x256>") : launch_wgrad16<CTGAN_MMA_F16, 4, 4>(p, w.splits, st, "wgrad16<256x256>");
    else if (w.bmc == 256) rc = bf ? launch_wgrad16<CTGAN_MMA_BF16, 4, 2>(p, w.splits, st, "wgrad16<256x128>") : launch_wgrad16<CTGAN_MMA_F16, 4, 2>(p, w.splits, st, "wgrad16<256x128>");
    else if (w.bmc == 128 && w.bnk == 128) rc = bf ? launch_wgrad16<CTGAN_MMA_BF16, 2, 2>(p, w.splits, st, "wgrad16<128x128>") : launch_wgrad16<CTGAN_MMA_F16, 2, 2>(p, w.splits, st, "wgrad16<128x128>");
    else if (w.bmc == 128) rc = bf ? launch_wgrad16<CTGAN_MMA_BF16, 2, 1>(p, w.splits, st, "wgrad16<128x64>") : launch_wgrad16<CTGAN_MMA_F16, 2, 1>(p, w.splits, st, "wgrad16<128x64>");
    else if (w.bnk == 128) rc = bf ? launch_wgrad16<CTGAN_MMA_BF16, 1, 2>(p, w.splits, st, "wgrad16<64x128>") : launch_wgrad16<CTGAN_MMA_F16, 1, 2>(p, w.splits, st, "wgrad16<64x128>");
    else rc = bf ? launch_wgrad16<CTGAN_MMA_BF16, 1, 1>(p, w.splits, st, "wgrad16<64x64>") : launch_wgrad16<CTGAN_MMA_F16, 1, 1>(p, w.splits, st, "wgrad16<64x64>");
    if (rc) return rc;
    if (slabs && !(flags & CTGAN_WGRAD16_GEMM_ONLY)) {
        const long long n_main = (long long)p.Mtot * p.Ng, n = n_main + (db ? p.Ng : 0);
        hipLaunchKernelGGL(reduce16_kernel, dim3(ctgan_blocks(n / 4, 256, 1 << 20)), dim3(256), 0, st, (const float*)ws, dw, db ? db : dw, n, n_main, w.splits);
        return ctgan_check_launch("reduce16");
    }
    return 0;
}

}  // extern "C"

// ---- grouped split-mode weight gradients -----------------------------------------------------------------------------------------
namespace {
struct G16Seg { W16 p; int tiles, splits; };
bool group16_member_ok(const ctgan_wgrad_group& G, int mma) {
    const ctgan_conv_desc* d = &G.d;
    if (G.nseg < 1 || G.nseg > CTGAN_WGRAD_MAX_SEGS || d->x_up || d->xs[1] != 1) return false;
    if (mma == CTGAN_MMA_F32X3 ? !shape_ok_wgrad_x3(d) : (d->C % 128 != 0 || d->K % 128 != 0 || d->Q % 4 != 0)) return false;
    if (d->ys[1] != 1 || d->ys[3] != d->K || d->ys[2] != (int64_t)d->Q * d->K || d->ys[0] != (int64_t)d->P * d->Q * d->K) return false;   // dense channels-last dy
    if (G.db && (d->K % 4 || (reinterpret_cast<uintptr_t>(G.db) & 15))) return false;
    for (int k = 0; k < G.nseg; ++k) {
        if (G.Ns[k] < 1) return false;
        const long long x_extent = (long long)(G.Ns[k] - 1) * d->xs[0] + (long long)(d->H - 1) * d->xs[2] + (long long)(d->W - 1) * d->xs[3] + d->C;
        if (x_extent * 4 >= (1LL << 32) || (long long)G.Ns[k] * d->P * d->Q * d->K * 4 >= (1LL << 32)) return false;
    }
    return true;
}
// The plan of a grouped call: pixels per split of every (problem, segment) and the slab workspace.  Problems the filter-column kernel
// takes (wgrad16c.hip: split mode, 8- / 16- / 32-pixel rows, stride 1 or 2) are planned by ctgan_wgrad16c_plan; the others ride the
// slice kernel's grouped launch with ONE chunk (pixels per split, multiple of 64) for all of them: workgroups of equal length, estimated
// time = rounds of 512 resident workgroups (two per CU) x (chunk pixels at ~70 ns each + a fixed prologue / 64 KB slab store); the
// smallest estimate wins, longer chunks (fewer slabs to write and reduce) among near-equals.  A function of the geometries and row counts
// only, so the workspace query and the launch agree.
struct G16Plan {
    size_t ws_bytes;
    bool col[CTGAN_WGRAD_GROUP_LIMIT];                                   // problem i rides the column kernel
    int chunk[CTGAN_WGRAD_GROUP_LIMIT][CTGAN_WGRAD_MAX_SEGS];
};
void group16_plan_compute(const ctgan_wgrad_group* groups, int n, int mma, G16Plan* plan);
// The plan of a job table is a pure function of its geometry, and a training loop presents the same two or three tables for ever: the last
// eight plans are kept per thread (the column kernel's planner simulates a list schedule over a grid of target times - 0.4 ms of host
// time per call, which an eager (un-graphed) step would pay at every flush).
void group16_plan(const ctgan_wgrad_group* groups, int n, int mma, G16Plan* plan) {
    constexpr int PER = 16 + CTGAN_WGRAD_MAX_SEGS, SLOTS = 8;
    struct Entry { int len; int key[2 + CTGAN_WGRAD_GROUP_LIMIT * PER]; G16Plan plan; };
    static thread_local Entry cache[SLOTS];
    static thread_local int used = 0, next = 0;
    static thread_local int key[2 + CTGAN_WGRAD_GROUP_LIMIT * PER];
    int len = 0;
    key[len++] = n; key[len++] = mma;
    for (int i = 0; i < n; ++i) {
        const ctgan_conv_desc& d = groups[i].d;
        const long long hi = (d.xs[0] | d.xs[1] | d.xs[2] | d.xs[3]) >> 31;      // (strides beyond 2^31 elements are outside every kernel here)
        const int f[16] = {d.C, d.H, d.W, d.K, d.R * 64 + d.S, d.P, d.Q, d.stride * 4 + (d.x_up ? 2 : 0) + (groups[i].db ? 1 : 0), d.pad_t, d.pad_l,
                           groups[i].nseg, (int)(hi != 0), (int)d.xs[0], (int)d.xs[1], (int)d.xs[2], (int)d.xs[3]};
        for (int k = 0; k < 16; ++k) key[len++] = f[k];
        for (int k = 0; k < CTGAN_WGRAD_MAX_SEGS; ++k) key[len++] = k < groups[i].nseg ? groups[i].Ns[k] : 0;
    }
    for (int s = 0; s < used; ++s)
        if (cache[s].len == len && memcmp(cache[s].key, key, (size_t)len * sizeof(int)) == 0) { *plan = cache[s].plan; return; }
    group16_plan_compute(groups, n, mma, plan);
    Entry& e = cache[next];
    e.len = len; memcpy(e.key, key, (size_t)len * sizeof(int)); e.plan = *plan;
    next = (next + 1) % SLOTS; if (used < SLOTS) ++used;
}
void group16_plan_compute(const ctgan_wgrad_group* groups, int n, int mma, G16Plan* plan) {
    const double px_us = mma == CTGAN_MMA_F32X3 ? 0.07 : 0.02;        // one MFMA per product instead of six, 64-pixel slices
    // column-kernel problems
    ctgan_wc_problem wc[CTGAN_WGRAD_GROUP_LIMIT * CTGAN_WGRAD_MAX_SEGS];
    int wc_chunk[CTGAN_WGRAD_GROUP_LIMIT * CTGAN_WGRAD_MAX_SEGS];
    int nwc = 0;
    for (int i = 0; i < n; ++i) {
        int max_rows = 1;
        for (int k = 0; k < groups[i].nseg; ++k) if (groups[i].Ns[k] > max_rows) max_rows = groups[i].Ns[k];
        plan->col[i] = ctgan_wgrad16c_takes(&groups[i].d, mma, max_rows);
        if (!plan->col[i]) continue;
        for (int k = 0; k < groups[i].nseg; ++k) { ctgan_wc_problem& w = wc[nwc++]; w = ctgan_wc_problem{}; w.d = &groups[i].d; w.N = groups[i].Ns[k]; }
    }
    if (nwc) ctgan_wgrad16c_plan(wc, nwc, mma, wc_chunk);
    // slice-kernel problems
    int best_chunk = 0;
    double best_t = 1e30;
    int max_kg = 0, nrest = 0;
    for (int i = 0; i < n; ++i) {
        if (plan->col[i]) continue;
        ++nrest;
        for (int k = 0; k < groups[i].nseg; ++k) { const int kg = groups[i].Ns[k] * groups[i].d.P * groups[i].d.Q; if (kg > max_kg) max_kg = kg; }
    }
    const int forced = 0;
    for (int chunk = 256; nrest && chunk <= 8192; chunk += 64) {
        if (forced && chunk != forced) continue;
        long long blocks = 0;
        for (int i = 0; i < n; ++i) {
            if (plan->col[i]) continue;
            const ctgan_conv_desc& d = groups[i].d;
            const long long tiles = (long long)d.R * d.S * (d.C / 128) * (d.K / 128);
            long long splits = 0;
            for (int k = 0; k < groups[i].nseg; ++k) splits += ((long long)groups[i].Ns[k] * d.P * d.Q + chunk - 1) / chunk;
            blocks += tiles * splits;
        }
        const long long rounds = (blocks + 511) / 512;
        const double t = (double)rounds * (chunk * px_us + 5.0) + (double)blocks * 0.02;
        if (t < best_t * 0.985) { best_t = t; best_chunk = chunk; }
        if (chunk >= max_kg && !forced) break;
    }
    if (nrest && !best_chunk) best_chunk = forced ? forced : 256;
    size_t ws = 0;
    int w = 0;
    for (int i = 0; i < n; ++i) {
        const ctgan_conv_desc& d = groups[i].d;
        size_t splits = 0;
        for (int k = 0; k < groups[i].nseg; ++k) {
            const int Kg = groups[i].Ns[k] * d.P * d.Q;
            int ch;
            if (plan->col[i]) ch = wc_chunk[w++];
            else {      // the planned number of splits, of EQUAL length within the problem (the planner's chunk is the upper bound)
                const int sp = (Kg + best_chunk - 1) / best_chunk;
                ch = (((Kg + sp - 1) / sp) + 63) / 64 * 64;
            }
            plan->chunk[i][k] = ch;
            splits += (size_t)((Kg + ch - 1) / ch);
        }
        ws += (splits * ((size_t)d.R * d.S * d.C + (groups[i].db ? 1 : 0)) * d.K * sizeof(float) + 255) & ~(size_t)255;
    }
    plan->ws_bytes = ws;
}
}  // namespace

template <int MMA>
static int launch_wgrad16_group(const W16Group& g, int blocks, hipStream_t st) {
    constexpr int bkp = planes<MMA>() == 3 ? 32 : 64;
    constexpr size_t lds = (size_t)(wgrad16_single_stage<MMA, 2, 2>() ? 1 : 2) * planes<MMA>() * (128 + 128) * (bkp + 8) * 2;
    auto kern = wgrad16_group_kernel<MMA, 2, 2>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "wgrad16_group: cannot reserve %zu B of LDS", lds);
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, g);
    ctgan_set_last_kernel(MMA == CTGAN_MMA_F32X3 ? "wgrad16x3_group<128x128>" : "wgrad16_group<128x128>");
    ctgan_set_last_symbol("wgrad16_group_kernel<%d, 2, 2>", MMA);
    return ctgan_check_launch("wgrad16_group");
}

extern "C" {

size_t ctgan_conv2d16_wgrad_group_workspace_bytes(const ctgan_wgrad_group* groups, int32_t n, int mma) {
    if (!groups || n < 1 || n > CTGAN_WGRAD_GROUP_LIMIT || !mma_ok(mma)) return 0;
    for (int i = 0; i < n; ++i) if (!group16_member_ok(groups[i], mma)) return 0;
    static thread_local G16Plan plan;
    group16_plan(groups, n, mma, &plan);
    return plan.ws_bytes;
}

int ctgan_conv2d16_wgrad_group(const ctgan_wgrad_group* groups, int32_t n, int mma, void* ws, size_t ws_bytes, int phases, ctgan_stream_t stream) {
    if (!groups || n < 1 || n > CTGAN_WGRAD_GROUP_LIMIT || !mma_ok(mma)) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_wgrad_group: bad argument");
    for (int i = 0; i < n; ++i) {
        if (!groups[i].dw) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_wgrad_group: null dw");
        for (int k = 0; k < groups[i].nseg && k < CTGAN_WGRAD_MAX_SEGS; ++k)
            if (!groups[i].xs[k] || !groups[i].dys[k]) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_wgrad_group: null operand");
        if (!group16_member_ok(groups[i], mma)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d16_wgrad_group: problem %d outside the grouped 128x128 tile", i);
    }
    static thread_local G16Plan plan;
    group16_plan(groups, n, mma, &plan);
    if (!ws || plan.ws_bytes > ws_bytes) return ctgan_fail(CTGAN_E_BADARG, "conv2d16_wgrad_group: workspace %zu B < %zu B", ws_bytes, plan.ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    static thread_local G16Seg segs[CTGAN_WGRAD_GROUP_LIMIT * CTGAN_WGRAD_MAX_SEGS];
    static thread_local ctgan_wc_problem wcs[CTGAN_WGRAD_GROUP_LIMIT * CTGAN_WGRAD_MAX_SEGS];
    static thread_local R16Job red[CTGAN_WGRAD_GROUP_LIMIT];
    int ns = 0, nwc = 0;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const ctgan_wgrad_group& G = groups[i];
        const ctgan_conv_desc* d = &G.d;
        const int rows = d->R * d->S * d->C + (G.db ? 1 : 0);
        float* slab = reinterpret_cast<float*>(static_cast<char*>(ws) + off);
        int splits_total = 0;
        for (int k = 0; k < G.nseg; ++k) {
            const int Kg = G.Ns[k] * d->P * d->Q;
            const int chunk = plan.chunk[i][k];
            const int with_bias = G.db ? ((G.seg_flags[k] & CTGAN_WGRAD_SEG_BIAS) ? 1 : 2) : 0;
            const int relu_x = (G.seg_flags[k] & CTGAN_IN_RELU) ? 1 : 0;
            float* out = slab + (size_t)splits_total * rows * d->K;
            if (plan.col[i]) {
                ctgan_wc_problem& w = wcs[nwc++];
                w.d = d; w.x = G.xs[k]; w.dy = G.dys[k]; w.out = out; w.N = G.Ns[k]; w.relu_x = relu_x; w.with_bias = with_bias; w.chunk = chunk;
                splits_total += (Kg + chunk - 1) / chunk;
                continue;
            }
            W16 p{};
            p.X = G.xs[k]; p.DY = G.dys[k];
            p.OUT = out;
            p.with_bias = with_bias;
            p.H = d->H; p.W = d->W; p.P = d->P; p.Q = d->Q; p.C = d->C; p.R = d->R; p.S = d->S; p.stride = d->stride;
            p.pad_t = d->pad_t; p.pad_l = d->pad_l;
            p.s_n = d->xs[0]; p.s_h = d->xs[2]; p.s_w = d->xs[3];
            p.Mtot = d->R * d->S * d->C; p.Ng = d->K; p.Kg = Kg;
            p.relu_x = relu_x; p.dbg = 0;
            p.chunk = chunk;
            {
                const int pq = d->P * d->Q;
                const bool pow2 = !(pq & (pq - 1)) && !(d->Q & (d->Q - 1));
                p.pq_shift = pow2 ? __builtin_ctz(pq) : -1; p.q_shift = pow2 ? __builtin_ctz(d->Q) : -1;
            }
            const long long x_extent = (long long)(G.Ns[k] - 1) * d->xs[0] + (long long)(d->H - 1) * d->xs[2] + (long long)(d->W - 1) * d->xs[3] + d->C;
            p.x_bytes = (unsigned)(x_extent * 4); p.dy_bytes = (unsigned)((long long)p.Kg * d->K * 4);
            G16Seg& sg = segs[ns++];
            sg.p = p; sg.tiles = d->R * d->S * (d->C / 128) * (d->K / 128); sg.splits = (p.Kg + p.chunk - 1) / p.chunk;
            splits_total += sg.splits;
        }
        const long long n_main = (long long)d->R * d->S * d->C * d->K;
        red[i].part = slab; red[i].out = G.dw; red[i].out2 = G.db ? G.db : G.dw;
        red[i].add = G.add_dw; red[i].add2 = G.db ? G.add_db : nullptr;
        red[i].n = n_main + (G.db ? d->K : 0); red[i].n_main = n_main; red[i].splits = splits_total; red[i].pad = 0;
        off += ((size_t)splits_total * rows * d->K * sizeof(float) + 255) & ~(size_t)255;
    }
    g_last_group_kinds = (nwc ? 1 : 0) | (ns ? 2 : 0);
    g_last_group_col_mask = 0;
    for (int i = 0; i < n; ++i) if (plan.col[i]) g_last_group_col_mask |= 1u << i;
    // (CTGAN_WGRAD_GROUP_TILE0 << 0 / << 1: only the filter-column / only the slice kernel's launch - bench.py times them apart)
    const int only = phases & CTGAN_WGRAD_GROUP_TILE_MASK;
    if ((phases & CTGAN_WGRAD_GROUP_GEMM) && nwc && (!only || (only & CTGAN_WGRAD_GROUP_TILE0))) {
        const int rc = ctgan_wgrad16c_launch(wcs, nwc, mma, st);
        if (rc) return rc;
    }
    if ((phases & CTGAN_WGRAD_GROUP_GEMM) && ns && (!only || (only & (CTGAN_WGRAD_GROUP_TILE0 << 1)))) {
        // longest workgroups first
        int order[CTGAN_WGRAD_GROUP_LIMIT * CTGAN_WGRAD_MAX_SEGS];
        for (int a = 0; a < ns; ++a) order[a] = a;
        for (int a = 1; a < ns; ++a)
            for (int b = a; b > 0 && segs[order[b]].p.chunk > segs[order[b - 1]].p.chunk; --b) { const int x = order[b]; order[b] = order[b - 1]; order[b - 1] = x; }
        for (int base = 0; base < ns; base += W16_GROUP_MAX) {
            W16Group g;
            g.n = (ns - base) < W16_GROUP_MAX ? (ns - base) : W16_GROUP_MAX;
            int b0 = 0;
            for (int k = 0; k < W16_GROUP_MAX; ++k) {
                const G16Seg& sg = segs[order[base + (k < g.n ? k : 0)]];
                g.first[k] = b0; g.tiles[k] = sg.tiles; g.j[k] = sg.p;
                if (k < g.n) b0 += sg.tiles * sg.splits;
            }
            g.first[W16_GROUP_MAX] = b0;
            const int rc = mma == CTGAN_MMA_F32X3 ? launch_wgrad16_group<CTGAN_MMA_F32X3>(g, b0, st)
                         : (mma == CTGAN_MMA_BF16 ? launch_wgrad16_group<CTGAN_MMA_BF16>(g, b0, st) : launch_wgrad16_group<CTGAN_MMA_F16>(g, b0, st));
            if (rc) return rc;
        }
    }
    for (int base = 0; base < n && (phases & CTGAN_WGRAD_GROUP_REDUCE); base += R16_BATCH) {
        R16Jobs jobs;
        jobs.n = (n - base) < R16_BATCH ? (n - base) : R16_BATCH; jobs.pad = 0;
        long long max_n = 0;
        for (int k = 0; k < R16_BATCH; ++k) {
            jobs.j[k] = red[base + (k < jobs.n ? k : 0)];
            if (k < jobs.n && jobs.j[k].n > max_n) max_n = jobs.j[k].n;
        }
        hipLaunchKernelGGL(reduce16_batch_kernel, dim3((unsigned)((max_n / 4 + 255) / 256), jobs.n), dim3(256), 0, st, jobs);
        int rc = ctgan_check_launch("reduce16_batch");
        if (rc) return rc;
    }
    return CTGAN_OK;
}

}  // extern "C"
