// wgrad16c.hip - the "filter column" weight-gradient kernel of the 16-bit matrix-core family (round 4).
//
//   dW[r][s][c][k] = sum over pixels (n,p,q) of  X(n, p*st - pad_t + r, q*st - pad_l + s, c) * dY(n,p,q,k)
//                    (the gradient TF builds for tf.nn.conv2d, TF/tflib/ops/conv2d.py:106-112, under compute_gradients,
//                     TF/CT_gan_cifar_resnet.py:335-336)
//
// The slice kernel (igemm16.hip: wgrad16_body) gives every tap its own 128x128 tile: each of the R*S tap tiles of a filter re-reads and
// re-splits the same dy slice and a shifted copy of the same x slice (4.8x the algorithmic traffic, 112 split VALU + 24 LDS stores per
// 48 MFMAs per wave: matrix pipes busy 0.45 - profiles/r03_pmc_traffic_x3.json).  Here ONE staged dy slice and ONE x stream feed all the
// taps of a filter COLUMN (fixed s, r = 0..R-1):
//   * the K axis (pixels) is walked in slices of 32 pixels; x lives in an LDS RING of four slices (128 pixels per channel row, pixel
//     contiguous, three bf16 planes of the fp32 split), staged one slice ahead of dy.  x is staged with the column's horizontal shift
//     applied (and its zero padding), once per slice: 32 new pixels per slice whatever the number of taps.
//   * tap r reads its "A" fragments from the ring at a ROW offset: pixel t + (r - pad_t)*Q.  Q is a multiple of 8, so the shifted
//     fragment is again a 16-byte aligned unit of 8 consecutive pixels - a horizontal shift would not be, which is why the column, not
//     the filter row, is the unit of reuse.  Rows outside the image (the SAME padding rows) are read from a zero row of LDS instead:
//     one v_cndmask on the fragment address, no exec masking, no branch.
//   * a stride-2 filter is the sum of four stride-1 filters on the (row parity, column parity) decimations of x (polyphase): ih = 2p - pad_t + r
//     = 2(p + dr) + rho.  Decimation is a stride change of the gather, so the same kernel takes the folded ConvMeanPool / UpsampleConv
//     filters (4x4, stride 2: eight columns of two taps each) - 48 % of the critic step's weight-gradient FLOPs.
//   * workgroup = 8 waves (4 channel blocks x 2 kout blocks), one per CU: wave = 32 channels x 64 kout x NTAP taps (2 accumulators per
//     tap); waves 0-3 stage x, waves 4-7 stage dy (one 4-pixel x 4-channel block per thread per slice, transposed by register naming as
//     in wgrad16_body); ONE barrier per slice.  Per wave and slice with three taps: 72 MFMAs against 56 split VALU + 12 LDS stores.
// Split over the pixel axis into fp32 slabs, fixed-order reduction (reduce16_batch_kernel): deterministic.  Same slab layout as the slice
// kernel, so both kinds of problem share the reduction launch of a grouped call.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"
#include "mma16.h"
#include "wgrad16c.h"

// WC_DBG: compile-time perf-diagnosis bits for A/B builds (tools/build_variant.sh; results are then WRONG by design, never set in the product):
// 2 = no split arithmetic (three roundings, no remainders), 4 = no MFMAs, 8 = no staging stores, 16 = remainders by unpack + v_sub instead of v_dot2c,
// 32 = no fragment reads from LDS, 64 = no barriers, 128 = no global loads in the loop
#ifndef WC_DBG
#define WC_DBG 0
#endif

namespace {

// LDS layout per arithmetic mode.  Split mode (three bf16 planes per operand): slices of 32 pixels; bf16 / fp16 (one plane): slices of 64.
template <int MMA> struct WC {
    static constexpr int NP = planes<MMA>();
    static constexpr int BKP = NP == 3 ? 32 : 64;        // pixels per K slice
    static constexpr int KS = BKP / 16;                  // MFMA k steps per slice
    static constexpr int XB = BKP / 32;                  // staging blocks (4 pixels x 4 channels) per thread and slice
    static constexpr int UNITS = 4 * BKP / 8;            // the x ring: four slices, in units of 8 pixels (16 B per plane)
    static constexpr int XPL = 4 * BKP * 2 + 16;         // bytes per (channel row, plane): the ring + 16 B, so that the row stride - 816 B =
    static constexpr int XROW = NP * XPL;                // 204 dwords = 12 (mod 64), or 528 B = 132 = 4 (mod 64) - spreads 16 rows over all banks
    static constexpr int XZERO = 128 * XROW;             // a row of zeros: the fragment address of a tap whose image row is padding
    static constexpr int XBYTES = 129 * XROW;
    static constexpr int YPL = BKP * 2;                  // dy: [kout row][plane][slice pixels]
    static constexpr int YROW = NP * YPL + 16;           // 208 B = 52 dwords = 4 * 13 / 144 B = 36 = 4 * 9: conflict-free the same way
    static constexpr int YSTAGE = 128 * YROW;
    static constexpr int LDS_TOTAL = XBYTES + 2 * YSTAGE;      // 158,512 B (split mode) / 104,976 B: one workgroup per CU
};

struct WCol { int xoff; unsigned pk; };  // pk = (dr0 + 8) | (dc + 8) << 4 | ntap << 8 | tap0 << 12 | tap1 << 18 | tap2 << 24   (tap = r*S + s < 64)
struct W16C {
    const float* X; const float* DY;
    float* OUT;                          // [splits][Mtot (+1)][Ng] fp32 slabs
    int P, Q, C, Ng, Kg, chunk;          // dy grid, channels, kout, pixels N*P*Q, pixels per split (multiple of 64)
    int s_n, s_h, s_w;                   // element strides of the DECIMATED x view (stride * the tensor's), channel stride 1
    int Mtot, relu_x, with_bias;
    unsigned x_bytes, dy_bytes;
    int pq_shift, q_shift;
    int ncols, tiles;                    // columns; workgroups per split = ncols * (C/128) * (Ng/128)
    int flags;                           // bit 1: split-major block order (A/B switch CTGAN_WGRAD16_COL_ORDER=1)
    int splits;
    WCol col[CTGAN_WC_MAXCOL];
};
constexpr int WC_GROUP_MAX = 16;
struct W16CGroup { int n; int first[WC_GROUP_MAX + 1]; W16C j[WC_GROUP_MAX]; };
static_assert(sizeof(W16CGroup) <= 4000, "kernel arguments");

template <int MMA, int NTAP, bool Q8>
__device__ __forceinline__ void wgrad16c_body(const W16C& p, const WCol col, const int cb, const int tn, const int by, unsigned char* smem) {
    using T = WC<MMA>;
    constexpr int NP = T::NP, BKP = T::BKP, KS = T::KS, XB = T::XB, UMASK = T::UNITS - 1, XPL = T::XPL, XROW = T::XROW, XZERO = T::XZERO,
                  XBYTES = T::XBYTES, YPL = T::YPL, YROW = T::YROW, YSTAGE = T::YSTAGE;
    unsigned char* const xs = smem;
    unsigned char* const ys = smem + XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // waves 0-3 stage x, waves 4-7 stage dy.  (Wave-uniform BY CONSTRUCTION for the compiler - readfirstlane - so that the two roles are
    // scalar branches / scalar selects of the buffer descriptor, not exec masks and v_readfirstlane waterfall loops around the loads.)
    const bool is_x = __builtin_amdgcn_readfirstlane(wave) < 4;
    const int stg = tid & 255, pg = stg & 7, cg = stg >> 3; // staging block: 4 pixels (group pg of the slice) x 4 channels (group cg)
    const int dr0 = (int)(col.pk & 15u) - 8, dc = (int)((col.pk >> 4) & 15u) - 8;
    const int c0 = cb * 128, n0 = tn * 128;
    const int k_begin = by * p.chunk;
    const int k_end = min(k_begin + p.chunk, p.Kg);
    const int t0 = k_begin / BKP, t1 = (k_end + BKP - 1) / BKP;
    const int qu = p.Q >> 3;                               // ring units (8 pixels) per image row

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, (unsigned)k_end * (unsigned)p.Ng * 4u, 0x00020000);
    const unsigned wstep = (unsigned)p.s_w * 4u, ystep = (unsigned)p.Ng * 4u;

    // x block of slice sl: pixels 32 sl + 4 pg .. +3 of the decimated view, shifted by the column's dc (zeros where the shifted column
    // leaves the image: the SAME padding columns).  x is dense per image (s_n = P * s_h, checked by the host), so the byte offset of a
    // pixel is linear in its global row index: a thread's four offsets are loop invariants plus one scalar per slice.  Pixels past Kg lie
    // past the tensor (hardware range check: zeros); an invalid column gets an offset past every tensor (x_bytes < 2 GiB).
    unsigned xoff_e[XB][4], yoff_e[XB][4];
#pragma unroll
    for (int b = 0; b < XB; ++b) {
        const int px = (pg + 8 * b) * 4, pp = px >> p.q_shift, qq = px & (p.Q - 1);       // (BKP % Q == 0: the slice adds whole rows)
        const unsigned base = (unsigned)(col.xoff + pp * p.s_h + c0 + cg * 4) * 4u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qe = qq + e + dc;
            xoff_e[b][e] = (unsigned)qe < (unsigned)p.Q ? base + (unsigned)qe * wstep : 0x80000000u;
            yoff_e[b][e] = ((unsigned)(px + e) * (unsigned)p.Ng + (unsigned)(n0 + cg * 4)) * 4u;
        }
    }
    const unsigned x_slice = (unsigned)(BKP >> p.q_shift) * (unsigned)p.s_h * 4u, y_slice = (unsigned)BKP * ystep;
    auto load_x = [&](int sl, float4 (&rv)[XB][4]) {
        const unsigned so = (unsigned)sl * x_slice;
#pragma unroll
        for (int b = 0; b < XB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((WC_DBG & 128) && sl > t0 + 2) { rv[b][e].x += 1.f; continue; }
                rv[b][e] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xoff_e[b][e] + so, 0, 0));
            }
    };
    auto load_y = [&](int sl, float4 (&rv)[XB][4]) {     // (the descriptor ends at this split's last pixel: the next split's read as zeros)
        const unsigned so = (unsigned)sl * y_slice;
#pragma unroll
        for (int b = 0; b < XB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((WC_DBG & 128) && sl > t0 + 2) { rv[b][e].x += 1.f; continue; }
                rv[b][e] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, yoff_e[b][e] + so, 0, 0));
            }
    };
    // A staging block = 4 consecutive pixels x 4 channels.  Transposed by register naming: channel j of pixels (0,1) and (2,3) -> two
    // packed dwords per plane (8-byte LDS stores, conflict-free).  The eight pairs go through the three split levels TOGETHER (level by
    // level, not pair by pair): eight independent chains, so no instruction waits for the one before it (v_dot2c -> v_cvt_pk needs two
    // idle slots when they are adjacent).
    auto put_block = [&](unsigned char* dst, int rowstride, int plane, const float4 (&v)[4]) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        unsigned klo = 0x0000BF80u, khi = 0xBF800000u;        // bf16 pairs (-1, 0) / (0, -1), kept opaque (see split_pk)
        asm("" : "+s"(klo));
        asm("" : "+s"(khi));
        const bf2 lo = __builtin_bit_cast(bf2, klo), hi = __builtin_bit_cast(bf2, khi);
        float a[8], b[8];
        a[0] = v[0].x; b[0] = v[1].x; a[1] = v[2].x; b[1] = v[3].x;
        a[2] = v[0].y; b[2] = v[1].y; a[3] = v[2].y; b[3] = v[3].y;
        a[4] = v[0].z; b[4] = v[1].z; a[5] = v[2].z; b[5] = v[3].z;
        a[6] = v[0].w; b[6] = v[1].w; a[7] = v[2].w; b[7] = v[3].w;
        unsigned pl[NP][8];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int k = 0; k < 8; ++k) pl[q][k] = Cvt<MMA>::pk(a[k], b[k]);
            if (q + 1 < NP && !(WC_DBG & 2)) {      // (one-plane modes: the rounded piece is all there is)
#pragma unroll
                for (int k = 0; k < 8; ++k) {       // remainder = value - piece, exact (one v_dot2c per value)
                    if (WC_DBG & 16) {
                        a[k] -= __builtin_bit_cast(float, pl[q][k] << 16);
                        b[k] -= __builtin_bit_cast(float, pl[q][k] & 0xFFFF0000u);
                    } else {
                        a[k] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pl[q][k]), lo, a[k], false);
                        b[k] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pl[q][k]), hi, b[k], false);
                    }
                }
            }
        }
        if (WC_DBG & 8) {      // (keep the values alive without storing them)
            unsigned x = 0;
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int k = 0; k < 8; ++k) x ^= pl[q][k];
            if (x == 0x12345678u) *reinterpret_cast<unsigned*>(dst) = x;
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) { const u32x2 o = {pl[q][2 * j], pl[q][2 * j + 1]}; *reinterpret_cast<u32x2*>(dst + j * rowstride + q * plane) = o; }
    };
    const int relu_lim = p.relu_x ? 0 : (int)0x80000000;
    auto store_x = [&](float4 (&vv)[XB][4], int slot) {
        // relu on load, branch-free: as signed integers every negative float (and -0) is below 0, every positive one unchanged by
        // max(., 0); a problem without relu takes max(., INT_MIN)
#pragma unroll
        for (int b = 0; b < XB; ++b) {
            float4 (&v)[4] = vv[b];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e].x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v[e].x), relu_lim)); v[e].y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v[e].y), relu_lim));
                v[e].z = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v[e].z), relu_lim)); v[e].w = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v[e].w), relu_lim));
            }
            unsigned char* dst = xs + (cg * 4) * XROW + slot * (BKP * 2) + (pg + 8 * b) * 8;
            put_block(dst, XROW, XPL, v);
        }
    };
    const bool bias_wg = p.with_bias && cb == 0 && (col.pk >> 31);      // the first column's workgroups sum the dy tiles they stage
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_y = [&](const float4 (&vv)[XB][4], int stage) {
#pragma unroll
        for (int b = 0; b < XB; ++b) {
            const float4 (&v)[4] = vv[b];
            if (bias_wg && p.with_bias == 1) {             // (with_bias == 2: a segment that does not contribute - its slab row stays zero)
                bsum.x += (v[0].x + v[1].x) + (v[2].x + v[3].x); bsum.y += (v[0].y + v[1].y) + (v[2].y + v[3].y);
                bsum.z += (v[0].z + v[1].z) + (v[2].z + v[3].z); bsum.w += (v[0].w + v[1].w) + (v[2].w + v[3].w);
            }
            unsigned char* dst = ys + stage * YSTAGE + (cg * 4) * YROW + (pg + 8 * b) * 8;
            put_block(dst, YROW, YPL, v);
        }
    };

    f32x16 acc[NTAP][2];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;

    const unsigned xrow_lane = (unsigned)(wm * 32 + l31) * XROW;
    const unsigned yrow_lane = (unsigned)(wn * 64 + l31) * YROW + (unsigned)h * 16u;
    // Fragment reads are software-pipelined at tap granularity: the 12 MFMAs of (k step, tap) run while the fragments of the next
    // (k step, tap) are in flight (left to itself the compiler reads a fragment right before the MFMA that needs it: a dozen exposed LDS
    // round trips per slice, waves parked 27 % of their time).  The first fragments of a slice are requested before the slice's
    // staging work, right behind the barrier.
    auto load_b = [&](int sl, int ks, u32x4 (&fb)[NP][2]) {
        const unsigned char* yst = ys + (sl & 1) * YSTAGE + yrow_lane;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (WC_DBG & 32) { fb[q][j] = u32x4{(unsigned)sl, (unsigned)ks, (unsigned)q, (unsigned)j}; continue; }
                fb[q][j] = *reinterpret_cast<const u32x4*>(yst + j * 32 * YROW + q * YPL + ks * 32);
            }
    };
    const unsigned xrow_lane_h = xrow_lane + (unsigned)h * 16u;
    auto load_a = [&](int sl, int ks, int t, u32x4 (&fa)[NP]) {
        const int dr = dr0 + t;
        unsigned a;
        if constexpr (Q8) {
            // a k step of 16 pixels spans two rows of an 8-wide image: the lane half's 8 pixels lie in image row prow
            const int prow = ((sl * BKP + ks * 16 + h * 8) >> 3) & (p.P - 1);
            const bool valid = (unsigned)(prow + dr) < (unsigned)p.P;
            const unsigned unit = (unsigned)(sl * (BKP / 8) + ks * 2 + dr + h) & (unsigned)UMASK;
            a = valid ? xrow_lane + unit * 16u : (unsigned)XZERO;
        } else {
            // rows of 16 / 32 pixels: the k step lies in ONE image row - validity and ring unit are wave-uniform (scalar ALU), the unit is
            // even, so the lane half adds its 16 bytes without wrapping
            const int prow = ((sl * BKP + ks * 16) >> p.q_shift) & (p.P - 1);
            const bool valid = (unsigned)(prow + dr) < (unsigned)p.P;
            const unsigned uoff = ((unsigned)(sl * (BKP / 8) + ks * 2 + dr * qu) & (unsigned)UMASK) * 16u;
            a = (valid ? xrow_lane_h : (unsigned)XZERO) + (valid ? uoff : 0u);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (WC_DBG & 32) { fa[q] = u32x4{a, (unsigned)q, 1u, 2u}; continue; }
            fa[q] = *reinterpret_cast<const u32x4*>(xs + a + q * XPL);
        }
    };
    auto mma_tap = [&](const u32x4 (&fa)[NP], const u32x4 (&fb)[NP][2], auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (NP == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][j] = Cvt<MMA>::mma(fa[0], fb[0][j], acc[t][j]);
        } else {
            constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};      // small products first (as conv16_kernel)
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (WC_DBG & 4) { acc[t][j][c] += __builtin_bit_cast(float, fa[QA[c]][0] ^ fb[QB[c]][j][1]); continue; }
                    acc[t][j] = Cvt<MMA>::mma(fa[QA[c]], fb[QB[c]][j], acc[t][j]);
                }
        }
    };
    u32x4 fbr[2][NP][2], far[2][NP];
    auto mma_head = [&](int sl) { load_b(sl, 0, fbr[0]); load_a(sl, 0, 0, far[0]); };
    auto mma_steps = [&](int sl) {
        // step i = ks * NTAP + t uses far[i & 1], fbr[ks & 1]; the next step's fragments are requested first
        auto step = [&](auto i_c) {
            constexpr int i = decltype(i_c)::value, ks = i / NTAP, t = i % NTAP;
            if constexpr (i + 1 < KS * NTAP) {
                constexpr int ks1 = (i + 1) / NTAP, t1 = (i + 1) % NTAP;
                if constexpr (t1 == 0) load_b(sl, ks1, fbr[ks1 & 1]);
                load_a(sl, ks1, t1, far[(i + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(far[i & 1], fbr[ks & 1], std::integral_constant<int, t>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        auto run = [&](auto self, auto i_c) {
            constexpr int i = decltype(i_c)::value;
            if constexpr (i < KS * NTAP) { step(i_c); self(self, std::integral_constant<int, i + 1>{}); }
        };
        run(run, std::integral_constant<int, 0>{});
    };

    // prologue: the zero row; x slices t0-1, t0, t0+1 and dy slice t0 staged, the next slice of each in registers
    if (tid < XROW / 4) reinterpret_cast<unsigned*>(xs + XZERO)[tid] = 0u;
    float4 rv[XB][4];
    if (is_x) {
        float4 r3[3][XB][4];
        if (t0 > 0) load_x(t0 - 1, r3[0]);
        else {      // before the first pixel: rows no tap reads as valid
#pragma unroll
            for (int b = 0; b < XB; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) r3[0][b][e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        load_x(t0, r3[1]); load_x(t0 + 1, r3[2]);
        load_x(t0 + 2, rv);
        store_x(r3[0], (t0 - 1) & 3); store_x(r3[1], t0 & 3); store_x(r3[2], (t0 + 1) & 3);
    } else {
        float4 r1[XB][4];
        load_y(t0, r1);
        load_y(t0 + 1, rv);
        store_y(r1, t0 & 1);
    }
    __syncthreads();
    // Slice sl is multiplied while x slice sl+2 / dy slice sl+1 go from registers to LDS and the loads of the slices after them are issued.
    // (Stores and loads run unconditionally, also in the last two iterations: what they stage is never multiplied - x of the next split,
    // dy past the descriptor's end = zeros - and a conditional load would keep the staging registers live across the iteration, which
    // costs a register copy per value in front of the in-place split.  Measured and dropped: the dy-staging waves multiplying first and
    // staging afterwards, so that each SIMD's two waves are in opposite phases - 351 vs 361 us on the critic step's table, 416 vs 403 on
    // the generator step's: neutral.)
    for (int sl = t0; sl < t1; ++sl) {
        mma_head(sl);
        __builtin_amdgcn_sched_barrier(0);
        if (is_x) { store_x(rv, (sl + 2) & 3); load_x(sl + 3, rv); }
        else { store_y(rv, (sl + 1) & 1); load_y(sl + 2, rv); }
        __builtin_amdgcn_sched_barrier(0);
        mma_steps(sl);
        if (!(WC_DBG & 64)) __syncthreads();
    }

    // acc[t][j][4g + e] = dW(tap_t, channel c0 + wm*32 + 8g + 4h + e, kout n0 + wn*64 + j*32 + l31): 32 lanes = 128-byte rows
    float* out = p.OUT + (long long)by * (p.Mtot + (p.with_bias ? 1 : 0)) * p.Ng;
    if (bias_wg) {
        // column sums of this split's dy tile: a dy-staging thread holds 4 kout of its pixel group; fold the 8 pixel groups through LDS in a fixed order
        float* red = reinterpret_cast<float*>(ys);
        if (!is_x) *reinterpret_cast<float4*>(&red[pg * 128 + cg * 4]) = bsum;
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 8; ++g2) t += red[g2 * 128 + tid];
            out[(long long)p.Mtot * p.Ng + n0 + tid] = t;
        }
    }
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int tap = (int)((col.pk >> (12 + 6 * t)) & 63u);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kcol = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = c0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                out[((long long)tap * p.C + c) * p.Ng + kcol] = acc[t][j][e];
            }
        }
    }
}

// Workgroups [first[j], first[j+1]) belong to problem j: split-major, then (column, channel block, kout block).
template <int MMA>
__global__ __launch_bounds__(512) void wgrad16c_group_kernel(const W16CGroup g) {      // dynamic LDS: WC<MMA>::LDS_TOTAL
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int job = 0;
    while (job + 1 < g.n && (int)blockIdx.x >= g.first[job + 1]) ++job;
    job = __builtin_amdgcn_readfirstlane(job);
    const W16C& p = g.j[job];
    const int local = (int)blockIdx.x - g.first[job];
    // Block order inside a problem.  The columns of a filter read the SAME dy chunk and (shifted) the same x chunk: launched as
    // (split, column) they land on different XCDs at different times and every column fetches its operands from memory again (L2 hit rate
    // 0.05, 2.5x the algorithmic bytes, and the launch takes 190 us with its MFMAs switched off: the fabric, not the matrix pipe, sets its
    // pace).  Workgroup b runs on XCD b % 8, so the blocks of eight consecutive splits are interleaved: all column / channel / kout tiles of
    // one split are 8 block indices apart - same XCD, dispatched back to back, sharing one L2.
    int by, tile;
    if (p.flags & 2) { by = local / p.tiles; tile = local - by * p.tiles; }
    else {
        const int per_group = 8 * p.tiles, q = local / per_group, r = local - q * per_group;
        const int m = min(8, p.splits - 8 * q);
        tile = r / m; by = 8 * q + (r - tile * m);
    }
    const int tiles_n = p.Ng >> 7, cblocks = p.C >> 7;
    const int tn = tile % tiles_n, t2 = tile / tiles_n, cb = t2 % cblocks, ci = t2 / cblocks;
    WCol col = p.col[ci];
    if (ci == 0) col.pk |= 0x80000000u;                    // (bit 31: the column that owns the bias row)
    const int ntap = (int)((col.pk >> 8) & 15u);
    if (p.Q == 8) {
        if (ntap == 3) wgrad16c_body<MMA, 3, true>(p, col, cb, tn, by, smem);
        else if (ntap == 2) wgrad16c_body<MMA, 2, true>(p, col, cb, tn, by, smem);
        else wgrad16c_body<MMA, 1, true>(p, col, cb, tn, by, smem);
    } else {
        if (ntap == 3) wgrad16c_body<MMA, 3, false>(p, col, cb, tn, by, smem);
        else if (ntap == 2) wgrad16c_body<MMA, 2, false>(p, col, cb, tn, by, smem);
        else wgrad16c_body<MMA, 1, false>(p, col, cb, tn, by, smem);
    }
}

// ---------------------------------------------------------------------------------------------- host side
int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
int posmod(int a, int b) { const int m = a % b; return m < 0 ? m + b : m; }

struct ColGeom { int ncols; WCol col[CTGAN_WC_MAXCOL]; int ntap[CTGAN_WC_MAXCOL]; };

// The columns of a filter: (row class rho, s) -> the taps r with (r - pad_t) mod stride == rho, in runs of at most three consecutive
// row offsets dr = (r - pad_t - rho) / stride.  false: outside the kernel (see ctgan_wgrad16c_takes).
bool col_geom(const ctgan_conv_desc* d, ColGeom* out, int bkp) {
    const int st = d->stride;
    ColGeom g{};
    for (int rho = 0; rho < st; ++rho) {
        int rs[16], nr = 0;
        for (int r = 0; r < d->R; ++r) if (posmod(r - d->pad_t, st) == rho) { if (nr == 16) return false; rs[nr++] = r; }
        if (!nr) continue;
        for (int s = 0; s < d->S; ++s) {
            const int sigma = posmod(s - d->pad_l, st), dc = floordiv(s - d->pad_l - sigma, st);
            if (dc < -8 || dc > 7) return false;
            for (int b = 0; b < nr; b += 3) {
                const int nt = std::min(3, nr - b);
                if (g.ncols == CTGAN_WC_MAXCOL) return false;
                const int dr0 = floordiv(rs[b] - d->pad_t - rho, st);
                if (dr0 < -8 || dr0 + nt - 1 > 7) return false;
                // the ring holds one slice (32 / 64 pixels) behind and one ahead of the slice being multiplied
                if (std::max(-dr0, 0) * d->Q > bkp || std::max(dr0 + nt - 1, 0) * d->Q > bkp) return false;
                WCol& c = g.col[g.ncols];
                c.xoff = (int)(rho * d->xs[2] + sigma * d->xs[3]);
                c.pk = (unsigned)(dr0 + 8) | ((unsigned)(dc + 8) << 4) | ((unsigned)nt << 8);
                for (int t = 0; t < nt; ++t) {
                    const int tap = rs[b + t] * d->S + s;
                    if (tap >= 64) return false;
                    c.pk |= (unsigned)tap << (12 + 6 * t);
                }
                g.ntap[g.ncols++] = nt;
            }
        }
    }
    if (!g.ncols) return false;
    if (out) *out = g;
    return true;
}

int slice_px(int mma) { return mma == CTGAN_MMA_F32X3 ? 32 : 64; }
// Microseconds of a slice and of a workgroup's fixed part (prologue, slab store).  Split mode: MFMA-bound, 0.35 + 1.05 per tap and 32-pixel
// slice.  (One round of 256 EQUAL workgroups measures more - 2.06 / 3.46 / 3.98 us for 1 / 2 / 3 taps, tools/wgrad_col_calib.py,
// profiles/r04_wgrad_col_calib.txt: a launch of identical MFMA-dense workgroups pulls the clock down - but the plan those costs select
// for the critic step's mixed table, two rounds at 80 MB of slabs, measured 487-531 us against 351 for this model's one-round plan.)
// One-plane modes: 64 KB of operands per 64-pixel slice and CU at ~25 GB/s per CU - the fabric, not the matrix pipe - whatever the number
// of taps: 3.19 / 2.60 / 2.65 us measured the same way.
double slice_us(int mma, int ntap) {
    static const double h16[3] = {3.19, 2.60, 2.65};
    if (mma == CTGAN_MMA_F32X3) return 0.35 + 1.05 * ntap;
    return h16[ntap < 1 ? 0 : (ntap > 3 ? 2 : ntap - 1)];
}
double fixed_us(int mma) { return mma == CTGAN_MMA_F32X3 ? 4.0 : 10.0; }
// ... of one workgroup
double wg_cost(int mma, int ntap, int chunk) { return (double)(chunk / slice_px(mma)) * slice_us(mma, ntap) + fixed_us(mma); }

}  // namespace

// Geometry the column kernel takes (mma = CTGAN_MMA_F32X3 for now): channel and kout counts multiples of 128, dense channels-last dy
// (checked by the caller), power-of-two dy grid with rows of 8 / 16 / 32 pixels and at least 64 pixels per image, SAME geometry with
// H = stride * P and W = stride * Q, stride 1 or 2, every tap within one slice of its pixel (|dr| * Q <= 32), x dense per image and, over
// `max_rows` samples, below 2 GiB.
bool ctgan_wgrad16c_takes(const ctgan_conv_desc* d, int mma, int max_rows) {
    static const int off = [] { const char* e = getenv("CTGAN_WGRAD16_COL"); return e && atoi(e) == 0; }();
    if (off || (mma != CTGAN_MMA_F32X3 && mma != CTGAN_MMA_BF16 && mma != CTGAN_MMA_F16)) return false;
    if (d->x_up || d->C % 128 || d->K % 128 || d->xs[1] != 1) return false;
    if (d->Q != 8 && d->Q != 16 && d->Q != 32 && !(d->Q == 64 && slice_px(mma) == 64)) return false;
    const int pq = d->P * d->Q;
    if ((pq & (pq - 1)) || pq < 64) return false;
    if (d->stride != 1 && d->stride != 2) return false;
    if (d->H != d->stride * d->P || d->W != d->stride * d->Q) return false;
    if (d->xs[0] != (int64_t)d->H * d->xs[2]) return false;           // images dense in memory: a pixel's offset is linear in its global row index
    if (d->xs[0] >= (1LL << 28) || d->xs[2] >= (1LL << 26) || d->xs[3] >= (1LL << 26)) return false;
    // byte offsets: an invalid column is addressed at 2 GiB + its offset, which must lie past the tensor and below 4 GiB
    if (((long long)max_rows * d->xs[0] + d->C) * 4 >= (1LL << 31) || (long long)max_rows * d->P * d->Q * d->K * 4 >= (1LL << 32)) return false;
    return col_geom(d, nullptr, slice_px(mma));
}

int ctgan_wgrad16c_tiles(const ctgan_conv_desc* d, int mma) {
    ColGeom g;
    if (!col_geom(d, &g, slice_px(mma))) return 0;
    return g.ncols * (d->C / 128) * (d->K / 128);
}

// Pixels per split for every problem of a grouped call.  One workgroup per CU is resident, workgroups differ in cost (one to three taps per
// column, problems of different length), and a launch is one to three rounds long: the plan is chosen by a target workgroup time T - every
// problem gets the longest chunk (multiple of 64 pixels, splits of equal length) whose workgroups stay within T - over a grid of T, by the
// simulated schedule on 256 CUs (most expensive first, as the launch orders them) plus the slab traffic the plan causes.  Deterministic: a
// function of the geometries and row counts only (the workspace query and the launch must agree).
void ctgan_wgrad16c_plan(const ctgan_wc_problem* probs, int n, int mma, int* chunks) {
    static const int forced = [] { const char* e = getenv("CTGAN_WGRAD16_COL_CHUNK"); return e ? atoi(e) : 0; }();      // (tools/wgrad_group_bench.py: chunk sweep)
    std::vector<ColGeom> geoms(n);
    std::vector<int> mt(n, 1), kg(n);
    for (int i = 0; i < n; ++i) {
        col_geom(probs[i].d, &geoms[i], slice_px(mma));
        for (int c = 0; c < geoms[i].ncols; ++c) mt[i] = std::max(mt[i], geoms[i].ntap[c]);
        kg[i] = probs[i].N * probs[i].d->P * probs[i].d->Q;
    }
    double best_t = 1e30;
    std::vector<int> cur(n);
    std::vector<std::pair<double, int>> wgs;       // (cost, count)
    std::vector<double> cus(256);
    for (double T = 40.; T < 2000.; T *= 1.04) {
        wgs.clear();
        double slab_bytes = 0.;
        for (int i = 0; i < n; ++i) {
            const ctgan_conv_desc* d = probs[i].d;
            int ch = forced ? forced : (int)((T - fixed_us(mma)) / slice_us(mma, mt[i])) * slice_px(mma);
            ch = std::max(128, ch / 64 * 64);
            const int sp = (kg[i] + ch - 1) / ch;
            ch = (((kg[i] + sp - 1) / sp) + 63) / 64 * 64;
            cur[i] = ch;
            const int splits = (kg[i] + ch - 1) / ch;
            const int per = (d->C / 128) * (d->K / 128);
            for (int c = 0; c < geoms[i].ncols; ++c) wgs.emplace_back(wg_cost(mma, geoms[i].ntap[c], ch), splits * per);
            slab_bytes += (double)splits * ((double)d->R * d->S * d->C + 1) * d->K * 4.;
        }
        std::sort(wgs.begin(), wgs.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
        // list scheduling on 256 CUs: a min-heap of finish times
        std::fill(cus.begin(), cus.end(), 0.);
        std::make_heap(cus.begin(), cus.end(), std::greater<double>());
        double makespan = 0.;
        for (const auto& w : wgs)
            for (int k = 0; k < w.second; ++k) {
                std::pop_heap(cus.begin(), cus.end(), std::greater<double>());
                const double t = cus.back() + w.first;
                cus.back() = t;
                std::push_heap(cus.begin(), cus.end(), std::greater<double>());
                if (t > makespan) makespan = t;
            }
        const double t = makespan + slab_bytes * 2. / 3.0e6;      // slabs written, then read by the reduction: ~3 TB/s each way
        if (t < best_t * 0.995) { best_t = t; for (int i = 0; i < n; ++i) chunks[i] = cur[i]; }
        if (forced) break;
    }
}

template <int MMA>
static int launch_wc(const W16CGroup& g, int blocks, hipStream_t st) {
    auto kern = wgrad16c_group_kernel<MMA>;
    constexpr int lds = WC<MMA>::LDS_TOTAL;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return ctgan_fail(CTGAN_E_LAUNCH, "wgrad16c: cannot reserve %d B of LDS", lds);
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, g);
    ctgan_set_last_kernel(MMA == CTGAN_MMA_F32X3 ? "wgrad16x3_group<col>" : "wgrad16_group<col>");
    ctgan_set_last_symbol("wgrad16c_group_kernel<%d>", MMA);
    return ctgan_check_launch("wgrad16c_group");
}

int ctgan_wgrad16c_launch(const ctgan_wc_problem* probs, int n, int mma, hipStream_t st) {
    if (mma != CTGAN_MMA_F32X3 && mma != CTGAN_MMA_BF16 && mma != CTGAN_MMA_F16) return ctgan_fail(CTGAN_E_UNSUPPORTED, "wgrad16c: unknown mode");
    static const int order_flag = [] { const char* e = getenv("CTGAN_WGRAD16_COL_ORDER"); return (e && atoi(e) == 1) ? 2 : 0; }();
    // most expensive workgroups first
    std::vector<int> order(n);
    std::vector<double> cost(n);
    std::vector<ColGeom> geoms(n);
    for (int i = 0; i < n; ++i) {
        order[i] = i;
        if (!col_geom(probs[i].d, &geoms[i], slice_px(mma))) return ctgan_fail(CTGAN_E_UNSUPPORTED, "wgrad16c: problem %d outside the column kernel", i);
        int mt = 1;
        for (int c = 0; c < geoms[i].ncols; ++c) mt = std::max(mt, geoms[i].ntap[c]);
        cost[i] = wg_cost(mma, mt, probs[i].chunk);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    for (int base = 0; base < n; base += WC_GROUP_MAX) {
        W16CGroup g;
        g.n = std::min(n - base, WC_GROUP_MAX);
        int b0 = 0;
        for (int k = 0; k < WC_GROUP_MAX; ++k) {
            const int i = order[base + (k < g.n ? k : 0)];
            const ctgan_wc_problem& pr = probs[i];
            const ctgan_conv_desc* d = pr.d;
            W16C& p = g.j[k];
            p = W16C{};
            p.X = pr.x; p.DY = pr.dy; p.OUT = pr.out;
            p.P = d->P; p.Q = d->Q; p.C = d->C; p.Ng = d->K; p.Kg = pr.N * d->P * d->Q; p.chunk = pr.chunk;
            p.s_n = (int)d->xs[0]; p.s_h = (int)(d->xs[2] * d->stride); p.s_w = (int)(d->xs[3] * d->stride);
            p.Mtot = d->R * d->S * d->C; p.relu_x = pr.relu_x; p.with_bias = pr.with_bias;
            const long long x_extent = (long long)(pr.N - 1) * d->xs[0] + (long long)(d->H - 1) * d->xs[2] + (long long)(d->W - 1) * d->xs[3] + d->C;
            p.x_bytes = (unsigned)(x_extent * 4); p.dy_bytes = (unsigned)((long long)p.Kg * d->K * 4);
            p.pq_shift = __builtin_ctz(d->P * d->Q); p.q_shift = __builtin_ctz(d->Q);
            p.ncols = geoms[i].ncols; p.tiles = p.ncols * (d->C / 128) * (d->K / 128);
            p.flags = order_flag;
            p.splits = (p.Kg + p.chunk - 1) / p.chunk;
            for (int c = 0; c < p.ncols; ++c) p.col[c] = geoms[i].col[c];
            g.first[k] = b0;
            if (k < g.n) b0 += p.tiles * p.splits;
        }
        g.first[WC_GROUP_MAX] = b0;
        const int rc = mma == CTGAN_MMA_F32X3 ? launch_wc<CTGAN_MMA_F32X3>(g, b0, st)
                     : (mma == CTGAN_MMA_BF16 ? launch_wc<CTGAN_MMA_BF16>(g, b0, st) : launch_wc<CTGAN_MMA_F16>(g, b0, st));
        if (rc) return rc;
    }
    return CTGAN_OK;
}
