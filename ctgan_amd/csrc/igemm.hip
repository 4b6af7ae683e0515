// igemm.hip - implicit-GEMM convolution family on gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One gather-GEMM template serves every conv-shaped op of the CT-WGAN step:
//   FWD   : D[m=(n,p,q)][j=k]     = sum_{(r,s,c)} X(n, p*st-pt+r, q*st-pl+s, c) * Wt(r,s,c,k)
//   DGRAD : the same kernel on dy with flipped / transposed filter taps and an input dilation
//           (dx = conv of the stride-dilated dy with the 180-degree-rotated filter)
//   WGRAD : D[m=(r,s,c)][j=k]     = sum_{(n,p,q)} X(n, p*st-pt+r, q*st-pl+s, c) * dy[n,p,q,k]
//           split over the pixel axis into per-block partial slabs + a fixed-order reduction
//           (deterministic: no float atomics).
// X() applies zero padding, the stride dilation of a transposed conv and the nearest-2x upsample
// of UpsampleConv on the fly, so none of those tensors is ever materialised.
//
// Tiling (MI355X: 256 CUs, 4 SIMDs, 64-wide waves): a workgroup is 4 waves; a wave owns TM x TN
// 32x32 MFMA accumulators; the K loop walks 32-deep slices staged through LDS with register
// prefetch of the next slice (global -> VGPR before the MFMAs, VGPR -> LDS after the barrier).
// Within a slice lane-half h=lane>>5 consumes k = 16h..16h+15, so the A fragment of a pixel row
// is 16 contiguous floats (4 x ds_read_b128, row stride 36 floats = conflict-free) and the B
// fragment is one ds_read_b32 per MFMA.  fp32 MFMA is exact fp32 (fmaf chain), 64 cycles per
// instruction, so 4 waves x 4 accumulators already saturate the matrix pipe; everything else
// hides behind it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "common.h"
#include "philox.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define LDA 36  // k-contiguous A tile row stride (floats): 16-B aligned, ds_read_b128 conflict-free

namespace {

struct Geom {               // how the pixel-indexed operand is gathered
    int H, W;               // physical source height/width
    int P, Q;               // pixel grid the GEMM rows (FWD) or K axis (WGRAD) enumerate
    int R, S, C;            // taps and channels per tap
    int stride, pad_t, pad_l;
    int shift, mask;        // logical index i is valid iff i>=0 && !(i&mask) && (i>>shift) < H
    long long s_n, s_h, s_w, s_c;
};

struct FwdParams {
    Geom g;
    const float* A;
    const float* B;
    const float* bias;
    const float* resid;
    float* D;
    int M, Ng, Kg;
    long long b_off, bs_r, bs_s, bs_c, bs_k;   // Wt(r,s,c,k) = B[b_off + r*bs_r + s*bs_s + c*bs_c + k*bs_k]
    long long ds_n, ds_p, ds_q, ds_k;          // D strides over (n,p,q,k)
    int relu;
    int relu_in;                               // A values pass through max(.,0) when staged (conv(relu(x)))
    const float* mask;                         // epilogue: keep the result only where mask[off] > 0 (ReLU backward)
    int d_lin;                                 // D offset = m*ds_q + col*ds_k (pixel-linear output)
    int d_vec;                                 // unit channel stride, 16-B aligned rows of D / mask / resid: float4 epilogue
    int resid_up;                              // resid is the dense channels-last [N, P/2, Q/2, Ng] tensor, read through a nearest-2x upsample
    unsigned a_bytes, b_bytes;                 // byte extents of A and B (buffer-descriptor range checks)
    int dbg;                                   // perf-diagnosis bits (env CTGAN_DBG): 1 no LDS store, 2 no global load, 4 no barrier
    // Output-phase decomposition of a stride-2 data gradient: dx pixels of parity (a,b) are a stride-1 conv of
    // dy with the taps of that parity only (no multiplies by the dilation zeros).  phases = 4: the M tile
    // index carries the phase; M / P / Q describe ONE phase grid; R,S = taps per phase (zero-padded to a common
    // count when R or S is odd).
    // dropout of the RESULT inside the epilogue (vector epilogue only): y *= floor(keep + u)/keep with u = element
    // off/4 .. of the Philox stream (seed, sid, ctr[0]) at the physical offset off of y - what ctgan_dropout_rng
    // applied to y would compute.  Used for the dropout that follows a conv (forward) and for the mask a data
    // gradient has to be multiplied with (backward).
    int drop; float drop_keep; unsigned long long drop_seed; unsigned drop_sid; const unsigned long long* drop_ctr;
    // row ranges with their own dropout (forward launches shared by several passes, functional.tape_record): rows
    // m < drop_mend[0] use range 0, ... ; keep >= 1 = no dropout in that range; the Philox element index is relative to the
    // range's first element (drop_roff), i.e. what a dropout on the range's own tensor would draw.  drop_nr = 0: one spec.
    int drop_nr; int drop_mend[CTGAN_DROP_RANGES]; float drop_rkeep[CTGAN_DROP_RANGES]; unsigned drop_rsid[CTGAN_DROP_RANGES];
    long long drop_roff[CTGAN_DROP_RANGES];
    int phases, ph_tiles_m;
    int ph_pad_t[2], ph_pad_l[2];              // top pad of row parity a / left pad of column parity b
    long long ph_b_stride;                     // filter elements per phase
    long long ph_d_h, ph_d_w;                  // D offset of phase (a,b) = a*ph_d_h + b*ph_d_w
};

struct PhaseSel { int pad_t, pad_l; long long b_off, d_off; };
__device__ __forceinline__ PhaseSel select_phase(const FwdParams& p, int& tile_m) {
    PhaseSel s{p.g.pad_t, p.g.pad_l, p.b_off, 0};
    if (p.phases > 1) {
        const int ph = tile_m / p.ph_tiles_m;
        tile_m -= ph * p.ph_tiles_m;
        const int a = ph >> 1, b = ph & 1;
        s.pad_t = p.ph_pad_t[a]; s.pad_l = p.ph_pad_l[b];
        s.b_off = p.b_off + ph * p.ph_b_stride;
        s.d_off = a * p.ph_d_h + b * p.ph_d_w;
    }
    return s;
}

struct WgradParams {
    Geom g;
    const float* X;
    const float* DY;
    float* OUT;              // [splits][Mtot][Ng] partial slabs (or the final dw when splits==1)
    int Mtot, Ng, Kg;        // R*S*C, K, N*P*Q
    long long dy_n, dy_p, dy_q, dy_k;
    int chunk;               // pixels per split (multiple of BK)
    int relu_x;              // A values pass through max(.,0) when staged (wgrad of conv(relu(x)))
    int with_bias;           // 1: slab row Mtot receives the column sums of dy (bias gradient)
    unsigned x_bytes, dy_bytes;
    // Multi-segment mode (pipelined kernel only): the pixel axis is the concatenation of up to CTGAN_WGRAD_MAX_SEGS
    // (x, dy) pairs of the same geometry - the uses of one filter in different passes of a step - so their weight
    // gradients come out of ONE launch, already summed.  A split never straddles segments: blockIdx.y in
    // [split0[s], split0[s+1]) works on segment s only.
    int nseg;
    struct Seg { const float* X; const float* DY; int Kg, split0, relu_x, bias; unsigned x_bytes, dy_bytes; } seg[CTGAN_WGRAD_MAX_SEGS];
};

__device__ __forceinline__ bool src_index(int i, int shift, int mask, int lim, int& o) {
    o = i >> shift;
    return (i >= 0) && !(i & mask) && (o < lim);
}

// ------------------------------------------------------------------------------------------
// MFMA over one 32-deep slice.  A tile k-contiguous [BM][LDA]; B tile [BK][BN].
template <int TM, int TN, int BN>
__device__ __forceinline__ void mma_slice_krow(const float* As, const float* Bs, int a_row0, int b_col0,
                                               int lane, f32x16 (&acc)[TM][TN]) {
    const int h = lane >> 5, l31 = lane & 31;
    float4 a[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            a[i][v] = *reinterpret_cast<const float4*>(&As[(a_row0 + i * 32 + l31) * LDA + h * 16 + v * 4]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float b[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[(h * 16 + s) * BN + b_col0 + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float4 av = a[i][s >> 2];
            const float ae = (s & 3) == 0 ? av.x : (s & 3) == 1 ? av.y : (s & 3) == 2 ? av.z : av.w;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b[j], acc[i][j], 0, 0, 0);
        }
    }
}

// A tile m-contiguous [BK][BM] (WGRAD)
template <int TM, int TN, int BM, int BN>
__device__ __forceinline__ void mma_slice_mrow(const float* As, const float* Bs, int a_row0, int b_col0,
                                               int lane, f32x16 (&acc)[TM][TN]) {
    const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[(h * 16 + s) * BM + a_row0 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[(h * 16 + s) * BN + b_col0 + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// FWD / DGRAD kernel
template <bool AVEC, bool BVEC, int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_fwd_kernel(const FwdParams p) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_VEC_PER = (BM * 8) / NT;       // float4 per thread (AVEC)
    constexpr int A_SCL_PER = (BM * BK) / NT;      // floats per thread (generic)
    constexpr int B_VEC_PER = (BK * BN / 4) / NT;
    constexpr int B_SCL_PER = (BK * BN) / NT;
    static_assert((BM * 8) % NT == 0 && (BK * BN / 4) % NT == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) float smem[BM * LDA + BK * BN];
    __shared__ long long row_off[BM];
    __shared__ int row_ih0[BM], row_iw0[BM];
    __shared__ long long ka_coff[2][BK], kb_off[2][BK];
    __shared__ int ka_r[2][BK], ka_s[2][BK];
    float* As = smem;
    float* Bs = smem + BM * LDA;

    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.Ng + BN - 1) / BN;
    int tile_m = blockIdx.x / tiles_n;
    const int tile_n = blockIdx.x - tile_m * tiles_n;
    const PhaseSel ph = select_phase(p, tile_m);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = (p.Kg + BK - 1) / BK;
    const int PQ = g.P * g.Q;

    for (int i = tid; i < BM; i += NT) {
        const int m = m0 + i;
        if (m < p.M) {
            const int n = m / PQ, rem = m - n * PQ, pp = rem / g.Q, qq = rem - pp * g.Q;
            row_off[i] = (long long)n * g.s_n;
            row_ih0[i] = pp * g.stride - ph.pad_t;
            row_iw0[i] = qq * g.stride - ph.pad_l;
        } else {
            row_off[i] = -1; row_ih0[i] = 0; row_iw0[i] = 0;
        }
    }
    auto fill_ktab = [&](int kt) {
        if (tid < BK) {
            const int gk = kt * BK + tid, buf = kt & 1;
            if (gk < p.Kg) {
                const int tap = gk / g.C, c = gk - tap * g.C, r = tap / g.S, s = tap - r * g.S;
                ka_r[buf][tid] = r; ka_s[buf][tid] = s;
                ka_coff[buf][tid] = (long long)c * g.s_c;
                kb_off[buf][tid] = ph.b_off + r * p.bs_r + s * p.bs_s + c * p.bs_c;
            } else {
                ka_r[buf][tid] = 0; ka_s[buf][tid] = 0; ka_coff[buf][tid] = -1; kb_off[buf][tid] = -1;
            }
        }
    };
    fill_ktab(0);
    __syncthreads();

    // per-thread A row cache (AVEC)
    long long a_off[AVEC ? A_VEC_PER : 1];
    int a_ih0[AVEC ? A_VEC_PER : 1], a_iw0[AVEC ? A_VEC_PER : 1];
    if constexpr (AVEC) {
#pragma unroll
        for (int i = 0; i < A_VEC_PER; ++i) {
            const int row = (tid >> 3) + i * (NT / 8);
            a_off[i] = row_off[row]; a_ih0[i] = row_ih0[row]; a_iw0[i] = row_iw0[row];
        }
    }

    float4 ra4[AVEC ? A_VEC_PER : 1];
    float ras[AVEC ? 1 : A_SCL_PER];
    float4 rb4[BVEC ? B_VEC_PER : 1];
    float rbs[BVEC ? 1 : B_SCL_PER];

    auto load_tile = [&](int kt) {
        const int buf = kt & 1;
        if constexpr (AVEC) {
            const int kk0 = kt * BK;               // C % 32 == 0: the slice lies inside one tap
            const int tap = kk0 / g.C, c0 = kk0 - tap * g.C, r = tap / g.S, s = tap - r * g.S;
            const int chunk = tid & 7;
#pragma unroll
            for (int i = 0; i < A_VEC_PER; ++i) {
                int ih, iw;
                const bool ok = (a_off[i] >= 0) & src_index(a_ih0[i] + r, g.shift, g.mask, g.H, ih) &
                                src_index(a_iw0[i] + s, g.shift, g.mask, g.W, iw);
                if (ok) ra4[i] = *reinterpret_cast<const float4*>(
                            p.A + a_off[i] + (long long)ih * g.s_h + (long long)iw * g.s_w + c0 + chunk * 4);
                else ra4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int kk = tid & 31;
            const int r = ka_r[buf][kk], s = ka_s[buf][kk];
            const long long coff = ka_coff[buf][kk];
#pragma unroll
            for (int i = 0; i < A_SCL_PER; ++i) {
                const int row = (tid >> 5) + i * (NT / 32);
                const long long ro = row_off[row];
                int ih, iw;
                const bool ok = (ro >= 0) & (coff >= 0) & src_index(row_ih0[row] + r, g.shift, g.mask, g.H, ih) &
                                src_index(row_iw0[row] + s, g.shift, g.mask, g.W, iw);
                ras[i] = ok ? p.A[ro + (long long)ih * g.s_h + (long long)iw * g.s_w + coff] : 0.f;
            }
        }
        if constexpr (BVEC) {
            constexpr int J4 = BN / 4;
            const int j4 = tid % J4;
#pragma unroll
            for (int i = 0; i < B_VEC_PER; ++i) {
                const int kk = tid / J4 + i * (NT / J4);
                const long long bo = kb_off[buf][kk];
                const int col = n0 + j4 * 4;
                if (bo >= 0 && col < p.Ng) rb4[i] = *reinterpret_cast<const float4*>(p.B + bo + col);
                else rb4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int j = tid % BN;
#pragma unroll
            for (int i = 0; i < B_SCL_PER; ++i) {
                const int kk = tid / BN + i * (NT / BN);
                const long long bo = kb_off[buf][kk];
                const int col = n0 + j;
                rbs[i] = (bo >= 0 && col < p.Ng) ? p.B[bo + (long long)col * p.bs_k] : 0.f;
            }
        }
    };
    auto store_tile = [&]() {
        if constexpr (AVEC) {
            const int chunk = tid & 7;
#pragma unroll
            for (int i = 0; i < A_VEC_PER; ++i) {
                const int row = (tid >> 3) + i * (NT / 8);
                float4 v = ra4[i];
                if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(&As[row * LDA + chunk * 4]) = v;
            }
        } else {
            const int kk = tid & 31;
#pragma unroll
            for (int i = 0; i < A_SCL_PER; ++i) As[((tid >> 5) + i * (NT / 32)) * LDA + kk] = p.relu_in ? fmaxf(ras[i], 0.f) : ras[i];
        }
        if constexpr (BVEC) {
            constexpr int J4 = BN / 4;
            const int j4 = tid % J4;
#pragma unroll
            for (int i = 0; i < B_VEC_PER; ++i)
                *reinterpret_cast<float4*>(&Bs[(tid / J4 + i * (NT / J4)) * BN + j4 * 4]) = rb4[i];
        } else {
            const int j = tid % BN;
#pragma unroll
            for (int i = 0; i < B_SCL_PER; ++i) Bs[(tid / BN + i * (NT / BN)) * BN + j] = rbs[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_tile(0);
    store_tile();
    if (nk > 1) fill_ktab(1);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
        mma_slice_krow<TM, TN, BN>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
        __syncthreads();
        if (kt + 1 < nk) store_tile();
        if (kt + 2 < nk) fill_ktab(kt + 2);
        __syncthreads();
    }

    // epilogue: acc[i][j][e] -> row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31
    const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        if (col >= p.Ng) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int m = m0 + row;
                if (m >= p.M) continue;
                const int n = m / PQ, rem = m - n * PQ, pp = rem / g.Q, qq = rem - pp * g.Q;
                const long long off = ph.d_off + n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col * p.ds_k;
                float v = acc[i][j][e] + bv;
                if (p.mask && !(p.mask[off] > 0.f)) v = 0.f;
                if (p.resid) v += p.resid[off];
                if (p.relu) v = fmaxf(v, 0.f);
                p.D[off] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Pipelined FWD / DGRAD kernel for the fully vectorisable case (C % (32*WAVES_K) == 0, unit channel
// stride, 16-B aligned rows; filter rows contiguous in k).  Differences from the generic kernel:
//   * two LDS stages, ONE barrier per K slice: while the MFMAs of slice t run, the registers
//     holding slice t+1 are written to the other stage and the global loads of slice t+2 are issued;
//   * B fragments are read PD steps ahead of the MFMA that consumes them (register ring), so a
//     single wave per SIMD keeps the matrix pipe busy;
//   * no LDS lookup tables or exec-masked loads in the loop: padding taps read a clamped address
//     and are zeroed by a select when staged;
//   * WAVES_K > 1 splits each slice between wave groups (for small M: more waves than output
//     tiles), partial accumulators are combined through LDS at the end;
//   * XCD-aware tile order: consecutive M tiles (which share halo rows) land on the same XCD / L2.
// occupancy the register allocator must keep (waves per SIMD): the 64x128 tile (2 accumulators per wave) sits at the edge of
// 3 - a few VGPRs more in the epilogue cost a whole wave (measured: -2 % on the step).  Only the 4-phase variant needs the
// hint; with it the allocator moves the accumulators out of the AGPRs, which costs the plain variant 4 %.
#ifndef CTGAN_B_DIRECT
#define CTGAN_B_DIRECT 0
#endif
// EXPERIMENT (off: -DCTGAN_B_DIRECT=1 to build it; parity tests pass).  Measured on the step: the 128x128 / 64x128 tiles do not
// change (118.6 vs 118.7, 105.6 vs 106.2 TFLOP/s - their pipe-busy gap is not the register staging of B), the K-split tiles of the
// small layers lose 3-6 % (one slice of latency slack less than the register ring).
// B operand (filter slices) straight into LDS: buffer_load_dwordx4 ... lds writes lane l's 16 bytes at M0 + 16*l, which is
// exactly the row-major [BK][BN] layout of the B stage when a wave covers 64/BC consecutive rows - no VGPR staging, no
// ds_write.  Inline asm: the compiler's waitcnt insertion must not see it (it would drain vmcnt before every ds_read);
// completion is awaited by hand before the slice barrier (loads retire in order: vmcnt(A loads issued after it)).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void buffer_load16_to_lds(u32x4_t rsrc, unsigned voff, unsigned soff, unsigned lds_byte_off /* wave-uniform */) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_off)
                 : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() {
    static_assert(N >= 0 && N < 16, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt(0xF70 | N);          // lgkmcnt / expcnt untouched, vmcnt <= N (N < 16: low field only)
}
constexpr int fwd_pipe_min_waves(int waves_k, int tm, int tn, int ksub) { return 1; }
template <int WAVES_M, int WAVES_N, int WAVES_K, int TM, int TN, int RD, bool RELU_IN, int KSUB>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WAVES_K, fwd_pipe_min_waves(WAVES_K, TM, TN, KSUB)) void igemm_fwd_pipe_kernel(const FwdParams p) {
    constexpr int NT = 64 * WAVES_M * WAVES_N * WAVES_K;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32, BKE = 32 * WAVES_K * KSUB, LDAE = BKE + 4;   // KSUB 32-deep slices per wave per stage
    constexpr int STAGE = BM * LDAE + BKE * BN;
    constexpr int AC = BKE / 4, BC = BN / 4;
    constexpr int A_PER = (BM * AC) / NT, B_PER = (BKE * BC) / NT;
    constexpr int PD = (TM * TN >= 4) ? 1 : (TM * TN == 2 ? 2 : 4);      // B prefetch distance (MFMA steps)
    static_assert((BM * AC) % NT == 0 && (BKE * BC) % NT == 0 && A_PER >= 1 && B_PER >= 1, "tile/threads mismatch");
    static_assert(NT % AC == 0 && NT % BC == 0, "loader mapping");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WAVES_K, wmn = wave / WAVES_K;
    const int wm = wmn / WAVES_N, wn = wmn % WAVES_N;
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0 && !(p.dbg & 16)) bid = (bid & 7) * (nb >> 3) + (bid >> 3);         // block b runs on XCD b%8 (observed)
    const int tiles_n = (p.Ng + BN - 1) / BN;
    int tile_m = bid / tiles_n;
    const int tile_n = bid - tile_m * tiles_n;
    const PhaseSel ph = select_phase(p, tile_m);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.Kg / BKE;
    const int cpt = g.C / BKE;
    const int PQ = g.P * g.Q;

    // Loader: buffer loads through wave-uniform resource descriptors.  The per-lane byte offset
    // (voffset) of a pixel row is fixed for the whole kernel when the gather is an affine map
    // (no upsample / dilation); the tap and channel-chunk displacement of a K slice is wave-uniform
    // and rides in the scalar offset.  Padding taps use voffset = ~0u: the hardware range check
    // (offset >= num_records) returns zeros, so no select and no branch is needed.
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);
    const bool affine = (g.shift | g.mask) == 0;
    const int a_chunk = tid % AC, a_row0 = tid / AC;
    unsigned a_voff[A_PER];          // affine: full byte offset of (row, tap 0, chunk); else byte offset of (n, chunk)
    int a_ih0[A_PER], a_iw0[A_PER];
    bool a_valid[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + a_row0 + i * (NT / AC);
        a_valid[i] = m < p.M;
        const int mm = a_valid[i] ? m : 0;
        const int n = mm / PQ, rem = mm - n * PQ, pp = rem / g.Q, qq = rem - pp * g.Q;
        a_ih0[i] = pp * g.stride - ph.pad_t;
        a_iw0[i] = qq * g.stride - ph.pad_l;
        long long o = (long long)n * g.s_n + a_chunk * 4;
        if (affine) o += (long long)a_ih0[i] * g.s_h + (long long)a_iw0[i] * g.s_w;   // may be "negative": wraps consistently mod 2^32
        a_voff[i] = (unsigned)(o * 4);
    }
    const int b_j4 = tid % BC, b_k0 = tid / BC;
    const bool b_ok = (n0 + b_j4 * 4) < p.Ng;
    unsigned b_voff[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
        b_voff[i] = b_ok ? (unsigned)(((long long)(b_k0 + i * (NT / BC)) * p.bs_c + b_j4 * 4) * 4) : 0xFFFFFFFFu;

    float4 ra[RD][A_PER], rb[RD][B_PER];                                // register ring: slices are loaded RD iterations ahead
    int ld_r = 0, ld_s = 0, ld_c = 0;                                   // tap / channel-chunk of the NEXT slice to load
#if CTGAN_B_DIRECT
    int lb_r = 0, lb_s = 0, lb_c = 0;                                   // ... and of the next B slice sent straight to LDS
    u32x4_t b_desc;
    {
        const unsigned long long ba = reinterpret_cast<unsigned long long>(p.B);
        b_desc.x = (unsigned)ba; b_desc.y = (unsigned)(ba >> 32); b_desc.z = p.b_bytes; b_desc.w = 0x00020000u;
    }
    const unsigned b_lds_wave = (unsigned)__builtin_amdgcn_readfirstlane((wave * (64 / BC)) * BN * 4);
    auto issue_b_direct = [&](float* Bs) {
        const unsigned bsoff = (unsigned)__builtin_amdgcn_readfirstlane((int)((ph.b_off + n0 + lb_r * p.bs_r + lb_s * p.bs_s + (long long)(lb_c * BKE) * p.bs_c) * 4));
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)Bs + b_lds_wave;
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            buffer_load16_to_lds(b_desc, b_voff[i], bsoff, __builtin_amdgcn_readfirstlane(base + (unsigned)(i * (NT / BC) * BN * 4)));
        if (++lb_c == cpt) { lb_c = 0; if (++lb_s == g.S) { lb_s = 0; ++lb_r; } }
    };
#endif

    auto load_tile = [&](float4 (&ra)[A_PER], float4 (&rb)[B_PER]) {
        const int c0 = ld_c * BKE;
        if (affine) {
            const unsigned soff = (unsigned)(((long long)ld_r * g.s_h + (long long)ld_s * g.s_w + c0) * 4);
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const bool ok = a_valid[i] & ((unsigned)(a_ih0[i] + ld_r) < (unsigned)g.H) & ((unsigned)(a_iw0[i] + ld_s) < (unsigned)g.W);
                const unsigned vo = ok ? a_voff[i] + soff : 0xFFFFFFFFu;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, vo, 0, 0);
                ra[i] = __builtin_bit_cast(float4, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                int ih, iw;
                const bool ok = a_valid[i] & src_index(a_ih0[i] + ld_r, g.shift, g.mask, g.H, ih) &
                                src_index(a_iw0[i] + ld_s, g.shift, g.mask, g.W, iw);
                const unsigned vo = ok ? a_voff[i] + (unsigned)(((long long)ih * g.s_h + (long long)iw * g.s_w + c0) * 4) : 0xFFFFFFFFu;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, vo, 0, 0);
                ra[i] = __builtin_bit_cast(float4, v);
            }
        }
#if !CTGAN_B_DIRECT
        const unsigned bsoff = (unsigned)((ph.b_off + n0 + ld_r * p.bs_r + ld_s * p.bs_s + (long long)c0 * p.bs_c) * 4);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_voff[i], bsoff, 0);
            rb[i] = __builtin_bit_cast(float4, v);
        }
#endif
        if (++ld_c == cpt) { ld_c = 0; if (++ld_s == g.S) { ld_s = 0; ++ld_r; } }
    };
    auto store_tile = [&](const float4 (&ra)[A_PER], const float4 (&rb)[B_PER], float* As, float* Bs) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            float4 v = ra[i];
            if constexpr (RELU_IN) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(&As[(a_row0 + i * (NT / AC)) * LDAE + a_chunk * 4]) = v;
        }
#if !CTGAN_B_DIRECT
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            *reinterpret_cast<float4*>(&Bs[(b_k0 + i * (NT / BC)) * BN + b_j4 * 4]) = rb[i];
#endif
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: slice 0 -> stage 0; slices 1..RD -> ring slots (slice t lives in slot t % RD)
#if CTGAN_B_DIRECT
    issue_b_direct(smem + BM * LDAE);
#endif
    load_tile(ra[0], rb[0]);
    store_tile(ra[0], rb[0], smem, smem + BM * LDAE);
#pragma unroll
    for (int t = 1; t <= RD; ++t)
        if (t < nk) load_tile(ra[t % RD], rb[t % RD]);
#if CTGAN_B_DIRECT
    wait_vmcnt_le<0>();
#endif
    __syncthreads();

    const int h = lane >> 5, l31 = lane & 31;
    const int a_rd0 = (wm * TM * 32 + l31) * LDAE + wk * KSUB * 32 + h * 16;   // + sub*32 + i*32*LDAE + v*4
    const int b_rd0 = (wk * KSUB * 32 + h * 16) * BN + wn * TN * 32 + l31;     // + sub*32*BN + s*BN + j*32

    for (int kt0 = 0; kt0 < nk; kt0 += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;
            constexpr int dummy = 0; (void)dummy;
            const float* As = smem + (kt & 1) * STAGE;
            const float* Bs = As + BM * LDAE;
            float* Asn = smem + ((kt + 1) & 1) * STAGE;
            float* Bsn = Asn + BM * LDAE;
#pragma unroll
            for (int sub = 0; sub < KSUB; ++sub) {
            const int a_rd = a_rd0 + sub * 32, b_rd = b_rd0 + sub * 32 * BN;
            float4 a[TM][4];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) a[i][v] = *reinterpret_cast<const float4*>(&As[a_rd + i * 32 * LDAE + v * 4]);
            float b[PD + 1][TN];
#pragma unroll
            for (int s0 = 0; s0 < PD; ++s0)
#pragma unroll
                for (int j = 0; j < TN; ++j) b[s0][j] = Bs[b_rd + s0 * BN + j * 32];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + PD < 16) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[(s + PD) % (PD + 1)][j] = Bs[b_rd + (s + PD) * BN + j * 32];
                }
                // keep the prefetch ABOVE this step's MFMAs: hipcc otherwise sinks the ds_read to just before its
                // use and every group of MFMAs then starts with an exposed LDS round trip
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float4 av = a[i][s >> 2];
                    const float ae = (s & 3) == 0 ? av.x : (s & 3) == 1 ? av.y : (s & 3) == 2 ? av.z : av.w;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, b[s % (PD + 1)][j], acc[i][j], 0, 0, 0);
                }
                // slot (u+1)%RD holds slice kt+1 (loaded RD iterations ago): stage it, then refill the slot
                if (sub == 0 && s == 0 && kt + 1 < nk && !(p.dbg & 1)) {
                    store_tile(ra[(u + 1) % RD], rb[(u + 1) % RD], Asn, Bsn);
#if CTGAN_B_DIRECT
                    issue_b_direct(Bsn);      // after the A store: its compiler-inserted vmcnt(0) must not cover these loads
#endif
                }
                if (sub == 0 && s == 1 && kt + 1 + RD < nk && !(p.dbg & 2)) load_tile(ra[(u + 1) % RD], rb[(u + 1) % RD]);
                // waves that are in their MFMA stretch win arbitration over a co-resident wave that is staging
                if (sub == 0 && s == 1) __builtin_amdgcn_s_setprio(1);
                if (sub == KSUB - 1 && s == 15) __builtin_amdgcn_s_setprio(0);
            }
            }
#if CTGAN_B_DIRECT
            if (kt + 1 + RD < nk) wait_vmcnt_le<(A_PER < 15 ? A_PER : 0)>(); else wait_vmcnt_le<0>();   // the B slice landed; A registers may still fly
#endif
            if (!(p.dbg & 4)) __syncthreads();
        }
    }

    if constexpr (WAVES_K > 1) {                                          // combine the K groups through LDS: one round
        float* red = smem;
        constexpr int PER_WAVE = TM * TN * 16 * 64;
        constexpr int GROUPS_MN = WAVES_M * WAVES_N;
        static_assert((size_t)(WAVES_K - 1) * GROUPS_MN * PER_WAVE <= 2 * (size_t)STAGE, "K-group reduction buffer must fit the staging buffers");
        if (wk != 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        red[((wk - 1) * GROUPS_MN + wmn) * PER_WAVE + ((i * TN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int r = 1; r < WAVES_K; ++r)                             // fixed order r = 1, 2, 3: deterministic
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            acc[i][j][e] += red[((r - 1) * GROUPS_MN + wmn) * PER_WAVE + ((i * TN + j) * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }

    if (p.d_vec) {
        // Vector epilogue: the accumulators go through LDS (free after the K loop) so that every lane owns 4
        // consecutive channels of one pixel: 16-B stores, and the mask / residual operands are fetched as 16-B
        // loads that are all in flight before the first store (no load->store serialisation).
        constexpr int LDC = BN + 8;                 // +8: the two lane halves (rows r, r+4) hit disjoint banks
        constexpr int C4 = BN / 4, ROWS_PER = NT / C4, ITERS = BM / ROWS_PER;
        static_assert(NT % C4 == 0 && BM % ROWS_PER == 0, "epilogue mapping");
        static_assert(BM * LDC <= 2 * STAGE, "epilogue tile must fit the staging buffers");
        float* ct = smem;
        if constexpr (WAVES_K == 1) __syncthreads();        // (the K-group reduction above ends with a barrier)
        if (wk == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        ct[(wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + wn * TN * 32 + j * 32 + l31] = acc[i][j][e];
        }
        __syncthreads();
        const int c4 = tid % C4, r0 = tid / C4;
        const int col = n0 + c4 * 4;
        const uint64_t drop_step = p.drop ? (p.drop_ctr ? p.drop_ctr[0] : 0) : 0;
        // dropout row ranges start at multiples of the largest M tile (checked by the host), so the range is workgroup-uniform
        float dkeep = p.drop_keep;
        unsigned dsid = p.drop_sid;
        unsigned doff4 = 0;                                  // first float4 index of the range (the counter is 32 bits wide anyway)
        if (p.drop_nr) {                                     // scalar selects, no dynamic indexing of the argument arrays
            const bool r1 = p.drop_nr > 1 && m0 >= p.drop_mend[0], r2 = p.drop_nr > 2 && m0 >= p.drop_mend[1];
            dkeep = r2 ? p.drop_rkeep[2] : (r1 ? p.drop_rkeep[1] : p.drop_rkeep[0]);
            dsid = r2 ? p.drop_rsid[2] : (r1 ? p.drop_rsid[1] : p.drop_rsid[0]);
            doff4 = (unsigned)((r2 ? p.drop_roff[2] : (r1 ? p.drop_roff[1] : p.drop_roff[0])) >> 2);
        }
        const bool do_drop = p.drop && dkeep < 1.f;
        if (col < p.Ng) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + col);
            constexpr int BATCH = ITERS < 8 ? ITERS : 8;
#pragma unroll
            for (int b0 = 0; b0 < ITERS; b0 += BATCH) {
                long long off[BATCH];
                float4 mv[BATCH], rv[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int m = m0 + r0 + (b0 + u) * ROWS_PER;
                    off[u] = -1;
                    if (m < p.M) {
                        if (p.d_lin) {
                            off[u] = (long long)m * p.ds_q + col;
                        } else {
                            const int n = m / PQ, rem = m - n * PQ, pp = rem / g.Q, qq = rem - pp * g.Q;
                            off[u] = ph.d_off + n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col;
                        }
                        if (p.mask) mv[u] = *reinterpret_cast<const float4*>(p.mask + off[u]);
                        if (p.resid) {
                            long long ro = off[u];
                            if (p.resid_up) {
                                const int n = m / PQ, rem = m - n * PQ, pp = rem / g.Q, qq = rem - pp * g.Q;
                                ro = (((long long)n * (g.P >> 1) + (pp >> 1)) * (g.Q >> 1) + (qq >> 1)) * p.Ng + col;
                            }
                            rv[u] = *reinterpret_cast<const float4*>(p.resid + ro);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    if (off[u] < 0) continue;
                    float4 v = *reinterpret_cast<const float4*>(&ct[(r0 + (b0 + u) * ROWS_PER) * LDC + c4 * 4]);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (p.mask) {
                        v.x = mv[u].x > 0.f ? v.x : 0.f; v.y = mv[u].y > 0.f ? v.y : 0.f;
                        v.z = mv[u].z > 0.f ? v.z : 0.f; v.w = mv[u].w > 0.f ? v.w : 0.f;
                    }
                    if (p.resid) { v.x += rv[u].x; v.y += rv[u].y; v.z += rv[u].z; v.w += rv[u].w; }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (do_drop) {
                        uint32_t c[4];
                        ctgan_philox::draw4(p.drop_seed, dsid, drop_step, (uint32_t)(off[u] >> 2) - doff4, c);
                        const float inv = 1.f / dkeep;
                        v.x *= inv * floorf(dkeep + ctgan_philox::u01(c[0])); v.y *= inv * floorf(dkeep + ctgan_philox::u01(c[1]));
                        v.z *= inv * floorf(dkeep + ctgan_philox::u01(c[2])); v.w *= inv * floorf(dkeep + ctgan_philox::u01(c[3]));
                    }
                    *reinterpret_cast<float4*>(p.D + off[u]) = v;
                }
            }
        }
        return;
    }

    if (wk != 0) return;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        if (col >= p.Ng) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                long long off;
                if (p.d_lin) {
                    off = (long long)m * p.ds_q + col * p.ds_k;
                } else {
                    const int n = m / PQ, rem = m - n * PQ, pp = rem / g.Q, qq = rem - pp * g.Q;
                    off = ph.d_off + n * p.ds_n + pp * p.ds_p + qq * p.ds_q + col * p.ds_k;
                }
                float v = acc[i][j][e] + bv;
                if (p.mask && !(p.mask[off] > 0.f)) v = 0.f;
                if (p.resid) v += p.resid[off];
                if (p.relu) v = fmaxf(v, 0.f);
                p.D[off] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// WGRAD kernel: grid = (tiles_m * tiles_n, splits)
template <bool AVEC, bool BVEC, int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_wgrad_kernel(const WgradParams p) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_VEC_PER = (BK * BM / 4) / NT, A_SCL_PER = (BK * BM) / NT;
    constexpr int B_VEC_PER = (BK * BN / 4) / NT, B_SCL_PER = (BK * BN) / NT;

    __shared__ __attribute__((aligned(16))) float smem[BK * BM + BK * BN];
    __shared__ long long px_xoff[2][BK], px_yoff[2][BK];
    __shared__ int px_ih0[2][BK], px_iw0[2][BK];
    float* As = smem;
    float* Bs = smem + BK * BM;

    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.Ng + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = blockIdx.y * p.chunk;
    const int k_end = min(p.Kg, k_begin + p.chunk);
    const int nk = (k_end - k_begin + BK - 1) / BK;
    const int PQ = g.P * g.Q;

    auto fill_ptab = [&](int kt) {
        if (tid < BK) {
            const int buf = kt & 1, px = k_begin + kt * BK + tid;
            if (px < k_end) {
                const int n = px / PQ, rem = px - n * PQ, pp = rem / g.Q, qq = rem - pp * g.Q;
                px_xoff[buf][tid] = (long long)n * g.s_n;
                px_ih0[buf][tid] = pp * g.stride - g.pad_t;
                px_iw0[buf][tid] = qq * g.stride - g.pad_l;
                px_yoff[buf][tid] = n * p.dy_n + pp * p.dy_p + qq * p.dy_q;
            } else {
                px_xoff[buf][tid] = -1; px_yoff[buf][tid] = -1; px_ih0[buf][tid] = 0; px_iw0[buf][tid] = 0;
            }
        }
    };

    // A-side row (= filter element) decode
    int ar = 0, as_ = 0;                   // AVEC: block-uniform tap
    long long acoff = 0;                   // AVEC: channel offset of this thread's float4
    int gr = 0, gs = 0; long long gcoff = -1;   // generic: this thread's fixed filter element
    if constexpr (AVEC) {
        const int tap = m0 / g.C, c0 = m0 - tap * g.C;   // C % BM == 0
        ar = tap / g.S; as_ = tap - ar * g.S;
        acoff = c0 + (tid % (BM / 4)) * 4;
    } else {
        const int m = m0 + tid % BM;
        if (m < p.Mtot) {
            const int tap = m / g.C, c = m - tap * g.C;
            gr = tap / g.S; gs = tap - gr * g.S; gcoff = (long long)c * g.s_c;
        }
    }

    float4 ra4[AVEC ? A_VEC_PER : 1];
    float ras[AVEC ? 1 : A_SCL_PER];
    float4 rb4[BVEC ? B_VEC_PER : 1];
    float rbs[BVEC ? 1 : B_SCL_PER];

    auto load_tile = [&](int kt) {
        const int buf = kt & 1;
        if constexpr (AVEC) {
            constexpr int M4 = BM / 4;
#pragma unroll
            for (int i = 0; i < A_VEC_PER; ++i) {
                const int kk = tid / M4 + i * (NT / M4);
                const long long xo = px_xoff[buf][kk];
                int ih, iw;
                const bool ok = (xo >= 0) & src_index(px_ih0[buf][kk] + ar, g.shift, g.mask, g.H, ih) &
                                src_index(px_iw0[buf][kk] + as_, g.shift, g.mask, g.W, iw);
                if (ok) ra4[i] = *reinterpret_cast<const float4*>(
                            p.X + xo + (long long)ih * g.s_h + (long long)iw * g.s_w + acoff);
                else ra4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_SCL_PER; ++i) {
                const int kk = tid / BM + i * (NT / BM);
                const long long xo = px_xoff[buf][kk];
                int ih, iw;
                const bool ok = (xo >= 0) & (gcoff >= 0) & src_index(px_ih0[buf][kk] + gr, g.shift, g.mask, g.H, ih) &
                                src_index(px_iw0[buf][kk] + gs, g.shift, g.mask, g.W, iw);
                ras[i] = ok ? p.X[xo + (long long)ih * g.s_h + (long long)iw * g.s_w + gcoff] : 0.f;
            }
        }
        if constexpr (BVEC) {
            constexpr int J4 = BN / 4;
            const int j4 = tid % J4;
#pragma unroll
            for (int i = 0; i < B_VEC_PER; ++i) {
                const int kk = tid / J4 + i * (NT / J4);
                const long long yo = px_yoff[buf][kk];
                const int col = n0 + j4 * 4;
                if (yo >= 0 && col < p.Ng) rb4[i] = *reinterpret_cast<const float4*>(p.DY + yo + col);
                else rb4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int j = tid % BN;
#pragma unroll
            for (int i = 0; i < B_SCL_PER; ++i) {
                const int kk = tid / BN + i * (NT / BN);
                const long long yo = px_yoff[buf][kk];
                const int col = n0 + j;
                rbs[i] = (yo >= 0 && col < p.Ng) ? p.DY[yo + (long long)col * p.dy_k] : 0.f;
            }
        }
    };
    auto store_tile = [&]() {
        if constexpr (AVEC) {
            constexpr int M4 = BM / 4;
#pragma unroll
            for (int i = 0; i < A_VEC_PER; ++i) {
                float4 v = ra4[i];
                if (p.relu_x) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(&As[(tid / M4 + i * (NT / M4)) * BM + (tid % M4) * 4]) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_SCL_PER; ++i) As[(tid / BM + i * (NT / BM)) * BM + tid % BM] = p.relu_x ? fmaxf(ras[i], 0.f) : ras[i];
        }
        if constexpr (BVEC) {
            constexpr int J4 = BN / 4;
#pragma unroll
            for (int i = 0; i < B_VEC_PER; ++i)
                *reinterpret_cast<float4*>(&Bs[(tid / J4 + i * (NT / J4)) * BN + (tid % J4) * 4]) = rb4[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_SCL_PER; ++i) Bs[(tid / BN + i * (NT / BN)) * BN + tid % BN] = rbs[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nk > 0) {
        fill_ptab(0);
        __syncthreads();
        load_tile(0);
        store_tile();
        if (nk > 1) fill_ptab(1);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1);
            mma_slice_mrow<TM, TN, BM, BN>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
            __syncthreads();
            if (kt + 1 < nk) store_tile();
            if (kt + 2 < nk) fill_ptab(kt + 2);
            __syncthreads();
        }
    }

    const int h = lane >> 5, l31 = lane & 31;
    float* out = p.OUT + (long long)blockIdx.y * p.Mtot * p.Ng;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        if (col >= p.Ng) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < p.Mtot) out[(long long)m * p.Ng + col] = acc[i][j][e];
            }
    }
}

// ------------------------------------------------------------------------------------------
// Pipelined WGRAD kernel (vector case: the M tile lies inside one filter tap, unit channel strides).
// Same structure as igemm_fwd_pipe_kernel: two LDS stages / one barrier per 32-pixel slice, buffer
// loads with hardware zero fill, operand prefetch ring.  Per-pixel gather offsets for slice t+3 are
// produced by 32 lanes while slice t is multiplied (4-deep table ring).  Workgroups of the first
// M tile also accumulate the column sums of the dy tiles they stage: the bias gradient rides along
// as slab row Mtot at no extra HBM traffic.
template <int WAVES_M, int WAVES_N, int TM, int TN>
__device__ __forceinline__ void wgrad_pipe_body(const WgradParams& p, const int bx, const int by) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int STAGE = BK * BM + BK * BN;
    constexpr int M4 = BM / 4, J4 = BN / 4;
    constexpr int A_PER = (BK * M4) / NT, B_PER = (BK * J4) / NT;
    constexpr int PD = (TM * TN >= 4) ? 1 : (TM * TN == 2 ? 2 : 4);
    static_assert((BK * M4) % NT == 0 && (BK * J4) % NT == 0 && A_PER >= 1 && B_PER >= 1, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned px_x[4][BK], px_y[4][BK];

    const Geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.Ng + BN - 1) / BN;
    const int tile_m = bx / tiles_n, tile_n = bx - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float* Xp = p.X; const float* DYp = p.DY;
    int seg_kg = p.Kg, split = by, relu_x = p.relu_x, seg_bias = 1;
    unsigned x_bytes = p.x_bytes, dy_bytes = p.dy_bytes;
    if (p.nseg > 0) {
        int si = 0;
#pragma unroll
        for (int t = 1; t < CTGAN_WGRAD_MAX_SEGS; ++t)
            if (t < p.nseg && by >= p.seg[t].split0) si = t;
        Xp = p.seg[si].X; DYp = p.seg[si].DY; seg_kg = p.seg[si].Kg; split = by - p.seg[si].split0;
        relu_x = p.seg[si].relu_x; seg_bias = p.seg[si].bias; x_bytes = p.seg[si].x_bytes; dy_bytes = p.seg[si].dy_bytes;
    }
    const int k_begin = split * p.chunk;
    const int k_end = min(seg_kg, k_begin + p.chunk);
    const int nk = (k_end - k_begin + BK - 1) / BK;
    const int PQ = g.P * g.Q;
    const int tap = m0 / g.C, c0 = m0 - tap * g.C;              // C % BM == 0
    const int ar = tap / g.S, as_ = tap - ar * g.S;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xp), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DYp), 0, dy_bytes, 0x00020000);

    const bool pow2 = ((PQ & (PQ - 1)) | (g.Q & (g.Q - 1))) == 0;      // the per-pixel table is filled by one wave: keep it short
    const int pq_sh = __builtin_ctz(PQ), q_sh = __builtin_ctz(g.Q);
    auto fill_ptab = [&](int kt) {
        if (tid < BK) {
            const int px = k_begin + kt * BK + tid;
            unsigned xo = 0xFFFFFFFFu, yo = 0xFFFFFFFFu;
            if (px < k_end) {
                int n, rem, pp, qq;
                if (pow2) { n = px >> pq_sh; rem = px & (PQ - 1); pp = rem >> q_sh; qq = rem & (g.Q - 1); }     // no integer divisions
                else { n = px / PQ; rem = px - n * PQ; pp = rem / g.Q; qq = rem - pp * g.Q; }
                int ih, iw;
                const bool ok = src_index(pp * g.stride - g.pad_t + ar, g.shift, g.mask, g.H, ih) &
                                src_index(qq * g.stride - g.pad_l + as_, g.shift, g.mask, g.W, iw);
                if (ok) xo = (unsigned)(((long long)n * g.s_n + (long long)ih * g.s_h + (long long)iw * g.s_w) * 4);
                yo = (unsigned)((n * p.dy_n + pp * p.dy_p + qq * p.dy_q) * 4);
            }
            px_x[kt & 3][tid] = xo; px_y[kt & 3][tid] = yo;
        }
    };

    const int a_m4 = tid % M4, a_k0 = tid / M4;
    const int b_j4 = tid % J4, b_k0 = tid / J4;
    const unsigned a_const = (unsigned)((c0 + a_m4 * 4) * 4);
    const bool b_ok = (n0 + b_j4 * 4) < p.Ng;
    const unsigned b_const = (unsigned)((n0 + b_j4 * 4) * 4);
    float4 ra[A_PER], rb[B_PER];

    auto load_tile = [&](int kt) {
        const int ring = kt & 3;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const unsigned xo = px_x[ring][a_k0 + i * (NT / M4)];
            const unsigned vo = xo == 0xFFFFFFFFu ? xo : xo + a_const;
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, vo, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const unsigned yo = px_y[ring][b_k0 + i * (NT / J4)];
            const unsigned vo = (yo == 0xFFFFFFFFu || !b_ok) ? 0xFFFFFFFFu : yo + b_const;
            rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, vo, 0, 0));
        }
    };
    auto store_tile = [&](float* As, float* Bs) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            float4 v = ra[i];
            if (relu_x) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(&As[(a_k0 + i * (NT / M4)) * BM + a_m4 * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) *reinterpret_cast<float4*>(&Bs[(b_k0 + i * (NT / J4)) * BN + b_j4 * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum = 0.f;
    const bool bias_row = p.with_bias && tile_m == 0 && tid < BN;       // this thread owns a column of the slab's bias row
    const bool do_bias = bias_row && seg_bias;                           // ... and this segment's dy contributes to it

    if (nk > 0) {
        fill_ptab(0);
        if (nk > 1) fill_ptab(1);
        if (nk > 2) fill_ptab(2);
        __syncthreads();
        load_tile(0);
        store_tile(smem, smem + BK * BM);
        if (nk > 1) load_tile(1);
        __syncthreads();
    }
    const int h = lane >> 5, l31 = lane & 31;
    const int a_rd = h * 16 * BM + wm * TM * 32 + l31;
    const int b_rd = h * 16 * BN + wn * TN * 32 + l31;

    for (int kt = 0; kt < nk; ++kt) {
        const float* As = smem + (kt & 1) * STAGE;
        const float* Bs = As + BK * BM;
        float* Asn = smem + ((kt + 1) & 1) * STAGE;
        float* Bsn = Asn + BK * BM;
        float a[PD + 1][TM], b[PD + 1][TN];
#pragma unroll
        for (int s0 = 0; s0 < PD; ++s0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[s0][i] = As[a_rd + s0 * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[s0][j] = Bs[b_rd + s0 * BN + j * 32];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s + PD < 16) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(s + PD) % (PD + 1)][i] = As[a_rd + (s + PD) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(s + PD) % (PD + 1)][j] = Bs[b_rd + (s + PD) * BN + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);     // prefetch stays above this step's MFMAs (see igemm_fwd_pipe_kernel)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s % (PD + 1)][i], b[s % (PD + 1)][j], acc[i][j], 0, 0, 0);
            if (s == 3 && kt + 1 < nk) store_tile(Asn, Bsn);
            if (s == 4 && kt + 2 < nk) load_tile(kt + 2);      // right after the store freed the registers: the longest latency budget
            if (s == 9 && kt + 3 < nk) fill_ptab(kt + 3);
            if (s == 11 && do_bias) {
                float t = 0.f;
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) t += Bs[kk * BN + tid];
                bsum += t;
            }
        }
        __syncthreads();
    }

    float* out = p.OUT + (long long)by * (p.Mtot + p.with_bias) * p.Ng;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + l31;
        if (col >= p.Ng) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < p.Mtot) out[(long long)m * p.Ng + col] = acc[i][j][e];
            }
    }
    if (bias_row && n0 + tid < p.Ng) out[(long long)p.Mtot * p.Ng + n0 + tid] = bsum;
}


template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_wgrad_pipe_kernel(const WgradParams p) {
    wgrad_pipe_body<WAVES_M, WAVES_N, TM, TN>(p, (int)blockIdx.x, (int)blockIdx.y);
}

// Several weight gradients of the SAME tile configuration in one launch (the deferred weight gradients of a step are
// independent of each other): workgroup b works on problem i with block0[i] <= b < block0[i+1], tile-major inside the
// problem like the single-problem grid.  Problems are ordered longest chunk first.
struct WgradGroupParams {
    int n;
    int block0[CTGAN_WGRAD_GROUP_MAX + 1];
    int tiles[CTGAN_WGRAD_GROUP_MAX];
    WgradParams p[CTGAN_WGRAD_GROUP_MAX];
};
template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_wgrad_pipe_group_kernel(const WgradGroupParams gp) {
    const int b = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int t = 1; t < CTGAN_WGRAD_GROUP_MAX; ++t)
        if (t < gp.n && b >= gp.block0[t]) i = t;
    const int r = b - gp.block0[i];
    const int by = r / gp.tiles[i];
    wgrad_pipe_body<WAVES_M, WAVES_N, TM, TN>(gp.p[i], r - by * gp.tiles[i], by);
}

// out[i] = sum_s part[s][i]   (fixed order => deterministic)
// elements [0, n_main) go to `out`, the trailing n - n_main (bias row) to `out2`.
// One thread per 4 consecutive elements (16-B loads), 4 independent accumulators over the slabs so that
// several loads are in flight; the summation order is fixed (slab k goes to accumulator k%4, then
// (a0+a1)+(a2+a3)) => deterministic.   n and n_main are multiples of 4 (Ng % 4 == 0 on this path).
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ out2,
                                     long long n, long long n_main, int splits) {
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = i4 * 4;
    if (i >= n) return;
    if (((n | n_main) & 3) == 0) {
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
        if (i < n_main) *reinterpret_cast<float4*>(out + i) = r;
        else *reinterpret_cast<float4*>(out2 + (i - n_main)) = r;
    } else {
        for (long long e = i; e < i + 4 && e < n; ++e) {
            float s0 = 0.f;
            for (int k = 0; k < splits; ++k) s0 += part[(long long)k * n + e];
            if (e < n_main) out[e] = s0; else out2[e - n_main] = s0;
        }
    }
}

// The same sums for SMALL outputs with many slabs (the 1x1 weight gradient of the im2col'd first critic conv: 12 K outputs x 85 slabs - one
// thread per float4 is 13 workgroups, each walking the slabs as a chain of dependent load batches: 15 us).  Four lanes per float4: lane j
// owns accumulator a_j of the kernel above - slabs j, j + 4, ... in ascending order, four loads in flight - and lane 0 combines
// (a0 + a1) + (a2 + a3): the same bits, a quarter of the chain.  n and n_main multiples of 4; blockDim 256 = 64 float4 x 4 lanes.
__global__ __launch_bounds__(256) void splitk_reduce_lanes_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ out2,
                                                                  long long n, long long n_main, int splits) {
    __shared__ float4 acc[4][64];
    const int g = threadIdx.x & 63, j = threadIdx.x >> 6;
    const long long i = ((long long)blockIdx.x * 64 + g) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const int full = splits & ~3;                        // the kernel above feeds a_j from the slabs of its unrolled loop only ...
        int k = j;
        for (; k + 12 < full; k += 16) {
            const float4 v0 = *reinterpret_cast<const float4*>(part + (long long)(k + 0) * n + i);
            const float4 v1 = *reinterpret_cast<const float4*>(part + (long long)(k + 4) * n + i);
            const float4 v2 = *reinterpret_cast<const float4*>(part + (long long)(k + 8) * n + i);
            const float4 v3 = *reinterpret_cast<const float4*>(part + (long long)(k + 12) * n + i);
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
            a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
            a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
        }
        for (; k < full; k += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(part + (long long)k * n + i);
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
        }
        if (j == 0)                                          // ... and the remaining splits % 4 slabs all go to a0
            for (k = full; k < splits; ++k) {
                const float4 v0 = *reinterpret_cast<const float4*>(part + (long long)k * n + i);
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            }
    }
    acc[j][g] = a;
    __syncthreads();
    if (j != 0 || i >= n) return;
    const float4 a0 = acc[0][g], a1 = acc[1][g], a2 = acc[2][g], a3 = acc[3][g];
    float4 r;
    r.x = (a0.x + a1.x) + (a2.x + a3.x); r.y = (a0.y + a1.y) + (a2.y + a3.y);
    r.z = (a0.z + a1.z) + (a2.z + a3.z); r.w = (a0.w + a1.w) + (a2.w + a3.w);
    if (i < n_main) *reinterpret_cast<float4*>(out + i) = r;
    else *reinterpret_cast<float4*>(out2 + (i - n_main)) = r;
}
// launches the reduction: the lane form where the float4-per-thread form would leave the chip empty in front of a long slab chain
static int g_reduce_lanes = 1;      // tests: ctgan_debug_reduce_lanes(0) = the float4-per-thread form everywhere
static void launch_splitk_reduce(const float* part, float* out, float* out2, long long n, long long n_main, int splits, hipStream_t st) {
    if (g_reduce_lanes && ((n | n_main) & 3) == 0 && splits >= 16 && n / 4 < 256 * 256)
        hipLaunchKernelGGL(splitk_reduce_lanes_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, st, part, out, out2, n, n_main, splits);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, part, out, out2, n, n_main, splits);
}

// wT[r',s',k,c] = w[R-1-r', S-1-s', c, k]   (dgrad filter: rotate 180 degrees, swap I/O)
__global__ void repack_dgrad_filter_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int S, int C, int K) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)R * S * C * K;
    if (i >= n) return;
    const int c = i % C; long long t = i / C;
    const int k = t % K; t /= K;
    const int s = t % S; const int r = t / S;
    wt[i] = w[(((long long)(R - 1 - r) * S + (S - 1 - s)) * C + c) * K + k];
}

// Phase-major filter of a stride-2 data gradient (see FwdParams::phases):
//   wt[ph=(a,b)][t][v][k][c] = w[u][x][c][k],  u = u0(a) + 2*(Tr-1-t),  x = x0(b) + 2*(Ts-1-v)   (0 where u >= R or x >= S)
// u0(a) = (a + pad_t) & 1: the filter rows that reach dx rows of parity a.
__global__ void repack_dgrad_phase_filter_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int S, int C, int K,
                                                 int pad_t, int pad_l) {
    const int Tr = (R + 1) / 2, Ts = (S + 1) / 2;
    const long long per = (long long)Tr * Ts * K * C, n = 4 * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ph = (int)(i / per);
        long long r = i - ph * per;
        const int c = (int)(r % C); r /= C;
        const int k = (int)(r % K); r /= K;
        const int v = (int)(r % Ts), t = (int)(r / Ts);
        const int a = ph >> 1, b = ph & 1;
        const int u = ((a + pad_t) & 1) + 2 * (Tr - 1 - t), x = ((b + pad_l) & 1) + 2 * (Ts - 1 - v);
        wt[i] = (u < R && x < S) ? w[(((long long)u * S + x) * C + c) * K + k] : 0.f;
    }
}

// shape-only rule shared by the repack, the workspace query and the launcher
inline bool dgrad_phase_mode(const ctgan_conv_desc* d) {
    return d->stride == 2 && !(d->H & 1) && !(d->W & 1) && (d->C % 4 == 0) && (d->K % 32 == 0) &&
           d->P * 2 == d->H && d->Q * 2 == d->W;
}
inline size_t dgrad_filter_elems(const ctgan_conv_desc* d) {
    if (dgrad_phase_mode(d)) return (size_t)4 * ((d->R + 1) / 2) * ((d->S + 1) / 2) * d->K * d->C;
    return (size_t)d->R * d->S * d->C * d->K;
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
thread_local char g_last_kernel[128] = "";
thread_local char g_last_symbol[160] = "";
bool g_force_generic = false;   // tests: route vectorisable shapes through the table-driven kernel too

template <bool AVEC, bool BVEC, int WM, int WN, int TM, int TN>
int launch_fwd(const FwdParams& p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    FwdParams q = p;
    q.ph_tiles_m = (p.M + BM - 1) / BM;
    const int tiles = q.ph_tiles_m * (p.phases > 1 ? p.phases : 1) * ((p.Ng + BN - 1) / BN);
    snprintf(g_last_kernel, sizeof g_last_kernel, "igemm_fwd<%s,%s,%dx%d%s>", AVEC ? "avec" : "agen",
             BVEC ? "bvec" : "bgen", BM, BN, p.phases > 1 ? ",ph4" : "");
    ctgan_set_last_symbol("igemm_fwd_kernel<%s, %s, %d, %d, %d, %d>", AVEC ? "true" : "false", BVEC ? "true" : "false", WM, WN, TM, TN);
    hipLaunchKernelGGL((igemm_fwd_kernel<AVEC, BVEC, WM, WN, TM, TN>), dim3(tiles), dim3(64 * WM * WN), 0, st, q);
    return ctgan_check_launch("igemm_fwd");
}

template <bool AVEC, bool BVEC>
int dispatch_fwd_tile(const FwdParams& p, hipStream_t st) {
    // pick the largest tile that still yields >= ~2 workgroups per CU; N tile by Ng
    const long long M = p.M;
    if (p.Ng > 64) {
        if (M >= 128LL * 512) return launch_fwd<AVEC, BVEC, 2, 2, 2, 2>(p, st);   // 128x128
        if (M >= 64LL * 384) return launch_fwd<AVEC, BVEC, 1, 4, 2, 1>(p, st);    // 64x128
        return launch_fwd<AVEC, BVEC, 1, 4, 1, 1>(p, st);                         // 32x128
    }
    if (p.Ng > 32) return launch_fwd<AVEC, BVEC, 2, 2, 1, 1>(p, st);              // 64x64
    return launch_fwd<AVEC, BVEC, 4, 1, 1, 1>(p, st);                             // 128x32
}

template <int WM, int WN, int WK, int TM, int TN, int RD, bool RELU_IN, int KSUB>
int launch_fwd_pipe_impl(const FwdParams& p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BKE = 32 * WK * KSUB;
    constexpr size_t smem_bytes = 2 * (size_t)(BM * (BKE + 4) + BKE * BN) * sizeof(float);
    static bool attr_set = false;     // one-time opt-in to > 64 KB of dynamic LDS (idempotent; benign race)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_pipe_kernel<WM, WN, WK, TM, TN, RD, RELU_IN, KSUB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
        if (e != hipSuccess) return ctgan_fail(CTGAN_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    FwdParams q = p;
    q.ph_tiles_m = (p.M + BM - 1) / BM;
    const int tiles = q.ph_tiles_m * (p.phases > 1 ? p.phases : 1) * ((p.Ng + BN - 1) / BN);
    snprintf(g_last_kernel, sizeof g_last_kernel, "igemm_fwd_pipe<%dx%d,k%d%s%s%s>", BM, BN, WK, KSUB > 1 ? ",bk64" : "", RELU_IN ? ",relu" : "",
             p.phases > 1 ? ",ph4" : "");
    ctgan_set_last_symbol("igemm_fwd_pipe_kernel<%d, %d, %d, %d, %d, %d, %s, %d>", WM, WN, WK, TM, TN, RD, RELU_IN ? "true" : "false", KSUB);
    hipLaunchKernelGGL((igemm_fwd_pipe_kernel<WM, WN, WK, TM, TN, RD, RELU_IN, KSUB>), dim3(tiles), dim3(64 * WM * WN * WK), smem_bytes, st, q);
    return ctgan_check_launch("igemm_fwd_pipe");
}

template <int WM, int WN, int WK, int TM, int TN, int RD, int KSUB = 1>
int launch_fwd_pipe(const FwdParams& p, hipStream_t st) {
    return p.relu_in ? launch_fwd_pipe_impl<WM, WN, WK, TM, TN, RD, true, KSUB>(p, st)
                     : launch_fwd_pipe_impl<WM, WN, WK, TM, TN, RD, false, KSUB>(p, st);
}

int dispatch_fwd_pipe(const FwdParams& p, hipStream_t st) {
    // Work per launch in units of 32x32 output tiles; 1024 SIMDs want >= 1024 waves of work.
    const long long M = p.M;
    int cfg = 0;
    {
        // measured on MI355X (tools/cfg_sweep.py, 128->128 3x3): the best tile shrinks with the number of
        // output rows so that >= ~1024 waves exist; the register ring depth RD bought nothing (kept at 1)
        const long long rows = M * (p.phases > 1 ? p.phases : 1) * ((p.Ng + 127) / 128);
        if (rows >= 65536) {
            // wave quantisation: 128x128 tiles run 2 per CU (512 slots), 64x128 tiles 3 per CU (768 slots); a last round
            // that is mostly empty costs more than the smaller tile's lower MFMA : LDS ratio (measured: 640 tiles 252 us
            // as 128x128, 198 us as 64x128)
            const long long t1 = (rows + 127) / 128, t2 = (rows + 63) / 64;
            const double e1 = (double)t1 / (double)(((t1 + 511) / 512) * 512), e2 = 0.95 * (double)t2 / (double)(((t2 + 767) / 768) * 768);
            cfg = e1 >= e2 ? 1 : 2;
            // the 4-phase data gradient has only 16 K slices per tile: three resident 64x128 workgroups per CU cover the
            // per-tile prologue / epilogue better than two 128x128 ones once there are several rounds (tools/ph4_sweep.py)
            if (p.phases > 1 && rows >= 131072) cfg = 2;
        }
        else if (rows > 24576) cfg = 2;
        else if (rows > 12288) cfg = (p.g.C % 64 == 0) ? 8 : 3;
        else if (rows > 8192) cfg = 4;
        else if (rows > 4096) cfg = (p.g.C % 128 == 0) ? 11 : 4;      // 64x64 x 4 K groups (16 waves): less L2 traffic per MFMA
        else if (rows > 2048) cfg = (p.g.C % 128 == 0) ? 10 : 4;      // 32x64 x 4 K groups (8 waves, one workgroup per CU)
        else cfg = 5;
    }
    if (((cfg == 4 || cfg == 6 || cfg == 7 || cfg == 8 || cfg == 9) && p.g.C % 64 != 0) || ((cfg == 5 || cfg == 10 || cfg == 11) && p.g.C % 128 != 0))
        cfg = (cfg == 6 || cfg == 7) ? 1 : 3;
    switch (cfg) {
        case 1: return launch_fwd_pipe<2, 2, 1, 2, 2, 1>(p, st);   // 128x128, 2 blocks/CU
        case 2: return launch_fwd_pipe<1, 4, 1, 2, 1, 1>(p, st);   // 64x128
        case 3: return launch_fwd_pipe<1, 4, 1, 1, 1, 1>(p, st);   // 32x128
        case 4: return launch_fwd_pipe<1, 2, 2, 1, 1, 1>(p, st);   // 32x64, K split over 2 wave groups
        case 6: return launch_fwd_pipe<2, 2, 1, 2, 2, 1, 2>(p, st);  // 128x128, 64-deep stages (1 block/CU)
        case 7: return launch_fwd_pipe<1, 4, 1, 2, 1, 1, 2>(p, st);  // 64x128, 64-deep stages
        case 8: return launch_fwd_pipe<2, 2, 2, 1, 1, 1>(p, st);   // 64x64, 8 waves: 2x2 tiles x 2 K groups
        case 9: return launch_fwd_pipe<2, 1, 2, 1, 1, 1>(p, st);   // 64x32, 2 K groups
        case 10: return launch_fwd_pipe<1, 2, 4, 1, 1, 1>(p, st);  // 32x64, 4 K groups (8 waves, 1 workgroup / CU)
        case 11: return launch_fwd_pipe<2, 2, 4, 1, 1, 1>(p, st);  // 64x64, 4 K groups (16 waves)
        default: return launch_fwd_pipe<1, 1, 4, 1, 1, 1>(p, st);  // 32x32, K split over 4 wave groups
    }
}

int run_fwd(const FwdParams& p0, hipStream_t st) {
    FwdParams p = p0;
    const Geom& g = p.g;
    p.d_lin = (p.phases <= 1) && (p.ds_p == (long long)g.Q * p.ds_q) && (p.ds_n == (long long)g.P * g.Q * p.ds_q);
    p.dbg = 0;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    p.d_vec = !(p.dbg & 32) && p.ds_k == 1 && (p.Ng % 4 == 0) && (p.ds_n % 4 == 0) && (p.ds_p % 4 == 0) && (p.ds_q % 4 == 0) &&
              (p.phases <= 1 || ((p.ph_d_h % 4 == 0) && (p.ph_d_w % 4 == 0))) && al16(p.D) && al16(p.mask) && al16(p.resid) &&
              al16(p.bias);
    const bool avec = (g.C % 32 == 0) && g.s_c == 1 && (g.s_n % 4 == 0) && (g.s_h % 4 == 0) && (g.s_w % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
    const bool bvec = p.bs_k == 1 && (p.Ng % 4 == 0) && (p.b_off % 4 == 0) && (p.bs_r % 4 == 0) && (p.bs_s % 4 == 0) &&
                      (p.bs_c % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);
    // byte extents for the buffer descriptors of the pipelined kernel (32-bit range => < 4 GiB)
    const long long nimg = g.P > 0 ? (p.M + (long long)g.P * g.Q - 1) / ((long long)g.P * g.Q) : 0;
    const long long a_elems = (nimg - 1) * g.s_n + (long long)(g.H - 1) * g.s_h + (long long)(g.W - 1) * g.s_w + g.C;
    const long long b_elems = (long long)(p.phases > 1 ? p.phases : 1) * g.R * g.S * g.C * p.Ng;
    const bool small = a_elems > 0 && a_elems * 4 < (1LL << 32) && b_elems * 4 < (1LL << 32) && p.b_off >= 0 && p.bs_r >= 0 && p.bs_s >= 0;
    if (avec && bvec && small && p.Ng > 64 && !g_force_generic && ((!p.drop && !p.resid_up) || p.d_vec)) {
        p.a_bytes = (unsigned)(a_elems * 4);
        p.b_bytes = (unsigned)(b_elems * 4);
        return dispatch_fwd_pipe(p, st);
    }
    if (p.drop || p.resid_up)
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d: epilogue dropout / upsampled residual need the pipelined kernel with a vector epilogue");
    if (avec && bvec) return dispatch_fwd_tile<true, true>(p, st);
    if (avec) return dispatch_fwd_tile<true, false>(p, st);
    if (bvec) return dispatch_fwd_tile<false, true>(p, st);
    return dispatch_fwd_tile<false, false>(p, st);
}

// ---- weight-gradient planning (shared by the launcher and the workspace query) ------------
enum WTile { W128x128, W64x128, W32x128, W64x64, W128x32 };
// the split-K reductions of several weight gradients in one launch: blockIdx.y = job
// add / add2 (optional): finished addends of the same shapes as out / out2 - the weight gradient a split-mode launch produced earlier
// for the same filter; summed AFTER the slabs (r = slabs + add).  They may alias out / out2 (accumulate in place: each thread reads
// and writes the same four elements).
struct ReduceJob { const float* part; float* out; float* out2; const float* add; const float* add2; long long n, n_main; int splits, pad; };
struct ReduceJobs { int n; int pad; ReduceJob j[CTGAN_REDUCE_BATCH]; };
__global__ void splitk_reduce_batch_kernel(const ReduceJobs jobs) {
    const ReduceJob& jb = jobs.j[blockIdx.y];
    const long long n = jb.n, n_main = jb.n_main;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;       // n, n_main multiples of 4 on this path
    if (i >= n) return;
    const float* part = jb.part;
    const int splits = jb.splits;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {                                                   // same order as splitk_reduce_kernel
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

struct WPlan { WTile tile; int bm, bn, tiles, splits, chunk; };

WPlan wgrad_plan(int C, int Mtot, int Ng, int Kg) {
    WPlan w;
    if (C % 32 == 0) {           // vector-capable: the M tile must lie inside one filter tap
        if (Ng > 64) {
            // few pixels => prefer smaller tiles: more output tiles, fewer split-K slabs to reduce
            // measured (tools/wgrad_sweep.py): ~2 workgroups per CU with the largest tile that still leaves
            // >= ~16 K slices per split
            if (C % 128 == 0 && Kg >= 32768) w.tile = W128x128;
            else if (C % 64 == 0 && Kg >= 8192) w.tile = W64x128;
            else if (C % 64 == 0) w.tile = W64x64;
            else w.tile = W32x128;
        } else if (Ng > 32) w.tile = (C % 64 == 0) ? W64x64 : W32x128;
        else w.tile = (C % 128 == 0) ? W128x32 : W32x128;
    } else {                     // generic gather: any M tile
        if (Ng > 64) w.tile = Mtot > 64 ? W128x128 : Mtot > 32 ? W64x128 : W32x128;
        else if (Ng > 32) w.tile = Mtot > 32 ? W64x64 : W32x128;
        else w.tile = Mtot > 32 ? W128x32 : W32x128;
    }
    static const int dims[5][2] = {{128, 128}, {64, 128}, {32, 128}, {64, 64}, {128, 32}};
    w.bm = dims[w.tile][0]; w.bn = dims[w.tile][1];
    w.tiles = ((Mtot + w.bm - 1) / w.bm) * ((Ng + w.bn - 1) / w.bn);
    ctgan_wgrad_split(w.tiles, Kg, &w.splits, &w.chunk);
    return w;
}

size_t wgrad_slab_bytes(const WPlan& w, int Mtot, int Ng) {
    return w.splits > 1 ? (size_t)w.splits * (Mtot + 1) * Ng * sizeof(float) : 0;
}

template <bool AVEC, bool BVEC, int WM, int WN, int TM, int TN>
int launch_wgrad(WgradParams p, const WPlan& w, float* dw, void* ws, hipStream_t st) {
    p.chunk = w.chunk;
    p.with_bias = 0;
    p.OUT = w.splits > 1 ? static_cast<float*>(ws) : dw;
    snprintf(g_last_kernel, sizeof g_last_kernel, "igemm_wgrad<%s,%s,%dx%d,split%d>", AVEC ? "avec" : "agen",
             BVEC ? "bvec" : "bgen", w.bm, w.bn, w.splits);
    ctgan_set_last_symbol("igemm_wgrad_kernel<%s, %s, %d, %d, %d, %d>", AVEC ? "true" : "false", BVEC ? "true" : "false", WM, WN, TM, TN);
    hipLaunchKernelGGL((igemm_wgrad_kernel<AVEC, BVEC, WM, WN, TM, TN>), dim3(w.tiles, w.splits), dim3(64 * WM * WN), 0, st, p);
    int rc = ctgan_check_launch("igemm_wgrad");
    if (rc) return rc;
    if (w.splits > 1) {
        const long long n = (long long)p.Mtot * p.Ng;
        launch_splitk_reduce(p.OUT, dw, dw, n, n, w.splits, st);
        rc = ctgan_check_launch("splitk_reduce");
    }
    return rc;
}

template <bool AVEC, bool BVEC>
int dispatch_wgrad_tile(const WgradParams& p, const WPlan& w, float* dw, void* ws, hipStream_t st) {
    switch (w.tile) {
        case W128x128: return launch_wgrad<AVEC, BVEC, 2, 2, 2, 2>(p, w, dw, ws, st);
        case W64x128: return launch_wgrad<AVEC, BVEC, 1, 4, 2, 1>(p, w, dw, ws, st);
        case W32x128: return launch_wgrad<AVEC, BVEC, 1, 4, 1, 1>(p, w, dw, ws, st);
        case W64x64: return launch_wgrad<AVEC, BVEC, 2, 2, 1, 1>(p, w, dw, ws, st);
        default: return launch_wgrad<AVEC, BVEC, 4, 1, 1, 1>(p, w, dw, ws, st);
    }
}

template <int WM, int WN, int TM, int TN>
int launch_wgrad_pipe(WgradParams p, const WPlan& w, float* dw, float* db, void* ws, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem_bytes = 2 * (size_t)(BK * BM + BK * BN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_pipe_kernel<WM, WN, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
        if (e != hipSuccess) return ctgan_fail(CTGAN_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    p.chunk = w.chunk;
    p.with_bias = db ? 1 : 0;
    // with a bias row the slab layout is [Mtot+1][Ng]; a single split still goes through the slab so that
    // dw stays exactly [Mtot][Ng]
    const bool direct = w.splits == 1 && !db;
    p.OUT = direct ? dw : static_cast<float*>(ws);
    snprintf(g_last_kernel, sizeof g_last_kernel, "igemm_wgrad_pipe<%dx%d,split%d%s>", BM, BN, w.splits, db ? ",bias" : "");
    ctgan_set_last_symbol("igemm_wgrad_pipe_kernel<%d, %d, %d, %d>", WM, WN, TM, TN);
    hipLaunchKernelGGL((igemm_wgrad_pipe_kernel<WM, WN, TM, TN>), dim3(w.tiles, w.splits), dim3(64 * WM * WN), smem_bytes, st, p);
    int rc = ctgan_check_launch("igemm_wgrad_pipe");
    if (rc || direct) return rc;
    const long long n_main = (long long)p.Mtot * p.Ng, n = n_main + (db ? p.Ng : 0);
    launch_splitk_reduce(p.OUT, dw, db, n, n_main, w.splits, st);
    return ctgan_check_launch("splitk_reduce");
}

// returns 1 if the bias gradient was produced by the fused path, 0 if not, <0 on error
int run_wgrad(WgradParams p, float* dw, float* db, void* ws, size_t wsb, hipStream_t st) {
    const Geom& g = p.g;
    const WPlan w = wgrad_plan(g.C, p.Mtot, p.Ng, p.Kg);
    const size_t need = (w.splits > 1 || db) ? (size_t)w.splits * (p.Mtot + 1) * p.Ng * sizeof(float) : 0;
    if (need > wsb) return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad: workspace too small (%zu < %zu)", wsb, need);
    const bool avec = (g.C % 32 == 0) && g.s_c == 1 && (g.s_n % 4 == 0) && (g.s_h % 4 == 0) && (g.s_w % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.X) & 15) == 0);
    const bool bvec = p.dy_k == 1 && (p.Ng % 4 == 0) && (p.dy_n % 4 == 0) && (p.dy_p % 4 == 0) && (p.dy_q % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.DY) & 15) == 0);
    if (avec && bvec && !g_force_generic && (w.tile == W128x128 || w.tile == W64x128 || w.tile == W64x64 || w.tile == W32x128)) {
        const long long nimg = (p.Kg + (long long)g.P * g.Q - 1) / ((long long)g.P * g.Q);
        const long long x_elems = (nimg - 1) * g.s_n + (long long)(g.H - 1) * g.s_h + (long long)(g.W - 1) * g.s_w + g.C;
        const long long y_elems = (nimg - 1) * p.dy_n + (long long)(g.P - 1) * p.dy_p + (long long)(g.Q - 1) * p.dy_q + p.Ng;
        if (x_elems * 4 < (1LL << 32) && y_elems * 4 < (1LL << 32)) {
            p.x_bytes = (unsigned)(x_elems * 4); p.dy_bytes = (unsigned)(y_elems * 4);
            int rc;
            if (w.tile == W128x128) rc = launch_wgrad_pipe<2, 2, 2, 2>(p, w, dw, db, ws, st);
            else if (w.tile == W64x128) rc = launch_wgrad_pipe<1, 4, 2, 1>(p, w, dw, db, ws, st);
            else if (w.tile == W32x128) rc = launch_wgrad_pipe<1, 4, 1, 1>(p, w, dw, db, ws, st);
            else rc = launch_wgrad_pipe<2, 2, 1, 1>(p, w, dw, db, ws, st);
            return rc ? rc : (db ? 1 : 0);
        }
    }
    int rc;
    if (avec && bvec) rc = dispatch_wgrad_tile<true, true>(p, w, dw, ws, st);
    else if (avec) rc = dispatch_wgrad_tile<true, false>(p, w, dw, ws, st);
    else if (bvec) rc = dispatch_wgrad_tile<false, true>(p, w, dw, ws, st);
    else rc = dispatch_wgrad_tile<false, false>(p, w, dw, ws, st);
    return rc;
}

int check_desc(const ctgan_conv_desc* d, const char* who) {
    if (!d) return ctgan_fail(CTGAN_E_BADARG, "%s: null descriptor", who);
    if (d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->K <= 0 || d->R <= 0 || d->S <= 0 || d->P <= 0 || d->Q <= 0)
        return ctgan_fail(CTGAN_E_BADARG, "%s: non-positive dimension", who);
    if (d->stride != 1 && d->stride != 2) return ctgan_fail(CTGAN_E_UNSUPPORTED, "%s: stride %d (1 or 2)", who, d->stride);
    if (d->x_up && (d->H % 2 || d->W % 2)) return ctgan_fail(CTGAN_E_BADARG, "%s: x_up needs even H,W", who);
    if ((long long)d->N * d->P * d->Q >= (1LL << 31) || (long long)d->R * d->S * d->C >= (1LL << 31))
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "%s: GEMM extent exceeds int32", who);
    return 0;
}

Geom geom_from_x(const ctgan_conv_desc* d) {   // gather from x (FWD / WGRAD)
    Geom g;
    g.H = d->x_up ? d->H / 2 : d->H; g.W = d->x_up ? d->W / 2 : d->W;
    g.P = d->P; g.Q = d->Q; g.R = d->R; g.S = d->S; g.C = d->C;
    g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l;
    g.shift = d->x_up ? 1 : 0; g.mask = 0;
    g.s_n = d->xs[0]; g.s_c = d->xs[1]; g.s_h = d->xs[2]; g.s_w = d->xs[3];
    return g;
}

}  // namespace

void ctgan_wgrad_split(int tiles, int Kg, int* splits, int* chunk) {
    // choose splits so that tiles*splits ~ k*256 workgroups (k small) with >= 4 slices per split
    const int max_splits = (Kg + 4 * BK - 1) / (4 * BK);
    int best = 1;
    // one or two output tiles (few-channel / skinny weight gradients): a streaming reduction over the pixel axis,
    // bound by load latency.  Measured (tools/skinny_w.py): ~384 pixels per workgroup, at most one workgroup per CU;
    // more splits only add slab traffic and pipeline fill/drain.
    if (tiles <= 2) {
        best = Kg / 384;
        if (best > 256 / tiles) best = 256 / tiles;
        if (best > max_splits) best = max_splits;
    } else {
        for (int k = 2; k <= 4; ++k) {
            int s = (256 * k) / tiles;
            if (s < 1) s = 1;
            if (s > max_splits) s = max_splits;
            best = s;
            if ((double)tiles * s >= 0.9 * 256 * k || s == max_splits) break;
        }
    }
    if (best < 1) best = 1;
    int ch = (Kg + best - 1) / best;
    ch = ((ch + BK - 1) / BK) * BK;
    *splits = (Kg + ch - 1) / ch;
    *chunk = ch;
}

void ctgan_set_last_kernel(const char* name) { snprintf(g_last_kernel, sizeof g_last_kernel, "%s", name); g_last_symbol[0] = 0; }
void ctgan_set_last_symbol(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_symbol, sizeof g_last_symbol, fmt, ap);
    va_end(ap);
}

extern "C" {

const char* ctgan_last_kernel(void) { return g_last_kernel; }
const char* ctgan_last_symbol(void) { return g_last_symbol[0] ? g_last_symbol : g_last_kernel; }
void ctgan_debug_force_generic(int on) { g_force_generic = on != 0; }
void ctgan_debug_reduce_lanes(int on) { g_reduce_lanes = on ? 1 : 0; }

size_t ctgan_conv2d_workspace_bytes(const ctgan_conv_desc* d, int op) {
    if (!d) return 0;
    if (op == CTGAN_CONV_DGRAD) return dgrad_filter_elems(d) * sizeof(float);
    if (op == CTGAN_CONV_WGRAD) {
        const int mt = d->R * d->S * d->C, Kg = d->N * d->P * d->Q;
        const WPlan w = wgrad_plan(d->C, mt, d->K, Kg);
        // slabs (always sized for the bias row) + room for the stand-alone bias column sums
        const size_t gemm = (size_t)w.splits * (mt + 1) * d->K * sizeof(float) + ctgan_colsum_workspace_bytes(Kg, d->K);
        const size_t few = ctgan_fewch_wgrad_workspace(d);
        return gemm > few ? gemm : few;
    }
    return 0;
}

int ctgan_conv2d_fwd(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, const float* resid,
                     float* y, int flags, ctgan_stream_t stream) {
    return ctgan_conv2d_fwd_ex(d, x, w, bias, resid, y, flags, nullptr, stream);
}

static bool ext_wants_drop(const ctgan_epilogue_ext* ext) {
    if (!ext) return false;
    if (ext->n_ranges > 0) {
        for (int i = 0; i < ext->n_ranges && i < CTGAN_DROP_RANGES; ++i)
            if (ext->range_keep[i] > 0.f && ext->range_keep[i] < 1.f) return true;
        return false;
    }
    return ext->drop_keep > 0.f && ext->drop_keep < 1.f;
}
// rows_per_sample = output pixels per sample (P*Q), sample_elems = elements of one sample of the (dense) output
static void set_drop(FwdParams& p, const ctgan_epilogue_ext* ext, int rows_per_sample = 0, long long sample_elems = 0) {
    p.drop = 0; p.drop_keep = 1.f; p.drop_seed = 0; p.drop_sid = 0; p.drop_ctr = nullptr; p.drop_nr = 0;
    for (int i = 0; i < CTGAN_DROP_RANGES; ++i) { p.drop_mend[i] = 0x7fffffff; p.drop_rkeep[i] = 1.f; p.drop_rsid[i] = 0; p.drop_roff[i] = 0; }
    if (!ext_wants_drop(ext)) return;
    p.drop = 1; p.drop_seed = ext->drop_seed;
    p.drop_ctr = reinterpret_cast<const unsigned long long*>(ext->drop_ctr);
    if (ext->n_ranges > 0) {
        p.drop_nr = ext->n_ranges < CTGAN_DROP_RANGES ? ext->n_ranges : CTGAN_DROP_RANGES;
        long long start = 0;
        for (int i = 0; i < p.drop_nr; ++i) {
            p.drop_mend[i] = (int)(ext->range_end[i] * rows_per_sample);
            p.drop_rkeep[i] = (ext->range_keep[i] > 0.f && ext->range_keep[i] < 1.f) ? ext->range_keep[i] : 1.f;
            p.drop_rsid[i] = (unsigned)ext->range_stream_id[i];
            p.drop_roff[i] = start * sample_elems;
            start = ext->range_end[i];
        }
    } else {
        p.drop_keep = ext->drop_keep; p.drop_sid = (unsigned)ext->drop_stream_id;
    }
}

int ctgan_conv2d_fwd_ex(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, const float* resid,
                        float* y, int flags, const ctgan_epilogue_ext* ext, ctgan_stream_t stream) {
    int rc = check_desc(d, "conv2d_fwd");
    const bool want_drop = ext_wants_drop(ext);
    if (rc) return rc;
    if (!x || !w || !y) return ctgan_fail(CTGAN_E_BADARG, "conv2d_fwd: null pointer");
    if (ext && ext->act) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_fwd_ex: the fused LeakyReLU + dropout epilogue exists in the 16-bit slice kernels only");
    const float* out_mask = ext ? ext->out_mask : nullptr;
    if (ext && (ext->in_bn_mean || ext->out_tanh)) {
        // batch norm of the input on load (+ tanh of the result): the one-pixel-per-lane many -> few kernel only
        if (!ext->in_bn_mean || !ext->in_bn_rstd || !ext->in_bn_scale || !ext->in_bn_offset || ext->in_bn_labels || resid || out_mask || want_drop ||
            (flags & (CTGAN_EPI_RELU | CTGAN_RESID_UP)))
            return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_fwd_ex: input batch norm / tanh epilogue with other epilogue operands");
        rc = ctgan_fewch_fwd_bn(d, x, w, bias, y, (flags & CTGAN_IN_RELU) ? 1 : 0, ext->in_bn_mean, ext->in_bn_rstd, ext->in_bn_scale, ext->in_bn_offset,
                                ext->in_bn_groups, ext->out_tanh, static_cast<hipStream_t>(stream));
        if (rc == 0) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_fwd_ex: input batch norm on load exists in the many -> few pixel kernel only (3x3, <= 4 output channels, 32-pixel rows)");
        return rc < 0 ? rc : CTGAN_OK;
    }
    if (ctgan_is_small_linear(d) && !resid && !out_mask && !(flags & CTGAN_IN_RELU) && !g_force_generic && !want_drop) {
        ctgan_set_last_kernel("linear_small_fwd");
        return ctgan_small_linear_fwd(d, x, w, bias, y, (flags & CTGAN_EPI_RELU) ? 1 : 0, static_cast<hipStream_t>(stream));
    }
    if (!g_force_generic && !want_drop && !(flags & CTGAN_RESID_UP)) {
        rc = ctgan_fewch_fwd(d, x, w, bias, out_mask, resid, y, (flags & CTGAN_EPI_RELU) ? 1 : 0, (flags & CTGAN_IN_RELU) ? 1 : 0,
                             static_cast<hipStream_t>(stream));
        if (rc) return rc < 0 ? rc : CTGAN_OK;
    }
    FwdParams p;
    p.g = geom_from_x(d);
    p.A = x; p.B = w; p.bias = bias; p.resid = resid; p.D = y;
    p.M = d->N * d->P * d->Q; p.Ng = d->K; p.Kg = d->R * d->S * d->C;
    p.b_off = 0; p.bs_r = (long long)d->S * d->C * d->K; p.bs_s = (long long)d->C * d->K; p.bs_c = d->K; p.bs_k = 1;
    p.ds_n = d->ys[0]; p.ds_k = d->ys[1]; p.ds_p = d->ys[2]; p.ds_q = d->ys[3];
    p.relu = (flags & CTGAN_EPI_RELU) ? 1 : 0;
    p.relu_in = (flags & CTGAN_IN_RELU) ? 1 : 0;
    p.mask = out_mask;
    p.phases = 1;
    p.resid_up = (resid && (flags & CTGAN_RESID_UP)) ? 1 : 0;
    if (p.resid_up && ((d->P | d->Q) & 1)) return ctgan_fail(CTGAN_E_BADARG, "conv2d_fwd: CTGAN_RESID_UP needs even P, Q");
    if (ext && ext->n_ranges > 0 && (ext->n_ranges > CTGAN_DROP_RANGES || d->ys[1] != 1 || d->ys[0] != (long long)d->P * d->Q * d->K))
        return ctgan_fail(CTGAN_E_BADARG, "conv2d_fwd: dropout row ranges need a dense channels-last result and <= %d ranges", CTGAN_DROP_RANGES);
    if (ext && ext->n_ranges > 0)
        for (int i = 0; i + 1 < ext->n_ranges; ++i)
            if (((long long)ext->range_end[i] * d->P * d->Q) % 128)      // a range boundary inside an M tile: caller drops per range
                return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_fwd: dropout row ranges must start at multiples of 128 output pixels");
    set_drop(p, ext, d->P * d->Q, (long long)d->P * d->Q * d->K);
    return run_fwd(p, static_cast<hipStream_t>(stream));
}

int ctgan_conv2d_repack_filter(const ctgan_conv_desc* d, const float* w, float* wt, ctgan_stream_t stream) {
    if (!d || !w || !wt) return ctgan_fail(CTGAN_E_BADARG, "conv2d_repack_filter: null pointer");
    const long long n = (long long)dgrad_filter_elems(d);
    if (dgrad_phase_mode(d)) {
        hipLaunchKernelGGL(repack_dgrad_phase_filter_kernel, dim3(ctgan_blocks(n, 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), w, wt, d->R, d->S, d->C, d->K, d->pad_t, d->pad_l);
        return ctgan_check_launch("repack_dgrad_phase_filter");
    }
    hipLaunchKernelGGL(repack_dgrad_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, wt, d->R, d->S, d->C, d->K);
    return ctgan_check_launch("repack_dgrad_filter");
}

int ctgan_conv2d_dgrad(const ctgan_conv_desc* d, const float* dy, const float* w, const float* bias, const float* mask,
                       const float* resid, float* dx, void* ws, size_t ws_bytes, int flags, ctgan_stream_t stream) {
    return ctgan_conv2d_dgrad_ex(d, dy, w, bias, mask, resid, dx, ws, ws_bytes, flags, nullptr, stream);
}

int ctgan_conv2d_dgrad_ex(const ctgan_conv_desc* d, const float* dy, const float* w, const float* bias, const float* mask,
                          const float* resid, float* dx, void* ws, size_t ws_bytes, int flags, const ctgan_epilogue_ext* ext,
                          ctgan_stream_t stream) {
    int rc = check_desc(d, "conv2d_dgrad");
    const bool want_drop = ext_wants_drop(ext);
    if (rc) return rc;
    if (!dy || !w || !dx) return ctgan_fail(CTGAN_E_BADARG, "conv2d_dgrad: null pointer");
    if (ext && ext->act) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_dgrad_ex: the fused LeakyReLU + dropout epilogue exists in the 16-bit slice kernels only");
    if (d->x_up) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_dgrad: x_up (pool the result instead)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ctgan_is_small_linear(d) && !mask && !resid && !g_force_generic && !want_drop) {
        ctgan_set_last_kernel("linear_small_dgrad");
        if (flags & CTGAN_DGRAD_W_REPACKED) return ctgan_fail(CTGAN_E_BADARG, "conv2d_dgrad: small linear takes the original filter");
        return ctgan_small_linear_dgrad(d, dy, w, bias, dx, st);
    }
    if (!mask && !resid && !(flags & CTGAN_DGRAD_W_REPACKED) && !g_force_generic && !want_drop) {
        rc = ctgan_fewch_dgrad(d, dy, w, bias, dx, st);
        if (rc) return rc < 0 ? rc : CTGAN_OK;
    }
    FwdParams p;
    Geom& g = p.g;
    g.H = d->P; g.W = d->Q;                 // physical source = dy
    g.C = d->K;                             // channels per tap = dy channels
    g.s_n = d->ys[0]; g.s_c = d->ys[1]; g.s_h = d->ys[2]; g.s_w = d->ys[3];
    p.A = dy; p.bias = bias; p.resid = resid; p.mask = mask; p.relu_in = 0; p.D = dx;
    p.Ng = d->C;
    p.ds_n = d->xs[0]; p.ds_k = d->xs[1]; p.ds_p = d->xs[2]; p.ds_q = d->xs[3];
    p.relu = 0;
    p.phases = 1;
    p.resid_up = 0;
    // Row ranges (round 5: the merged backward of a critic step carries the main rows and the gradient-penalty rows in ONE data gradient,
    // each range with the mask of its own forward dropout): stride-1 form on a dense channels-last dx, boundaries on 128-pixel tiles.
    if (ext && ext->n_ranges > 0) {
        if (ext->n_ranges > CTGAN_DROP_RANGES || d->stride != 1 || d->xs[1] != 1 || d->xs[3] != d->C || d->xs[2] != (long long)d->W * d->C ||
            d->xs[0] != (long long)d->H * d->W * d->C)
            return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_dgrad: dropout row ranges need stride 1, a dense channels-last dx and <= %d ranges", CTGAN_DROP_RANGES);
        for (int i = 0; i + 1 < ext->n_ranges; ++i)
            if (((long long)ext->range_end[i] * d->H * d->W) % 128)
                return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_dgrad: dropout row ranges must start at multiples of 128 pixels");
    }
    set_drop(p, ext, d->H * d->W, (long long)d->H * d->W * d->C);
    const size_t need = dgrad_filter_elems(d) * sizeof(float);
    const bool pre = (flags & CTGAN_DGRAD_W_REPACKED) != 0;
    const bool repack = !pre && ws && ws_bytes >= need && (d->C % 4 == 0) && (d->K % 32 == 0);
    if (dgrad_phase_mode(d) && (pre || repack)) {
        // stride 2: four stride-1 convs of dy, one per output parity, in one launch
        const int Tr = (d->R + 1) / 2, Ts = (d->S + 1) / 2;
        g.P = d->H / 2; g.Q = d->W / 2;     // rows enumerate the dx pixels of ONE phase
        g.R = Tr; g.S = Ts;
        g.stride = 1; g.pad_t = 0; g.pad_l = 0; g.shift = 0; g.mask = 0;
        p.M = d->N * g.P * g.Q; p.Kg = Tr * Ts * d->K;
        p.phases = 4;
        for (int a = 0; a < 2; ++a) {
            const int u0 = (a + d->pad_t) & 1, x0 = (a + d->pad_l) & 1;
            p.ph_pad_t[a] = Tr - 1 - (a + d->pad_t - u0) / 2;
            p.ph_pad_l[a] = Ts - 1 - (a + d->pad_l - x0) / 2;
        }
        p.ph_b_stride = (long long)Tr * Ts * d->K * d->C;
        p.ph_d_h = d->xs[2]; p.ph_d_w = d->xs[3];
        p.ds_p = 2 * d->xs[2]; p.ds_q = 2 * d->xs[3];
        if (repack) {
            hipLaunchKernelGGL(repack_dgrad_phase_filter_kernel, dim3(ctgan_blocks((long long)dgrad_filter_elems(d), 256)), dim3(256), 0, st,
                               w, static_cast<float*>(ws), d->R, d->S, d->C, d->K, d->pad_t, d->pad_l);
            rc = ctgan_check_launch("repack_dgrad_phase_filter");
            if (rc) return rc;
        }
        p.B = pre ? w : static_cast<const float*>(ws);
        p.b_off = 0; p.bs_r = (long long)Ts * d->K * d->C; p.bs_s = (long long)d->K * d->C; p.bs_c = d->C; p.bs_k = 1;
        return run_fwd(p, st);
    }
    if (pre && dgrad_phase_mode(d)) return ctgan_fail(CTGAN_E_BADARG, "conv2d_dgrad: phase-mode filter expected");
    g.P = d->H; g.Q = d->W;                 // rows enumerate dx pixels
    g.R = d->R; g.S = d->S;
    g.stride = 1; g.pad_t = d->R - 1 - d->pad_t; g.pad_l = d->S - 1 - d->pad_l;
    g.shift = d->stride == 2 ? 1 : 0; g.mask = d->stride == 2 ? 1 : 0;
    p.M = d->N * d->H * d->W; p.Kg = d->R * d->S * d->K;
    if (pre) {                                  // `w` already is wT[r',s',k,c] (ctgan_conv2d_repack_filter)
        p.B = w;
        p.b_off = 0; p.bs_r = (long long)d->S * d->K * d->C; p.bs_s = (long long)d->K * d->C; p.bs_c = d->C; p.bs_k = 1;
    } else if (repack) {
        const long long n = (long long)d->R * d->S * d->C * d->K;
        hipLaunchKernelGGL(repack_dgrad_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w,
                           static_cast<float*>(ws), d->R, d->S, d->C, d->K);
        rc = ctgan_check_launch("repack_dgrad_filter");
        if (rc) return rc;
        p.B = static_cast<const float*>(ws);
        p.b_off = 0; p.bs_r = (long long)d->S * d->K * d->C; p.bs_s = (long long)d->K * d->C; p.bs_c = d->C; p.bs_k = 1;
    } else {
        p.B = w;   // strided view of the original HWIO filter: rotate via negative tap strides
        p.b_off = ((long long)(d->R - 1) * d->S + (d->S - 1)) * d->C * d->K;
        p.bs_r = -(long long)d->S * d->C * d->K; p.bs_s = -(long long)d->C * d->K; p.bs_c = 1; p.bs_k = d->K;
    }
    return run_fwd(p, st);
}

int ctgan_conv2d_wgrad(const ctgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws,
                       size_t ws_bytes, int flags, ctgan_stream_t stream) {
    int rc = check_desc(d, "conv2d_wgrad");
    if (rc) return rc;
    if (!x || !dy || !dw) return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad: null pointer");
    if (ctgan_is_small_linear(d) && !(flags & CTGAN_IN_RELU) && !g_force_generic) {
        ctgan_set_last_kernel("linear_small_wgrad");
        return ctgan_small_linear_wgrad(d, x, dy, dw, db, static_cast<hipStream_t>(stream));
    }
    if (!g_force_generic) {
        rc = ctgan_fewch_wgrad(d, x, dy, dw, db, ws, ws_bytes, (flags & CTGAN_IN_RELU) ? 1 : 0, static_cast<hipStream_t>(stream));
        if (rc) return rc < 0 ? rc : CTGAN_OK;
    }
    WgradParams p;
    p.g = geom_from_x(d);
    p.X = x; p.DY = dy; p.OUT = dw;
    p.Mtot = d->R * d->S * d->C; p.Ng = d->K; p.Kg = d->N * d->P * d->Q;
    p.dy_n = d->ys[0]; p.dy_k = d->ys[1]; p.dy_p = d->ys[2]; p.dy_q = d->ys[3];
    p.chunk = 0; p.with_bias = 0; p.x_bytes = p.dy_bytes = 0; p.nseg = 0;
    p.relu_x = (flags & CTGAN_IN_RELU) ? 1 : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = run_wgrad(p, dw, db, ws, ws_bytes, st);
    if (rc < 0) return rc;
    if (db && rc == 0) {
        // bias gradient not fused: column sums of dy.  Needs a pixel-linear channels-last dy.
        // pixel-linear rows of K contiguous channels (strides of size-1 dims are irrelevant)
        const long long ld = d->Q > 1 ? p.dy_q : (d->P > 1 ? p.dy_p : p.dy_n);
        const bool lin = p.dy_k == 1 && (d->Q == 1 || p.dy_q == ld) && (d->P == 1 || p.dy_p == (long long)d->Q * ld) &&
                         (d->N == 1 || p.dy_n == (long long)d->P * d->Q * ld);
        if (!lin) return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_wgrad: bias gradient needs a channels-last dy");
        const WPlan w = wgrad_plan(d->C, p.Mtot, p.Ng, p.Kg);
        const size_t slab = (size_t)w.splits * (p.Mtot + 1) * p.Ng * sizeof(float);
        if (ws_bytes < slab + ctgan_colsum_workspace_bytes(p.Kg, p.Ng))
            return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad: workspace too small for the bias gradient");
        return ctgan_colsum(dy, p.Kg, p.Ng, ld, db, static_cast<char*>(ws) + slab, ws_bytes - slab, stream);
    }
    return CTGAN_OK;
}

}  // extern "C"

// ---- multi-segment weight gradient ---------------------------------------------------------------------------
namespace {
struct MultiPlan { WPlan w; int seg_splits[CTGAN_WGRAD_MAX_SEGS]; int splits; };
MultiPlan multi_plan(const ctgan_conv_desc* d, int nseg, const int32_t* Ns) {
    MultiPlan m;
    const int mt = d->R * d->S * d->C;
    long long kg = 0;
    for (int i = 0; i < nseg; ++i) kg += (long long)Ns[i] * d->P * d->Q;
    m.w = wgrad_plan(d->C, mt, d->K, (int)kg);
    // a split never straddles segments, so rounding each segment up can exceed the planned number of workgroups by
    // nseg-1 - one more than a full round of the chip costs a whole extra round: grow the chunk until it fits
    const int target = m.w.splits;
    for (;;) {
        m.splits = 0;
        for (int i = 0; i < nseg; ++i) {
            const long long k = (long long)Ns[i] * d->P * d->Q;
            m.seg_splits[i] = (int)((k + m.w.chunk - 1) / m.w.chunk);
            m.splits += m.seg_splits[i];
        }
        // (every segment needs at least one split: with a planned count below the number of segments - three small segments, one planned
        // split - `splits <= target` can never hold; this loop then never ended: the 128x128 ResNet's 8x8 shortcut at B = 4, round 3)
        if (m.splits <= target || m.splits <= nseg || nseg == 1) break;
        m.w.chunk += BK;
    }
    return m;
}
}  // namespace

extern "C" size_t ctgan_conv2d_wgrad_multi_workspace_bytes(const ctgan_conv_desc* d, int32_t nseg, const int32_t* Ns) {
    if (!d || !Ns || nseg < 1 || nseg > CTGAN_WGRAD_MAX_SEGS) return 0;
    if (!g_force_generic && ctgan_fewch_handles(d)) {        // few-channel convs: up to two segments in the direct kernel
        ctgan_conv_desc dd = *d;
        dd.N = 0;
        for (int i = 0; i < nseg; ++i) dd.N += Ns[i];
        return nseg <= 2 ? ctgan_fewch_wgrad_workspace(&dd) : 0;
    }
    const MultiPlan m = multi_plan(d, nseg, Ns);
    return (size_t)m.splits * ((size_t)d->R * d->S * d->C + 1) * d->K * sizeof(float);
}

namespace {
// validates one multi-segment weight gradient and fills its kernel parameters / plan (no launch)
int prepare_multi(const ctgan_conv_desc* d, int32_t nseg, const float* const* xs, const float* const* dys, const int32_t* Ns,
                  const int32_t* seg_flags, const float* dw, const float* db, WgradParams& p, MultiPlan& m, const char* who) {
    int rc = check_desc(d, who);
    if (rc) return rc;
    if (!xs || !dys || !Ns || !seg_flags || !dw || nseg < 1 || nseg > CTGAN_WGRAD_MAX_SEGS)
        return ctgan_fail(CTGAN_E_BADARG, "%s: bad argument", who);
    if (g_force_generic || ctgan_is_small_linear(d) || ctgan_fewch_handles(d))
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "%s: shape is served by another kernel family", who);
    p.g = geom_from_x(d);
    const Geom& g = p.g;
    p.X = xs[0]; p.DY = dys[0]; p.OUT = nullptr;
    p.Mtot = d->R * d->S * d->C; p.Ng = d->K;
    p.dy_n = d->ys[0]; p.dy_k = d->ys[1]; p.dy_p = d->ys[2]; p.dy_q = d->ys[3];
    p.relu_x = 0; p.x_bytes = p.dy_bytes = 0;
    m = multi_plan(d, nseg, Ns);
    const WPlan& w = m.w;
    bool ok = (g.C % 32 == 0) && g.s_c == 1 && (g.s_n % 4 == 0) && (g.s_h % 4 == 0) && (g.s_w % 4 == 0) && p.dy_k == 1 && (p.Ng % 4 == 0) &&
              (p.dy_n % 4 == 0) && (p.dy_p % 4 == 0) && (p.dy_q % 4 == 0) &&
              (w.tile == W128x128 || w.tile == W64x128 || w.tile == W64x64 || w.tile == W32x128);
    long long kg = 0;
    int split0 = 0;
    bool any_bias = false;
    for (int i = 0; i < nseg && ok; ++i) {
        if (!xs[i] || !dys[i] || Ns[i] <= 0) return ctgan_fail(CTGAN_E_BADARG, "%s: bad segment %d", who, i);
        ok = ok && ((reinterpret_cast<uintptr_t>(xs[i]) | reinterpret_cast<uintptr_t>(dys[i])) & 15) == 0;
        const long long x_elems = (long long)(Ns[i] - 1) * g.s_n + (long long)(g.H - 1) * g.s_h + (long long)(g.W - 1) * g.s_w + g.C;
        const long long y_elems = (long long)(Ns[i] - 1) * p.dy_n + (long long)(g.P - 1) * p.dy_p + (long long)(g.Q - 1) * p.dy_q + p.Ng;
        ok = ok && x_elems * 4 < (1LL << 32) && y_elems * 4 < (1LL << 32);
        WgradParams::Seg& sg = p.seg[i];
        sg.X = xs[i]; sg.DY = dys[i]; sg.Kg = Ns[i] * d->P * d->Q; sg.split0 = split0;
        sg.relu_x = (seg_flags[i] & CTGAN_IN_RELU) ? 1 : 0; sg.bias = (seg_flags[i] & CTGAN_WGRAD_SEG_BIAS) ? 1 : 0;
        sg.x_bytes = (unsigned)(x_elems * 4); sg.dy_bytes = (unsigned)(y_elems * 4);
        any_bias = any_bias || sg.bias;
        split0 += m.seg_splits[i];
        kg += sg.Kg;
    }
    if (!ok) return ctgan_fail(CTGAN_E_UNSUPPORTED, "%s: operands do not qualify for the pipelined kernel", who);
    if (any_bias != (db != nullptr)) return ctgan_fail(CTGAN_E_BADARG, "%s: db must be given iff a segment carries the bias flag", who);
    p.nseg = nseg; p.Kg = (int)kg;
    return CTGAN_OK;
}

template <int WM, int WN, int TM, int TN>
int launch_wgrad_pipe_group(const WgradGroupParams& gp, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem_bytes = 2 * (size_t)(BK * BM + BK * BN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_pipe_group_kernel<WM, WN, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
        if (e != hipSuccess) return ctgan_fail(CTGAN_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    snprintf(g_last_kernel, sizeof g_last_kernel, "igemm_wgrad_pipe_group<%dx%d,n%d>", BM, BN, gp.n);
    ctgan_set_last_symbol("igemm_wgrad_pipe_group_kernel<%d, %d, %d, %d>", WM, WN, TM, TN);
    hipLaunchKernelGGL((igemm_wgrad_pipe_group_kernel<WM, WN, TM, TN>), dim3(gp.block0[gp.n]), dim3(64 * WM * WN), smem_bytes, st, gp);
    return ctgan_check_launch("igemm_wgrad_pipe_group");
}
}  // namespace

extern "C" int ctgan_conv2d_wgrad_multi(const ctgan_conv_desc* d, int32_t nseg, const float* const* xs, const float* const* dys,
                                        const int32_t* Ns, const int32_t* seg_flags, float* dw, float* db, void* ws, size_t ws_bytes,
                                        ctgan_stream_t stream) {
    if (d && xs && dys && Ns && seg_flags && dw && nseg >= 1 && nseg <= 2 && !g_force_generic && ctgan_fewch_handles(d)) {
        const int b0 = (seg_flags[0] & CTGAN_WGRAD_SEG_BIAS) ? 1 : 0, b1 = (nseg > 1 && (seg_flags[1] & CTGAN_WGRAD_SEG_BIAS)) ? 1 : 0;
        if ((b0 || b1) != (db != nullptr)) return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad_multi: db must be given iff a segment carries the bias flag");
        const int r = ctgan_fewch_wgrad2(d, xs[0], dys[0], Ns[0], (seg_flags[0] & CTGAN_IN_RELU) ? 1 : 0, b0, nseg > 1 ? xs[1] : nullptr,
                                         nseg > 1 ? dys[1] : nullptr, nseg > 1 ? Ns[1] : 0, (nseg > 1 && (seg_flags[1] & CTGAN_IN_RELU)) ? 1 : 0, b1,
                                         dw, db, ws, ws_bytes, static_cast<hipStream_t>(stream));
        if (r < 0) return r;
        if (r == 1) return CTGAN_OK;
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "conv2d_wgrad_multi: the few-channel kernel declined these operands");
    }
    WgradParams p;
    MultiPlan m;
    int rc = prepare_multi(d, nseg, xs, dys, Ns, seg_flags, dw, db, p, m, "conv2d_wgrad_multi");
    if (rc) return rc;
    const WPlan& w = m.w;
    const size_t need = (size_t)m.splits * (p.Mtot + 1) * p.Ng * sizeof(float);
    if (!ws || ws_bytes < need) return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad_multi: workspace too small (%zu < %zu)", ws_bytes, need);
    WPlan w2 = w;
    w2.splits = m.splits;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (w.tile == W128x128) rc = launch_wgrad_pipe<2, 2, 2, 2>(p, w2, dw, db, ws, st);
    else if (w.tile == W64x128) rc = launch_wgrad_pipe<1, 4, 2, 1>(p, w2, dw, db, ws, st);
    else if (w.tile == W32x128) rc = launch_wgrad_pipe<1, 4, 1, 1>(p, w2, dw, db, ws, st);
    else rc = launch_wgrad_pipe<2, 2, 1, 1>(p, w2, dw, db, ws, st);
    return rc;
}

// ---- grouped weight gradients: every deferred weight gradient of a step in one launch per tile configuration, and ONE
// launch for all their split-K reductions --------------------------------------------------------------------------------
namespace {
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// A grouped launch fills the chip with the workgroups of ALL its problems, so each problem can take fewer, longer splits than
// the stand-alone plan (which fills the chip by itself): fewer partial slabs to write and to reduce.
void coarsen_plan(MultiPlan& m, WgradParams& p, int div) {
    if (div == 100) return;
    int chunk = (int)((long long)m.w.chunk * div / 100);
    chunk = ((chunk + BK - 1) / BK) * BK;
    if (chunk < 4 * BK) chunk = 4 * BK;
    m.w.chunk = chunk;
    m.splits = 0;
    for (int i = 0; i < p.nseg; ++i) {
        p.seg[i].split0 = m.splits;
        m.seg_splits[i] = (p.seg[i].Kg + chunk - 1) / chunk;
        m.splits += m.seg_splits[i];
    }
}
// (coarser or finer split-K chunks inside a grouped launch were measured - 2x coarser -1.5 %, 2x finer -1.2 % - and the stand-alone plan
// kept: group_div() == 100)
int group_div() { return 100; }
}
extern "C" size_t ctgan_conv2d_wgrad_group_workspace_bytes(const ctgan_wgrad_group* groups, int32_t n) {
    if (!groups || n < 1) return 0;
    size_t tot = 0;
    for (int i = 0; i < n; ++i) {
        const ctgan_wgrad_group& G = groups[i];
        if (G.nseg < 1 || G.nseg > CTGAN_WGRAD_MAX_SEGS) return 0;
        MultiPlan m = multi_plan(&G.d, G.nseg, G.Ns);
        WgradParams p;
        p.nseg = G.nseg;
        for (int k = 0; k < G.nseg; ++k) p.seg[k].Kg = G.Ns[k] * G.d.P * G.d.Q;
        if (n > 1) coarsen_plan(m, p, group_div());
        tot += align256((size_t)m.splits * ((size_t)G.d.R * G.d.S * G.d.C + 1) * G.d.K * sizeof(float));
    }
    return tot;
}

extern "C" int ctgan_conv2d_wgrad_group_tile(const ctgan_wgrad_group* G) {
    // which of the (up to four) launches of a grouped call problem G rides in: index into the launch order {128x128, 64x128, 64x64, 32x128}
    if (!G || G->nseg < 1 || G->nseg > CTGAN_WGRAD_MAX_SEGS) return -1;
    const MultiPlan m = multi_plan(&G->d, G->nseg, G->Ns);
    static const WTile order[4] = {W128x128, W64x128, W64x64, W32x128};
    for (int t = 0; t < 4; ++t) if (m.w.tile == order[t]) return t;
    return -1;
}

extern "C" int ctgan_conv2d_wgrad_group(const ctgan_wgrad_group* groups, int32_t n, void* ws, size_t ws_bytes, ctgan_stream_t stream) {
    return ctgan_conv2d_wgrad_group_ex(groups, n, ws, ws_bytes, CTGAN_WGRAD_GROUP_GEMM | CTGAN_WGRAD_GROUP_REDUCE, stream);
}

extern "C" int ctgan_conv2d_wgrad_group_ex(const ctgan_wgrad_group* groups, int32_t n, void* ws, size_t ws_bytes, int phases,
                                           ctgan_stream_t stream) {
    if (!groups || n < 1 || n > CTGAN_WGRAD_GROUP_LIMIT) return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad_group: bad argument");
    static thread_local MultiPlan M[CTGAN_WGRAD_GROUP_LIMIT];
    static thread_local WgradParams PT[CTGAN_WGRAD_GROUP_LIMIT];
    size_t off = 0;
    for (int i = 0; i < n; ++i) {                        // validate everything before the first launch
        const ctgan_wgrad_group& G = groups[i];
        int rc = prepare_multi(&G.d, G.nseg, G.xs, G.dys, G.Ns, G.seg_flags, G.dw, G.db, PT[i], M[i], "conv2d_wgrad_group");
        if (rc) return rc;
        if (n > 1) coarsen_plan(M[i], PT[i], group_div());
        const size_t need = (size_t)M[i].splits * (PT[i].Mtot + 1) * PT[i].Ng * sizeof(float);
        if (!ws || off + need > ws_bytes) return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad_group: workspace too small");
        PT[i].OUT = reinterpret_cast<float*>(static_cast<char*>(ws) + off);
        PT[i].chunk = M[i].w.chunk;
        PT[i].with_bias = G.db ? 1 : 0;
        off += align256(need);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const WTile order[4] = {W128x128, W64x128, W64x64, W32x128};
    for (int t = 0; t < 4 && (phases & CTGAN_WGRAD_GROUP_GEMM); ++t) {
        if ((phases & CTGAN_WGRAD_GROUP_TILE_MASK) && !(phases & (CTGAN_WGRAD_GROUP_TILE0 << t))) continue;      // one tile configuration only
        int idx[CTGAN_WGRAD_GROUP_LIMIT], cnt = 0;
        for (int i = 0; i < n; ++i) if (M[i].w.tile == order[t]) idx[cnt++] = i;
        // longest chunk first: the workgroups that run longest start first
        for (int a = 1; a < cnt; ++a)
            for (int b = a; b > 0 && M[idx[b]].w.chunk > M[idx[b - 1]].w.chunk; --b) { const int x = idx[b]; idx[b] = idx[b - 1]; idx[b - 1] = x; }
        for (int base = 0; base < cnt; base += CTGAN_WGRAD_GROUP_MAX) {
            WgradGroupParams gp;
            gp.n = (cnt - base) < CTGAN_WGRAD_GROUP_MAX ? (cnt - base) : CTGAN_WGRAD_GROUP_MAX;
            int b0 = 0;
            for (int k = 0; k < CTGAN_WGRAD_GROUP_MAX; ++k) {
                if (k < gp.n) {
                    const int i = idx[base + k];
                    gp.block0[k] = b0; gp.tiles[k] = M[i].w.tiles; gp.p[k] = PT[i];
                    b0 += M[i].w.tiles * M[i].splits;
                } else {
                    gp.block0[k] = b0; gp.tiles[k] = 1; gp.p[k] = PT[idx[base]];
                }
            }
            gp.block0[CTGAN_WGRAD_GROUP_MAX] = b0;
            for (int k = gp.n; k <= CTGAN_WGRAD_GROUP_MAX; ++k) gp.block0[k] = b0;
            int rc;
            if (order[t] == W128x128) rc = launch_wgrad_pipe_group<2, 2, 2, 2>(gp, st);
            else if (order[t] == W64x128) rc = launch_wgrad_pipe_group<1, 4, 2, 1>(gp, st);
            else if (order[t] == W32x128) rc = launch_wgrad_pipe_group<1, 4, 1, 1>(gp, st);
            else rc = launch_wgrad_pipe_group<2, 2, 1, 1>(gp, st);
            if (rc) return rc;
        }
    }
    for (int base = 0; base < n && (phases & CTGAN_WGRAD_GROUP_REDUCE); base += CTGAN_REDUCE_BATCH) {
        ReduceJobs jobs;
        jobs.n = (n - base) < CTGAN_REDUCE_BATCH ? (n - base) : CTGAN_REDUCE_BATCH;
        jobs.pad = 0;
        long long max_n = 0;
        for (int k = 0; k < CTGAN_REDUCE_BATCH; ++k) {
            const int i = base + (k < jobs.n ? k : 0);
            const long long n_main = (long long)PT[i].Mtot * PT[i].Ng, nn = n_main + (groups[i].db ? PT[i].Ng : 0);
            jobs.j[k].part = PT[i].OUT; jobs.j[k].out = groups[i].dw; jobs.j[k].out2 = groups[i].db ? groups[i].db : groups[i].dw;
            jobs.j[k].add = groups[i].add_dw; jobs.j[k].add2 = groups[i].db ? groups[i].add_db : nullptr;
            jobs.j[k].n = nn; jobs.j[k].n_main = n_main; jobs.j[k].splits = M[i].splits; jobs.j[k].pad = 0;
            if (k < jobs.n && nn > max_n) max_n = nn;
        }
        hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3((unsigned)((max_n / 4 + 255) / 256), jobs.n), dim3(256), 0, st, jobs);
        int rc = ctgan_check_launch("splitk_reduce_batch");
        if (rc) return rc;
    }
    if (phases & CTGAN_WGRAD_GROUP_GEMM)
        snprintf(g_last_kernel, sizeof g_last_kernel, "igemm_wgrad_pipe_group<n%d>", n);      // (the symbol stays that of the last grouped launch)
    return CTGAN_OK;
}
