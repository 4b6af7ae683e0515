// fewch.hip - direct (VALU) kernels for convolutions with <= 4 channels on one side.
//
// The CT-WGAN nets touch 3-channel images at both ends: the critic's first block (3 -> 128, 3x3 and the 1x1
// shortcut; TF/CT_gan_cifar_resnet.py:143-153), the generator's output conv (128 -> 3, :165) and, through
// the gradient penalty, their data gradients.  As implicit GEMMs these have one dimension of 3 (of a 32-wide
// MFMA tile) or a reduction depth of 27; they are HBM-bound streaming problems, so they run as plain fp32 FMA
// kernels whose job is to touch every byte of the wide ("many"-channel) tensor exactly once, coalesced:
//
//   f2m   : few -> many   y[n,p,q,k]  = sum_{r,s,c<CS} X(n, p*st-pt+r, q*st-pl+s, c) * w(r,s,c,k)
//           (first critic conv forward; data gradient of the generator's output conv)
//           the few-channel image band sits in LDS (zero padding materialised), each thread owns 4 output
//           channels and keeps its R*S*CS filter slice in registers; stores are 16 B per lane.
//   m2f   : many -> few   y[n,j,p,q]  = sum_{r,s,c} X(n, p-pt+r, q-pl+s, c) * w(r,s,c,j)            (stride 1)
//           (generator output conv forward; data gradient of the first critic conv)
//           the wide band sits in LDS once; a group of C/4 lanes owns one pixel (4 channels per lane, filter
//           slice in registers) and reduces its JS partial sums with cross-lane adds.
//   wgrad : dW(r,s,few,many) = sum_px MANY(px) * FEW(px shifted by the tap)
//           one 16-B load of the wide tensor per lane and pixel, the few-channel band in LDS (one 16-B LDS read
//           per tap), R*S*JS float4 accumulators per lane; per-workgroup partial filters are combined in a fixed
//           order (LDS tree inside the workgroup, slab reduction across workgroups) => deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int NT = 256;

// packed fp32 FMA (v_pk_fma_f32): two lanes of fp32 per VALU lane and cycle
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
struct Acc4 { f32x2 lo, hi; };
__device__ __forceinline__ void fma4(float s, const Acc4& w, Acc4& a) {
    const f32x2 sv = {s, s};
    a.lo = pkfma(sv, w.lo, a.lo); a.hi = pkfma(sv, w.hi, a.hi);
}
__device__ __forceinline__ float4 to_f4(const Acc4& a) { return make_float4(a.lo.x, a.lo.y, a.hi.x, a.hi.y); }
__device__ __forceinline__ Acc4 to_acc(const float4& v) { Acc4 a; a.lo = f32x2{v.x, v.y}; a.hi = f32x2{v.z, v.w}; return a; }

// v + (v of the DPP-selected lane): one v_add_f32 with a DPP operand
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, t);
}
// sum over a 16-lane row, valid in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror
    return v;
}

// ------------------------------------------------------------------------------------------ few -> many
struct F2MParams {
    const float* x; long long xs_n, xs_c, xs_h, xs_w; int H, W;      // few-channel input, any strides
    float* y; long long ys_n, ys_p, ys_q; int P, Q, KM;               // many-channel output, unit channel stride
    const float* w; long long w_off, ws_r, ws_s, ws_c, ws_k;           // w(r,s,c,k) = w[w_off + r*ws_r + s*ws_s + c*ws_c + k*ws_k]
    const float* bias; const float* resid;
    const float* mask;                                                 // result kept where mask > 0 (strides of y; before resid)
    int N, stride, pad_t, pad_l, relu, relu_in, band;                  // band = output rows per workgroup
};

template <int R, int S, int CS>
__global__ __launch_bounds__(NT) void f2m_kernel(const F2MParams p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];       // [CS][TR][TW]
    const int tid = threadIdx.x;
    const int KQ = p.KM >> 2, PXP = NT / KQ;
    const int bands = (p.P + p.band - 1) / p.band;
    const int n = blockIdx.x / bands, b = blockIdx.x - n * bands;
    const int p0 = b * p.band, np = min(p.band, p.P - p0);
    const int TR = (p.band - 1) * p.stride + R, TW = (p.Q - 1) * p.stride + S;
    const int ih0 = p0 * p.stride - p.pad_t, iw0 = -p.pad_l;
    for (int i = tid; i < CS * TR * TW; i += NT) {
        const int c = i / (TR * TW), rem = i - c * TR * TW, tr = rem / TW, tw = rem - tr * TW;
        const int ih = ih0 + tr, iw = iw0 + tw;
        float v = 0.f;
        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            v = p.x[n * p.xs_n + c * p.xs_c + ih * p.xs_h + iw * p.xs_w];
        tile[i] = p.relu_in ? fmaxf(v, 0.f) : v;
    }
    const int kq = tid % KQ, pl = tid / KQ;
    Acc4 wr[R * S * CS];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const float* q = p.w + p.w_off + r * p.ws_r + s * p.ws_s + c * p.ws_c + (long long)(kq * 4) * p.ws_k;
                if (p.ws_k == 1) wr[(r * S + s) * CS + c] = to_acc(*reinterpret_cast<const float4*>(q));
                else wr[(r * S + s) * CS + c] = to_acc(make_float4(q[0], q[p.ws_k], q[2 * p.ws_k], q[3 * p.ws_k]));
            }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + kq * 4);
    __syncthreads();
    const int npx = np * p.Q;
    int pr = 0, qc = pl;
    while (qc >= p.Q) { qc -= p.Q; ++pr; }
    for (int px = pl; px < npx; px += PXP) {
        Acc4 acc = to_acc(bv);
        const float* base = tile + (pr * p.stride) * TW + qc * p.stride;
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float* row = base + (c * TR + r) * TW;
#pragma unroll
                for (int s = 0; s < S; ++s) fma4(row[s], wr[(r * S + s) * CS + c], acc);
            }
        const long long off = n * p.ys_n + (long long)(p0 + pr) * p.ys_p + (long long)qc * p.ys_q + kq * 4;
        float4 o = to_f4(acc);
        if (p.mask) {
            const float4 mv = *reinterpret_cast<const float4*>(p.mask + off);
            o.x = mv.x > 0.f ? o.x : 0.f; o.y = mv.y > 0.f ? o.y : 0.f; o.z = mv.z > 0.f ? o.z : 0.f; o.w = mv.w > 0.f ? o.w : 0.f;
        }
        if (p.resid) {
            const float4 rv = *reinterpret_cast<const float4*>(p.resid + off);
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(p.y + off) = o;
        qc += PXP;
        while (qc >= p.Q) { qc -= p.Q; ++pr; }
    }
}

// ------------------------------------------------------------------------------------------ many -> few
struct M2FParams {
    const float* x; long long xs_n, xs_h, xs_w; int H, W, CM;         // many-channel input, channels-last
    float* y; long long ys_n, ys_c, ys_p, ys_q; int P, Q;              // few-channel output, any strides
    const float* w; long long w_off, ws_r, ws_s, ws_c, ws_j;            // w(r,s,c_many,j_few)
    const float* bias;
    int N, pad_t, pad_l, band, total, relu_in;                          // total = N * bands workgroup tasks
    int dbg;                                                            // perf diagnosis (env CTGAN_M2F_DBG): 1 no row loads, 2 no compute
    // m2f_px_kernel only (PXParams): batch norm of the input while it is staged, tanh of the result
    const float* bn_mean = nullptr; const float* bn_rstd = nullptr; const float* bn_scale = nullptr; const float* bn_offset = nullptr;
    int bn_per = 1, tanh_out = 0;
};

template <int R, int S, int JS>
__global__ __launch_bounds__(NT) void m2f_kernel(const M2FParams p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];       // [TR][TW][CM]
    const int tid = threadIdx.x;
    const int LPP = p.CM >> 2, GROUPS = NT / LPP;                      // lanes per pixel (16 / 32 / 64)
    const int l = tid % LPP, grp = tid / LPP;
    const int bands = (p.P + p.band - 1) / p.band;
    const int TR = p.band + R - 1, TW = p.Q + S - 1;
    f32x2 wlo[R * S][JS], whi[R * S][JS];
#pragma unroll
    for (int t = 0; t < R * S; ++t)
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            const float* q = p.w + p.w_off + (t / S) * p.ws_r + (t % S) * p.ws_s + (long long)(l * 4) * p.ws_c + j * p.ws_j;
            wlo[t][j] = f32x2{q[0], q[p.ws_c]}; whi[t][j] = f32x2{q[2 * p.ws_c], q[3 * p.ws_c]};
        }
    float bj[JS];
#pragma unroll
    for (int j = 0; j < JS; ++j) bj[j] = p.bias ? p.bias[j] : 0.f;

    for (int task = blockIdx.x; task < p.total; task += gridDim.x) {
        const int n = task / bands, b = task - n * bands;
        const int p0 = b * p.band, np = min(p.band, p.P - p0);
        __syncthreads();                                                 // previous band fully consumed
        for (int i = tid; i < TR * TW * LPP; i += NT) {
            const int c4 = i % LPP, pix = i / LPP, tr = pix / TW, tw = pix - tr * TW;
            const int ih = p0 - p.pad_t + tr, iw = tw - p.pad_l;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                v = *reinterpret_cast<const float4*>(p.x + n * p.xs_n + ih * p.xs_h + iw * p.xs_w + c4 * 4);
            if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4*>(tile)[i] = v;
        }
        __syncthreads();
        const int npx = np * p.Q;
        int pr = 0, qc = grp;
        while (qc >= p.Q) { qc -= p.Q; ++pr; }
        for (int px = grp; px < npx; px += GROUPS) {
            f32x2 acc[JS];
#pragma unroll
            for (int j = 0; j < JS; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float4 xv = reinterpret_cast<const float4*>(tile)[((pr + r) * TW + qc + s) * LPP + l];
                    const f32x2 xlo = {xv.x, xv.y}, xhi = {xv.z, xv.w};
#pragma unroll
                    for (int j = 0; j < JS; ++j) acc[j] = pkfma(xhi, whi[r * S + s][j], pkfma(xlo, wlo[r * S + s][j], acc[j]));
                }
            float red[JS];
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                float v = row16_sum(acc[j].x + acc[j].y);                  // 16-lane rows by DPP, the rest by one permute each
                if (LPP >= 32) v += __shfl_xor(v, 16, 64);
                if (LPP >= 64) v += __shfl_xor(v, 32, 64);
                red[j] = v;
            }
            if (l == 0) {
                const long long off = n * p.ys_n + (long long)(p0 + pr) * p.ys_p + (long long)qc * p.ys_q;
#pragma unroll
                for (int j = 0; j < JS; ++j) p.y[off + j * p.ys_c] = red[j] + bj[j];
            }
            qc += GROUPS;
            while (qc >= p.Q) { qc -= p.Q; ++pr; }
        }
    }
}

// ---- many -> few, 3x3, 128 channels: row-ring variant ----------------------------------------------------------
// The band kernel above stages a tile, waits, computes, and so exposes the full load latency once per band.  Here
// a workgroup walks a strip of output rows with a ring of 8 input-row slots in LDS that is filled by DIRECT
// global->LDS loads (global_load_lds_dwordx4: no registers, no LDS store instruction): the rows of the next 4
// iterations are always in flight while the current row is multiplied, and every input row is fetched once per
// strip.  vmcnt is managed by hand (the compiler does not see asm loads): nothing else touches vector memory inside
// the loop - the filter slice is loaded first, the outputs are parked in LDS and written after the loop.
constexpr int RING_NT = 512, RING_SLOTS = 8, RING_AHEAD = 3;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_load16(const void* gptr, unsigned lds_byte_off /* wave-uniform */) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte_off) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void wait_vmcnt(int n) {      // n <= 8 here
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0xF70 | 0); break;
        case 1: __builtin_amdgcn_s_waitcnt(0xF70 | 1); break;
        case 2: __builtin_amdgcn_s_waitcnt(0xF70 | 2); break;
        case 3: __builtin_amdgcn_s_waitcnt(0xF70 | 3); break;
        case 4: __builtin_amdgcn_s_waitcnt(0xF70 | 4); break;
        case 5: __builtin_amdgcn_s_waitcnt(0xF70 | 5); break;
        case 6: __builtin_amdgcn_s_waitcnt(0xF70 | 6); break;
        default: __builtin_amdgcn_s_waitcnt(0xF70 | 0); break;
    }
}

template <int JS>
__global__ __launch_bounds__(RING_NT) void m2f_ring_kernel(const M2FParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [8 slots][W+2 px][128 ch] then [band][JS][W] outputs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = tid & 31, grp = tid >> 5;                            // 16 pixel groups of 32 lanes (4 channels per lane)
    const int W = p.W, H = p.H;
    const int strips = (p.P + p.band - 1) / p.band;
    const int n = blockIdx.x / strips, sidx = blockIdx.x - n * strips;
    const int p0 = sidx * p.band, np = min(p.band, p.P - p0);
    const int slot_f4 = (W + 2) * 32;
    float4* tile4 = reinterpret_cast<float4*>(smem);
    float* obuf = smem + RING_SLOTS * slot_f4 * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < RING_SLOTS * 64; i += RING_NT) {             // halo columns: never written by the row loads
        const int slot = i >> 6, side = (i >> 5) & 1, c = i & 31;
        tile4[slot * slot_f4 + (side ? (W + 1) * 32 : 0) + c] = zero4;
    }
    // filter: staged once per workgroup through LDS ([tap][j][128 channels], ~7 coalesced loads per thread) - a direct
    // per-thread gather is 108 dword loads per thread and cost more than the whole strip's arithmetic
    float* wbuf = obuf + p.band * JS * W;
    for (int i = tid; i < 9 * JS * 128; i += RING_NT) {
        const int c = i & 127, tj = i >> 7, j = tj % JS, t = tj / JS;
        wbuf[i] = p.w[p.w_off + (t / 3) * p.ws_r + (t % 3) * p.ws_s + (long long)c * p.ws_c + j * p.ws_j];
    }
    __builtin_amdgcn_s_waitcnt(0xF70 | 0);                             // all ordinary loads retired before any row load is issued
    __syncthreads();
    f32x2 wlo[9][JS], whi[9][JS];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            const float4 q = *reinterpret_cast<const float4*>(&wbuf[(t * JS + j) * 128 + l * 4]);
            wlo[t][j] = f32x2{q.x, q.y}; whi[t][j] = f32x2{q.z, q.w};
        }
    __builtin_amdgcn_sched_barrier(0);
    const int LPR = W >> 4;                                            // direct loads per wave and row (2 pixels = 1 KB each)
    const float* img = p.x + (long long)n * p.xs_n;
    auto issue_row = [&](int r) {                                      // input row r into slot (r+1)%8; zero rows outside the image
        const int slot = (r + 1) & (RING_SLOTS - 1);
        if (r < 0 || r >= H) {
            for (int i = tid; i < W * 32; i += RING_NT) tile4[slot * slot_f4 + 32 + i] = zero4;
            return;
        }
        if (p.dbg & 1) return;
        const float* rowp = img + (long long)r * p.xs_h + (lane & 31) * 4;
        for (int c = 0; c < LPR; ++c) {
            const int chunk = c * 8 + wave;
            const unsigned dst = (unsigned)((slot * slot_f4 + (1 + 2 * chunk) * 32) * 16);
            lds_load16(rowp + (long long)(2 * chunk + (lane >> 5)) * p.xs_w, __builtin_amdgcn_readfirstlane(dst));
        }
    };
    const int last_row = p0 + np;                                      // highest input row any output of the strip reads
    for (int r = p0 - 1; r <= p0 + RING_AHEAD && r <= last_row; ++r) issue_row(r);

    for (int pp = 0; pp < np; ++pp) {
        const int prow = p0 + pp;
        if (prow + 1 + RING_AHEAD <= last_row) issue_row(prow + 1 + RING_AHEAD);
        int after = 0;                                                 // loads issued after those of row prow+1
        for (int r = prow + 2; r <= prow + 1 + RING_AHEAD && r <= last_row; ++r) after += (r >= 0 && r < H) ? LPR : 0;
        wait_vmcnt(after);
        __syncthreads();                                               // rows prow-1 .. prow+1 complete and visible to every wave
        const float4* r0 = tile4 + ((prow + 0) & (RING_SLOTS - 1)) * slot_f4;      // slot of input row prow-1
        const float4* r1 = tile4 + ((prow + 1) & (RING_SLOTS - 1)) * slot_f4;
        const float4* r2 = tile4 + ((prow + 2) & (RING_SLOTS - 1)) * slot_f4;
        for (int qc = grp; qc < W && !(p.dbg & 2); qc += 16) {
            f32x2 acc[JS];
#pragma unroll
            for (int j = 0; j < JS; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float4* rowp = r == 0 ? r0 : (r == 1 ? r1 : r2);
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    float4 xv = rowp[(qc + s2) * 32 + l];
                    if (p.relu_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
                    const f32x2 xlo = {xv.x, xv.y}, xhi = {xv.z, xv.w};
#pragma unroll
                    for (int j = 0; j < JS; ++j) acc[j] = pkfma(xhi, whi[r * 3 + s2][j], pkfma(xlo, wlo[r * 3 + s2][j], acc[j]));
                }
            }
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                float v = row16_sum(acc[j].x + acc[j].y);
                v += __shfl_xor(v, 16, 64);
                if (l == 0) obuf[(pp * JS + j) * W + qc] = v;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < np * JS * W; i += RING_NT) {
        const int qc = i % W, j = (i / W) % JS, pp = i / (W * JS);
        p.y[n * p.ys_n + (long long)j * p.ys_c + (long long)(p0 + pp) * p.ys_p + (long long)qc * p.ys_q] = obuf[i] + (p.bias ? p.bias[j] : 0.f);
    }
}


// ---- many -> few, 3x3, 32-pixel rows: one output pixel per lane ("px" variant, round 5) ---------------------------------
// The lane-group kernels above give a pixel to C/4 lanes and finish every pixel with cross-lane reductions (DPP adds + a bpermute per
// output channel): a chain of dependent cross-lane operations per pixel at two waves per SIMD - the ring kernel spent 26 of its 48 us
// (128 rows) in arithmetic that overlapped nothing.  Here a lane OWNS a pixel: 256 threads = an 8 x 32 output tile, the
// many-channel operand is staged 32 channels at a time into LDS (10 x 34 halo pixels, 36-float pixel pitch: a 16-lane group of a
// ds_read_b128 touches 16 different 4-bank slots) and every lane walks the 9 taps x 32 channels of the chunk.  The filter is the same
// for all pixels: its chunk (9 taps x 8 channel quads x JS outputs, one float4 each) sits in LDS too and is read with WAVE-UNIFORM
// addresses (a broadcast ds_read_b128: one LDS cycle per lane group whatever the lane count) - no filter registers held per lane, no
// reductions, no per-row barriers (two per chunk), 2 * JS independent accumulator chains.  The next chunk's global loads are issued
// into registers before the current chunk's arithmetic and parked in LDS after it.
// (Measured and dropped: the filter as SCALAR operands (s_load_dwordx16 -> v_pk_fma_f32 with SGPR pairs; 4 us per 64 rows of arithmetic) -
// the tap loop needs 96 SGPRs per step and cannot run ahead, so on the cold scalar cache of a fresh launch each of its 36 steps waited an
// L2 round trip: 32 us per launch whatever the batch.)
struct PXParams {
    const float* x; long long xs_n, xs_h, xs_w; int H, CM;
    float* y; long long ys_n, ys_c, ys_p, ys_q;
    const float* w; long long w_off, ws_r, ws_s, ws_c, ws_j;
    const float* bias;
    int relu_in, tiles;
    // round 5: training-mode batch norm of the INPUT applied while it is staged (the generator's output stage under no_grad:
    // tanh(conv(relu(bn(h)))), TF/CT_gan_cifar_resnet.py:164-166) - x' = (x - mean[g]) * rstd[g] * scale + offset per channel, sample n in group
    // n / bn_per; the zero padding stays zero.  bn_mean == NULL: off.  tanh_out: tanh of the result in the epilogue.
    const float* bn_mean; const float* bn_rstd; const float* bn_scale; const float* bn_offset; int bn_per, tanh_out;
};
constexpr int PX_TH = 8, PX_W = 32, PX_CC = 32, PX_PS = 36, PX_NT = 256;
constexpr int PX_PIX = (PX_TH + 2) * (PX_W + 2);            // 340 staged pixels per chunk
constexpr int PX_F4 = PX_PIX * (PX_CC / 4);                 // 2720 float4 per chunk
constexpr int PX_PF = (PX_F4 + PX_NT - 1) / PX_NT;          // 11 prefetch registers (float4) per thread

template <int JS>
__global__ __launch_bounds__(PX_NT) void m2f_px_kernel(const PXParams p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [340 px][36] then the filter chunk [9][8][JS] float4
    float4* wl = reinterpret_cast<float4*>(tile + PX_PIX * PX_PS);
    constexpr int WF4 = 9 * (PX_CC / 4) * JS;
    static_assert(WF4 <= PX_NT, "one filter float4 per thread");
    const int tid = threadIdx.x;
    const int n = blockIdx.x / p.tiles, tb = blockIdx.x - n * p.tiles;
    const int p0 = tb * PX_TH;
    const int row = tid >> 5, col = tid & 31;
    const float* img = p.x + (long long)n * p.xs_n;
    float4 pf[PX_PF], wpf = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int chunk) {
        // (a thread's channel quad is the same for all its items: PX_NT % 8 == 0 - one set of BN coefficients per chunk)
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu, ga = mu, be = mu;
        if (p.bn_mean) {
            const int c = chunk * PX_CC + (tid & 7) * 4, g = n / p.bn_per;
            mu = *reinterpret_cast<const float4*>(p.bn_mean + (long long)g * p.CM + c); rs = *reinterpret_cast<const float4*>(p.bn_rstd + (long long)g * p.CM + c);
            ga = *reinterpret_cast<const float4*>(p.bn_scale + c); be = *reinterpret_cast<const float4*>(p.bn_offset + c);
        }
#pragma unroll
        for (int k = 0; k < PX_PF; ++k) {
            const int i = tid + k * PX_NT;
            const int pix = i >> 3, c4 = i & 7;
            const int tr = pix / (PX_W + 2), tw = pix - tr * (PX_W + 2);
            const int ih = p0 - 1 + tr, iw = tw - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < PX_F4 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)PX_W) {
                v = *reinterpret_cast<const float4*>(img + (long long)ih * p.xs_h + (long long)iw * p.xs_w + chunk * PX_CC + c4 * 4);
                if (p.bn_mean) {      // bn_apply_vec_kernel's operation order: ((x - mean) * rstd) * scale + offset
                    v.x = (v.x - mu.x) * rs.x * ga.x + be.x; v.y = (v.y - mu.y) * rs.y * ga.y + be.y;
                    v.z = (v.z - mu.z) * rs.z * ga.z + be.z; v.w = (v.w - mu.w) * rs.w * ga.w + be.w;
                }
            }
            pf[k] = v;
        }
        if (tid < WF4) {                                               // wl[(t * 8 + c4) * JS + j] = w(t, channels c .. c+3, j)
            const int j = tid % JS, tc = tid / JS, c4 = tc & 7, t = tc >> 3;
            const float* q = p.w + p.w_off + (t / 3) * p.ws_r + (t % 3) * p.ws_s + (long long)(chunk * PX_CC + c4 * 4) * p.ws_c + j * p.ws_j;
            wpf = make_float4(q[0], q[p.ws_c], q[2 * p.ws_c], q[3 * p.ws_c]);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int k = 0; k < PX_PF; ++k) {
            const int i = tid + k * PX_NT;
            if (i < PX_F4) {
                float4 v = pf[k];
                if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(&tile[(i >> 3) * PX_PS + (i & 7) * 4]) = v;
            }
        }
        if (tid < WF4) wl[tid] = wpf;
    };
    f32x2 acc[JS][2];
#pragma unroll
    for (int j = 0; j < JS; ++j) { acc[j][0] = f32x2{0.f, 0.f}; acc[j][1] = f32x2{0.f, 0.f}; }
    const int nchunk = p.CM / PX_CC;
    fetch(0);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();                      // the previous chunk's tile fully consumed
        stash();
        __syncthreads();
        if (chunk + 1 < nchunk) fetch(chunk + 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s2 = t % 3;
            const float* xp = &tile[((row + r) * (PX_W + 2) + col + s2) * PX_PS];
#pragma unroll
            for (int c4 = 0; c4 < PX_CC / 4; ++c4) {
                const float4 xv = *reinterpret_cast<const float4*>(xp + c4 * 4);
#pragma unroll
                for (int j = 0; j < JS; ++j) {
                    const float4 wv = wl[(t * (PX_CC / 4) + c4) * JS + j];
                    acc[j][0] = pkfma(f32x2{xv.x, xv.y}, f32x2{wv.x, wv.y}, acc[j][0]);
                    acc[j][1] = pkfma(f32x2{xv.z, xv.w}, f32x2{wv.z, wv.w}, acc[j][1]);
                }
            }
        }
    }
    if (p0 + row < p.H) {
        const long long off = (long long)n * p.ys_n + (long long)(p0 + row) * p.ys_p + (long long)col * p.ys_q;
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            const f32x2 a = acc[j][0] + acc[j][1];
            const float r = (a.x + a.y) + (p.bias ? p.bias[j] : 0.f);
            p.y[off + j * p.ys_c] = p.tanh_out ? tanhf(r) : r;
        }
    }
}

// ------------------------------------------------------------------------------------------ weight gradient
struct FWParams {
    const float* many; long long ms_n, ms_h, ms_w; int MH, MW, CM;    // wide operand [N, MH, MW, CM], channels-last
    const float* few; long long fs_n, fs_c, fs_h, fs_w; int FH, FW;    // few-channel operand [N, JS, FH, FW], any strides
    int fst;                                                           // few row of (many row h, tap r) = h*fst + off_r[r]
    int off_r[5], off_s[5];
    float* slab;                                                       // [workgroups][n_out]
    int N, band, bands, total, relu_many, relu_few, few_in, with_bias, n_out, n_main;
    // optional second (x, dy) pair of the same geometry and strides (another pass's use of the filter): images [N0, N) come from it
    const float* many2; const float* few2; int N0, relu_many2, relu_few2, bias1, bias2;
};

// few_in  (C small): many = dy, few = x:   dW[(tap*JS + j)*CM + c]   bias (sum of dy) at n_main + c
// few_out (K small): many = x,  few = dy:  dW[(tap*CM + c)*JS + j]   bias (sum of dy) at n_main + j
// Persistent workgroups: each walks its (image, row band) tasks with the accumulators kept in registers, so there
// is one partial filter ("slab") per workgroup, not per band.
template <int R, int S, int JS>
__global__ __launch_bounds__(NT) void fw_wgrad_kernel(const FWParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* tile = reinterpret_cast<float4*>(smem);                     // [TR][TW] few-channel pixels (padded to 4)
    const int tid = threadIdx.x;
    const int LPP = p.CM >> 2, GROUPS = NT / LPP;
    const int l = tid % LPP, grp = tid / LPP;
    int rmin = p.off_r[0], rmax = p.off_r[0], smin = p.off_s[0], smax = p.off_s[0];
#pragma unroll
    for (int r = 1; r < R; ++r) { rmin = min(rmin, p.off_r[r]); rmax = max(rmax, p.off_r[r]); }
#pragma unroll
    for (int s = 1; s < S; ++s) { smin = min(smin, p.off_s[s]); smax = max(smax, p.off_s[s]); }
    const int tw0 = smin, TW = (p.MW - 1) * p.fst + smax - smin + 1;

    Acc4 acc[R * S * JS];
#pragma unroll
    for (int i = 0; i < R * S * JS; ++i) { acc[i].lo = f32x2{0.f, 0.f}; acc[i].hi = f32x2{0.f, 0.f}; }
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int task = blockIdx.x; task < p.total; task += gridDim.x) {
        int n = task / p.bands;
        const int b = task - n * p.bands;
        const bool sg2 = n >= p.N0;
        const float* __restrict__ many = sg2 ? p.many2 : p.many;
        const float* __restrict__ few = sg2 ? p.few2 : p.few;
        const int relu_many = sg2 ? p.relu_many2 : p.relu_many, relu_few = sg2 ? p.relu_few2 : p.relu_few;
        const int seg_bias = sg2 ? p.bias2 : p.bias1;
        if (sg2) n -= p.N0;
        const int h0 = b * p.band, nh = min(p.band, p.MH - h0);
        const int tr0 = h0 * p.fst + rmin, TR = (nh - 1) * p.fst + rmax - rmin + 1;
        __syncthreads();                                                 // previous band fully consumed
        for (int i = tid; i < TR * TW; i += NT) {
            const int tr = i / TW, tw = i - tr * TW;
            const int fh = tr0 + tr, fw = tw0 + tw;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)fh < (unsigned)p.FH && (unsigned)fw < (unsigned)p.FW) {
#pragma unroll
                for (int j = 0; j < JS; ++j) {
                    const float t = few[n * p.fs_n + j * p.fs_c + fh * p.fs_h + fw * p.fs_w];
                    v[j] = relu_few ? fmaxf(t, 0.f) : t;
                }
            }
            tile[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        for (int hr = grp; hr < nh; hr += GROUPS) {
            const float* mrow = many + n * p.ms_n + (long long)(h0 + hr) * p.ms_h + l * 4;
            float4 mv = *reinterpret_cast<const float4*>(mrow);
            for (int w = 0; w < p.MW; ++w) {
                float4 cur = mv;
                if (w + 1 < p.MW) mv = *reinterpret_cast<const float4*>(mrow + (long long)(w + 1) * p.ms_w);
                if (relu_many) { cur.x = fmaxf(cur.x, 0.f); cur.y = fmaxf(cur.y, 0.f); cur.z = fmaxf(cur.z, 0.f); cur.w = fmaxf(cur.w, 0.f); }
                if (p.few_in && seg_bias) { accb.x += cur.x; accb.y += cur.y; accb.z += cur.z; accb.w += cur.w; }
                const Acc4 cm = to_acc(cur);
                const float4* trow = tile + (hr * p.fst - rmin) * TW + (w * p.fst - smin);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float4 fv = trow[p.off_r[r] * TW + p.off_s[s]];
                        const float f[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
                        for (int j = 0; j < JS; ++j) fma4(f[j], cm, acc[(r * S + s) * JS + j]);
                    }
                if (!p.few_in && seg_bias) {                             // bias of the few-channel dy: the pixel itself (tap offset 0)
                    const float4 fv = trow[0];
                    accb.x += fv.x; accb.y += fv.y; accb.z += fv.z; accb.w += fv.w;
                }
            }
        }
    }
    __syncthreads();                                                     // tile no longer needed: LDS becomes the reduction buffer
    float4* red = reinterpret_cast<float4*>(smem);
    constexpr int NA = R * S * JS + 1;
    for (int half = GROUPS >> 1; half >= 1; half >>= 1) {
        if (grp >= half && grp < 2 * half) {
            float4* dst = red + ((grp - half) * LPP + l) * NA;
#pragma unroll
            for (int i = 0; i < R * S * JS; ++i) dst[i] = to_f4(acc[i]);
            dst[R * S * JS] = accb;
        }
        __syncthreads();
        if (grp < half) {
            const float4* src = red + (grp * LPP + l) * NA;
#pragma unroll
            for (int i = 0; i < R * S * JS; ++i) {
                const Acc4 v = to_acc(src[i]);
                acc[i].lo += v.lo; acc[i].hi += v.hi;
            }
            const float4 v = src[R * S * JS];
            accb.x += v.x; accb.y += v.y; accb.z += v.z; accb.w += v.w;
        }
        __syncthreads();
    }
    if (grp == 0) {
        float* out = p.slab + (long long)blockIdx.x * p.n_out;
        if (p.few_in) {
#pragma unroll
            for (int i = 0; i < R * S * JS; ++i) *reinterpret_cast<float4*>(out + (long long)i * p.CM + l * 4) = to_f4(acc[i]);
            if (p.with_bias) *reinterpret_cast<float4*>(out + p.n_main + l * 4) = accb;
        } else {
#pragma unroll
            for (int t = 0; t < R * S; ++t)
#pragma unroll
                for (int j = 0; j < JS; ++j) {
                    const float4 a = to_f4(acc[t * JS + j]);
                    float* o = out + ((long long)t * p.CM + l * 4) * JS + j;
                    o[0] = a.x; o[JS] = a.y; o[2 * JS] = a.z; o[3 * JS] = a.w;
                }
            if (p.with_bias && l == 0) *reinterpret_cast<float4*>(out + p.n_main) = accb;
        }
    }
}

// ---- the same weight gradient on the matrix cores (CM = 128) ------------------------------------------------------------------
// dW[m][c] = sum_px F[px][m] * G[px][c] with m = (tap, few channel) <= 27 rows (padded to the 32 of v_mfma_f32_32x32x2_f32),
// c = the 128 channels of the wide operand, two pixels per MFMA step.  Per step a lane fetches ONE float4 of the wide operand
// (channels 4*l31 .. 4*l31+3 of pixel w + h: a half-wave reads the pixel's 512 contiguous bytes) and uses its four components
// as the B operands of four MFMAs - accumulator t then holds the channels 4*j + t - and ONE value of the zero-padded
// few-channel LDS tile as the A operand (row m of lane l31 = its tap / channel).  Row 31 of A is the constant 1 when the
// segment contributes to the bias of a few_in conv, so the column sums of the wide operand ride along; the few_out bias (sum of
// the few-channel dy) is the sum of the centre-tap A values a lane sees.  The direct-FMA kernel above is VALU-bound at ~2.5x
// this kernel's HBM floor.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int R, int S, int JS>
__global__ __launch_bounds__(NT) void fw_wgrad_mfma_kernel(const FWParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* tile = reinterpret_cast<float4*>(smem);
    const float* tile_f = smem;
    constexpr int M = R * S * JS;
    static_assert(M <= 31, "taps x channels must leave row 31 for the bias");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    int rmin = p.off_r[0], rmax = p.off_r[0], smin = p.off_s[0], smax = p.off_s[0];
#pragma unroll
    for (int r = 1; r < R; ++r) { rmin = min(rmin, p.off_r[r]); rmax = max(rmax, p.off_r[r]); }
#pragma unroll
    for (int s = 1; s < S; ++s) { smin = min(smin, p.off_s[s]); smax = max(smax, p.off_s[s]); }
    const int tw0 = smin, TW = (p.MW - 1) * p.fst + smax - smin + 1;
    // this lane's A row: tap (ar, as) and few channel aj; rows M..30 are zero, row 31 is the bias row
    const int am = l31;
    const int atap = am / JS, aj = am - atap * JS, ar = atap / S, as_ = atap - ar * S;
    const bool a_live = am < M;
    const int a_off = a_live ? ((p.off_r[ar] - rmin) * TW + (p.off_s[as_] - smin)) * 4 + aj : 0;
    const bool a_centre = a_live && !p.few_in && p.off_r[ar] == 0 && p.off_s[as_] == 0;   // few_out bias: the pixel itself

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;

    for (int task = blockIdx.x; task < p.total; task += gridDim.x) {
        int n = task / p.bands;
        const int b = task - n * p.bands;
        const bool sg2 = n >= p.N0;
        const float* __restrict__ many = sg2 ? p.many2 : p.many;
        const float* __restrict__ few = sg2 ? p.few2 : p.few;
        const int relu_many = sg2 ? p.relu_many2 : p.relu_many, relu_few = sg2 ? p.relu_few2 : p.relu_few;
        const int seg_bias = sg2 ? p.bias2 : p.bias1;
        if (sg2) n -= p.N0;
        const int h0 = b * p.band, nh = min(p.band, p.MH - h0);
        const int tr0 = h0 * p.fst + rmin, TR = (nh - 1) * p.fst + rmax - rmin + 1;
        __syncthreads();                                                 // previous band fully consumed
        for (int i = tid; i < TR * TW; i += NT) {
            const int tr = i / TW, tw = i - tr * TW;
            const int fh = tr0 + tr, fw = tw0 + tw;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)fh < (unsigned)p.FH && (unsigned)fw < (unsigned)p.FW) {
#pragma unroll
                for (int j = 0; j < JS; ++j) {
                    const float t = few[n * p.fs_n + j * p.fs_c + fh * p.fs_h + fw * p.fs_w];
                    v[j] = relu_few ? fmaxf(t, 0.f) : t;
                }
            }
            tile[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        const float a_bias = (am == 31 && p.few_in && seg_bias) ? 1.f : 0.f;
        const float c_flag = (a_centre && seg_bias) ? 1.f : 0.f;
        for (int hr = wave; hr < nh; hr += 4) {
            const float* mrow = many + n * p.ms_n + (long long)(h0 + hr) * p.ms_h + l31 * 4 + (long long)h * p.ms_w;
            const float* arow = tile_f + (hr * p.fst * TW + h * p.fst) * 4 + a_off;
            const int npair = p.MW >> 1;                                 // MW even (checked by the host)
            constexpr int UQ = 8;                                        // pixel pairs per batch: 8 float4 loads in flight per lane
            // the loads of batch q0 + UQ are issued before the MFMAs of batch q0 (two register sets): a wave's load round trip no longer
            // sits between its own MFMA bursts (round 5; one batch in flight per wave before)
            float4 gn[UQ];
#pragma unroll
            for (int u = 0; u < UQ; ++u)
                gn[u] = u < npair ? *reinterpret_cast<const float4*>(mrow + (long long)(2 * u) * p.ms_w) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q0 = 0; q0 < npair; q0 += UQ) {
                float4 gv[UQ];
                float av[UQ];
#pragma unroll
                for (int u = 0; u < UQ; ++u) gv[u] = gn[u];
#pragma unroll
                for (int u = 0; u < UQ; ++u) {
                    const int q = q0 + UQ + u;
                    gn[u] = q < npair ? *reinterpret_cast<const float4*>(mrow + (long long)(2 * q) * p.ms_w) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < UQ; ++u) {
                    const int q = q0 + u;
                    av[u] = q < npair ? (a_live ? arow[q * 2 * p.fst * 4] : a_bias) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UQ; ++u) {
                    float4 cur = gv[u];
                    if (relu_many) { cur.x = fmaxf(cur.x, 0.f); cur.y = fmaxf(cur.y, 0.f); cur.z = fmaxf(cur.z, 0.f); cur.w = fmaxf(cur.w, 0.f); }
                    const float a = av[u];
                    bsum += a * c_flag;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cur.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cur.y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cur.z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cur.w, acc[3], 0, 0, 0);
                }
            }
        }
    }
    // combine the four waves through LDS in a fixed order (wave 1, 2, 3 onto wave 0), one wave at a time through ONE 16 KB buffer - the
    // three-wave buffer (49 KB) was what limited the kernel to three workgroups per CU - then wave 0 writes the slab
    float* red = smem;                                                   // [64 regs + 1][64 lanes]
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        __syncthreads();                                                 // the tile / the previous wave's values are consumed
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(t * 16 + e) * 64 + lane] = acc[t][e];
            red[64 * 64 + lane] = bsum;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] += red[(t * 16 + e) * 64 + lane];
            bsum += red[64 * 64 + lane];
        }
    }
    if (wave != 0) return;
    float* out = p.slab + (long long)blockIdx.x * p.n_out;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = (e & 3) + 8 * (e >> 2) + 4 * h;                    // accumulator row of register e in lane half h
        const float4 v = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);      // channels 4*l31 .. 4*l31+3
        if (m < M) {
            if (p.few_in) {
                *reinterpret_cast<float4*>(out + (long long)m * p.CM + l31 * 4) = v;
            } else {
                const int tap = m / JS, j = m - tap * JS;
                float* o = out + ((long long)tap * p.CM + l31 * 4) * JS + j;
                o[0] = v.x; o[JS] = v.y; o[2 * JS] = v.z; o[3 * JS] = v.w;
            }
        } else if (m == 31 && p.few_in && p.with_bias) {
            *reinterpret_cast<float4*>(out + p.n_main + l31 * 4) = v;
        }
    }
    if (!p.few_in && p.with_bias) {                                      // few_out bias: lanes of the centre tap hold sum of dy[., j]
        const float other = __shfl_xor(bsum, 32, 64);                    // the two pixel halves
        const float tot = bsum + other;
        if (a_centre && h == 0) out[p.n_main + aj] = tot;
        if (lane == 0 && JS < 4) for (int j = JS; j < 4; ++j) out[p.n_main + j] = 0.f;
    }
}

// dw = sum over workgroup slabs, fixed order: 16 slab lanes x 16 column groups (float4) per workgroup, then a
// 16-way LDS combine.  n_tot (filter + bias section) is a multiple of 4; db receives the first nb bias values.
__global__ __launch_bounds__(NT) void fw_reduce_kernel(const float* __restrict__ slab, int stride, int blocks, float* __restrict__ dw,
                                                      int n_main, float* __restrict__ db, int nb, int n_tot) {
    __shared__ float4 part[16][16];
    const int cg = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int col = (blockIdx.x * 16 + cg) * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (col < n_tot) {
        int k = sl;
        for (; k + 16 < blocks; k += 32) {
            const float4 v0 = *reinterpret_cast<const float4*>(slab + (long long)k * stride + col);
            const float4 v1 = *reinterpret_cast<const float4*>(slab + (long long)(k + 16) * stride + col);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        }
        if (k < blocks) {
            const float4 v0 = *reinterpret_cast<const float4*>(slab + (long long)k * stride + col);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
    }
    part[sl][cg] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (sl == 0 && col < n_tot) {
        float4 r = part[0][cg];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = part[k][cg]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = col + e;
            if (i < n_main) dw[i] = rv[e];
            else if (i - n_main < nb) db[i - n_main] = rv[e];
        }
    }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
bool many_ok(int c) { return c == 64 || c == 128 || c == 256; }

template <typename KernelT>
int set_smem(KernelT k, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return ctgan_fail(CTGAN_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    return 0;
}

// instantiated (R, S, few) triples: 3x3x3 (CIFAR critic conv 1 / generator output), 1x1x3 (critic shortcut), 5x5x1 (MNIST)
int tap_case(int R, int S, int few) {
    if (R == 3 && S == 3 && few == 3) return 1;
    if (R == 1 && S == 1 && few == 3) return 2;
    if (R == 5 && S == 5 && few == 1) return 3;
    if (R == 3 && S == 3 && few == 1) return 4;
    return 0;
}

int pick_band(int N, int rows, int unit, int min_tasks = 512) {         // rows per workgroup: >= ~512 workgroups when the batch allows it
    int band = unit;
    while (band * 2 <= rows && (long long)N * ((rows + band * 2 - 1) / (band * 2)) >= min_tasks) band *= 2;
    return band;
}
#ifndef FW_TASKS
#define FW_TASKS 512      // (A/B builds) tasks / workgroups of the MFMA weight-gradient kernel
#endif

}  // namespace

// ---- host entry points (called from igemm.hip's C-ABI functions) -------------------------------------------
bool ctgan_fewch_handles(const ctgan_conv_desc* d) {
    if (d->x_up) return false;
    if (d->C <= 4 && tap_case(d->R, d->S, d->C) && many_ok(d->K)) return true;                       // few -> many
    if (d->K <= 4 && d->stride == 1 && tap_case(d->R, d->S, d->K) && many_ok(d->C)) return true;     // many -> few
    return false;
}

// launch helpers -------------------------------------------------------------------------------------------
static int launch_f2m(const F2MParams& p, int R, int S, int CS, hipStream_t st) {
    const int bands = (p.P + p.band - 1) / p.band;
    const int TR = (p.band - 1) * p.stride + R, TW = (p.Q - 1) * p.stride + S;
    const size_t smem = (size_t)CS * TR * TW * sizeof(float);
    const dim3 grid(p.N * bands), blk(NT);
    switch (tap_case(R, S, CS)) {
        case 1: hipLaunchKernelGGL((f2m_kernel<3, 3, 3>), grid, blk, smem, st, p); break;
        case 2: hipLaunchKernelGGL((f2m_kernel<1, 1, 3>), grid, blk, smem, st, p); break;
        case 3: hipLaunchKernelGGL((f2m_kernel<5, 5, 1>), grid, blk, smem, st, p); break;
        case 4: hipLaunchKernelGGL((f2m_kernel<3, 3, 1>), grid, blk, smem, st, p); break;
        default: return ctgan_fail(CTGAN_E_UNSUPPORTED, "fewch f2m: taps");
    }
    ctgan_set_last_symbol("f2m_kernel<%d, %d, %d>", R, S, CS);
    return ctgan_check_launch("fewch_f2m");
}

static bool m2f_ring_ok(const M2FParams& p, int R, int S) {
    return R == 3 && S == 3 && p.CM == 128 && p.pad_t == 1 && p.pad_l == 1 && (p.W == 32 || p.W == 16) && p.P == p.H && p.Q == p.W &&
           (p.xs_w % 4 == 0) && (p.xs_h % 4 == 0) && (p.xs_n % 4 == 0);
}

template <int JS>
static int launch_m2f_ring(M2FParams p, hipStream_t st) {
#ifndef M2F_DBG
#define M2F_DBG 0      // diagnosis builds (tools/build_variant.sh): 1 = no row loads, 2 = no arithmetic (results are then wrong by design)
#endif
    p.dbg = M2F_DBG;
    p.band = p.P >= 8 ? 8 : p.P;                                      // strip rows per workgroup
    const int strips = (p.P + p.band - 1) / p.band;
    const size_t smem = (size_t)RING_SLOTS * (p.W + 2) * 128 * 4 + (size_t)p.band * JS * p.W * 4 + (size_t)9 * JS * 128 * 4;
    int rc = set_smem(&m2f_ring_kernel<JS>, smem);
    if (rc) return rc;
    hipLaunchKernelGGL((m2f_ring_kernel<JS>), dim3(p.N * strips), dim3(RING_NT), smem, st, p);
    ctgan_set_last_symbol("m2f_ring_kernel<%d>", JS);
    return ctgan_check_launch("fewch_m2f_ring");
}

static int g_m2f_px = 1;       // tests / A-B: ctgan_debug_m2f_px(0) puts the 3x3 many -> few convs back on the row-ring kernel
extern "C" void ctgan_debug_m2f_px(int on) { g_m2f_px = on ? 1 : 0; }

static bool m2f_px_ok(const M2FParams& p, int R, int S, int JS) {
    return g_m2f_px && JS == 3 && R == 3 && S == 3 && p.pad_t == 1 && p.pad_l == 1 && p.W == PX_W && p.P == p.H && p.Q == p.W && p.CM % PX_CC == 0 &&
           (p.xs_w % 4 == 0) && (p.xs_h % 4 == 0) && (p.xs_n % 4 == 0);
}

static int launch_m2f_px(const M2FParams& m, hipStream_t st) {
    PXParams p;
    p.x = m.x; p.xs_n = m.xs_n; p.xs_h = m.xs_h; p.xs_w = m.xs_w; p.H = m.H; p.CM = m.CM;
    p.y = m.y; p.ys_n = m.ys_n; p.ys_c = m.ys_c; p.ys_p = m.ys_p; p.ys_q = m.ys_q;
    p.w = m.w; p.w_off = m.w_off; p.ws_r = m.ws_r; p.ws_s = m.ws_s; p.ws_c = m.ws_c; p.ws_j = m.ws_j;
    p.bias = m.bias; p.relu_in = m.relu_in;
    p.bn_mean = m.bn_mean; p.bn_rstd = m.bn_rstd; p.bn_scale = m.bn_scale; p.bn_offset = m.bn_offset; p.bn_per = m.bn_per; p.tanh_out = m.tanh_out;
    p.tiles = (m.H + PX_TH - 1) / PX_TH;
    const size_t smem = (size_t)PX_PIX * PX_PS * sizeof(float) + (size_t)9 * (PX_CC / 4) * 3 * sizeof(float4);
    int rc = set_smem(&m2f_px_kernel<3>, smem);
    if (rc) return rc;
    hipLaunchKernelGGL((m2f_px_kernel<3>), dim3(m.N * p.tiles), dim3(PX_NT), smem, st, p);
    ctgan_set_last_symbol("m2f_px_kernel<3>");
    return ctgan_check_launch("fewch_m2f_px");
}

static int launch_m2f(const M2FParams& p, int R, int S, int JS, hipStream_t st) {
    if (m2f_px_ok(p, R, S, JS)) return launch_m2f_px(p, st);
    if (m2f_ring_ok(p, R, S)) {
        if (JS == 3) return launch_m2f_ring<3>(p, st);
        if (JS == 1) return launch_m2f_ring<1>(p, st);
    }
    const int TR = p.band + R - 1, TW = p.Q + S - 1;
    const size_t smem = (size_t)TR * TW * p.CM * sizeof(float);
    const int grid = p.total < 512 ? p.total : 512;
    int rc = 0;
    switch (tap_case(R, S, JS)) {
        case 1: rc = set_smem(&m2f_kernel<3, 3, 3>, smem); if (!rc) hipLaunchKernelGGL((m2f_kernel<3, 3, 3>), dim3(grid), dim3(NT), smem, st, p); break;
        case 2: rc = set_smem(&m2f_kernel<1, 1, 3>, smem); if (!rc) hipLaunchKernelGGL((m2f_kernel<1, 1, 3>), dim3(grid), dim3(NT), smem, st, p); break;
        case 3: rc = set_smem(&m2f_kernel<5, 5, 1>, smem); if (!rc) hipLaunchKernelGGL((m2f_kernel<5, 5, 1>), dim3(grid), dim3(NT), smem, st, p); break;
        case 4: rc = set_smem(&m2f_kernel<3, 3, 1>, smem); if (!rc) hipLaunchKernelGGL((m2f_kernel<3, 3, 1>), dim3(grid), dim3(NT), smem, st, p); break;
        default: return ctgan_fail(CTGAN_E_UNSUPPORTED, "fewch m2f: taps");
    }
    if (rc) return rc;
    ctgan_set_last_symbol("m2f_kernel<%d, %d, %d>", R, S, JS);
    return ctgan_check_launch("fewch_m2f");
}

// forward: returns 1 when handled, 0 when the caller should use the GEMM kernels, < 0 on error
int ctgan_fewch_fwd(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, const float* mask, const float* resid,
                    float* y, int relu, int relu_in, hipStream_t st) {
    if (!ctgan_fewch_handles(d)) return 0;
    if (d->C <= 4) {
        if (d->ys[1] != 1 || (d->ys[0] | d->ys[2] | d->ys[3]) % 4 || !al16(y) || !al16(resid) || !al16(bias) || !al16(mask)) return 0;
        F2MParams p;
        p.x = x; p.xs_n = d->xs[0]; p.xs_c = d->xs[1]; p.xs_h = d->xs[2]; p.xs_w = d->xs[3]; p.H = d->H; p.W = d->W;
        p.y = y; p.ys_n = d->ys[0]; p.ys_p = d->ys[2]; p.ys_q = d->ys[3]; p.P = d->P; p.Q = d->Q; p.KM = d->K;
        p.w = w; p.w_off = 0; p.ws_r = (long long)d->S * d->C * d->K; p.ws_s = (long long)d->C * d->K; p.ws_c = d->K; p.ws_k = 1;
        p.bias = bias; p.resid = resid; p.mask = mask;
        p.N = d->N; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.relu = relu; p.relu_in = relu_in;
        p.band = pick_band(d->N, d->P, 4);
        ctgan_set_last_kernel("fewch_f2m");
        const int rc = launch_f2m(p, d->R, d->S, d->C, st);
        return rc ? rc : 1;
    }
    if (resid || relu || mask) return 0;
    if (d->xs[1] != 1 || (d->xs[0] | d->xs[2] | d->xs[3]) % 4 || !al16(x)) return 0;
    M2FParams p; p.dbg = 0;
    p.x = x; p.xs_n = d->xs[0]; p.xs_h = d->xs[2]; p.xs_w = d->xs[3]; p.H = d->H; p.W = d->W; p.CM = d->C;
    p.y = y; p.ys_n = d->ys[0]; p.ys_c = d->ys[1]; p.ys_p = d->ys[2]; p.ys_q = d->ys[3]; p.P = d->P; p.Q = d->Q;
    p.w = w; p.w_off = 0; p.ws_r = (long long)d->S * d->C * d->K; p.ws_s = (long long)d->C * d->K; p.ws_c = d->K; p.ws_j = 1;
    p.bias = bias;
    p.N = d->N; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.relu_in = relu_in;
    p.band = 2;
    if ((size_t)(p.band + d->R - 1) * (d->Q + d->S - 1) * d->C * 4 > 150 * 1024) return 0;
    p.total = d->N * ((d->P + p.band - 1) / p.band);
    ctgan_set_last_kernel("fewch_m2f");
    const int rc = launch_m2f(p, d->R, d->S, d->K, st);
    return rc ? rc : 1;
}

// forward of a many -> few conv with training-mode batch norm (+ ReLU) of the input applied on load and an optional tanh of the result:
// only the one-pixel-per-lane kernel has that staging.  Returns 1 when handled, 0 when not (the caller reports UNSUPPORTED), < 0 on error.
int ctgan_fewch_fwd_bn(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu_in, const float* mean,
                       const float* rstd, const float* scale, const float* offset, int groups, int tanh_out, hipStream_t st) {
    if (!ctgan_fewch_handles(d) || d->C <= 4 || groups <= 0 || d->N % groups) return 0;
    if (d->xs[1] != 1 || (d->xs[0] | d->xs[2] | d->xs[3]) % 4 || !al16(x) || !al16(mean) || !al16(rstd) || !al16(scale) || !al16(offset)) return 0;
    M2FParams p; p.dbg = 0;
    p.x = x; p.xs_n = d->xs[0]; p.xs_h = d->xs[2]; p.xs_w = d->xs[3]; p.H = d->H; p.W = d->W; p.CM = d->C;
    p.y = y; p.ys_n = d->ys[0]; p.ys_c = d->ys[1]; p.ys_p = d->ys[2]; p.ys_q = d->ys[3]; p.P = d->P; p.Q = d->Q;
    p.w = w; p.w_off = 0; p.ws_r = (long long)d->S * d->C * d->K; p.ws_s = (long long)d->C * d->K; p.ws_c = d->K; p.ws_j = 1;
    p.bias = bias;
    p.N = d->N; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.relu_in = relu_in;
    p.band = 2; p.total = d->N * ((d->P + 1) / 2);
    p.bn_mean = mean; p.bn_rstd = rstd; p.bn_scale = scale; p.bn_offset = offset; p.bn_per = d->N / groups; p.tanh_out = tanh_out;
    if (!m2f_px_ok(p, d->R, d->S, d->K)) return 0;
    ctgan_set_last_kernel("fewch_m2f(bn)");
    const int rc = launch_m2f_px(p, st);
    return rc ? rc : 1;
}

// data gradient dx = conv^T(dy, w) (+ bias): returns 1 / 0 / < 0 like ctgan_fewch_fwd
int ctgan_fewch_dgrad(const ctgan_conv_desc* d, const float* dy, const float* w, const float* bias, float* dx, hipStream_t st) {
    if (!ctgan_fewch_handles(d) || d->stride != 1) return 0;
    const long long tapR = (long long)d->S * d->C * d->K, tapS = (long long)d->C * d->K;
    const long long rot_off = (long long)(d->R - 1) * tapR + (long long)(d->S - 1) * tapS;       // w(R-1-r, S-1-s, ., .)
    if (d->C <= 4) {
        // dy has many channels, dx few: many -> few with the rotated filter, w'(r,s,k,c) = w[R-1-r,S-1-s,c,k]
        if (d->ys[1] != 1 || (d->ys[0] | d->ys[2] | d->ys[3]) % 4 || !al16(dy)) return 0;
        M2FParams p; p.dbg = 0;
        p.x = dy; p.xs_n = d->ys[0]; p.xs_h = d->ys[2]; p.xs_w = d->ys[3]; p.H = d->P; p.W = d->Q; p.CM = d->K;
        p.y = dx; p.ys_n = d->xs[0]; p.ys_c = d->xs[1]; p.ys_p = d->xs[2]; p.ys_q = d->xs[3]; p.P = d->H; p.Q = d->W;
        p.w = w; p.w_off = rot_off; p.ws_r = -tapR; p.ws_s = -tapS; p.ws_c = 1; p.ws_j = d->K;
        p.bias = bias;
        p.N = d->N; p.pad_t = d->R - 1 - d->pad_t; p.pad_l = d->S - 1 - d->pad_l; p.relu_in = 0;
        p.band = 2;
        if ((size_t)(p.band + d->R - 1) * (p.Q + d->S - 1) * p.CM * 4 > 150 * 1024) return 0;
        p.total = d->N * ((p.P + p.band - 1) / p.band);
        ctgan_set_last_kernel("fewch_m2f(dgrad)");
        const int rc = launch_m2f(p, d->R, d->S, d->C, st);
        return rc ? rc : 1;
    }
    // dy has few channels, dx many: few -> many with w'(r,s,j,c) = w[R-1-r,S-1-s,c,j]
    if (d->xs[1] != 1 || (d->xs[0] | d->xs[2] | d->xs[3]) % 4 || !al16(dx) || !al16(bias)) return 0;
    F2MParams p;
    p.x = dy; p.xs_n = d->ys[0]; p.xs_c = d->ys[1]; p.xs_h = d->ys[2]; p.xs_w = d->ys[3]; p.H = d->P; p.W = d->Q;
    p.y = dx; p.ys_n = d->xs[0]; p.ys_p = d->xs[2]; p.ys_q = d->xs[3]; p.P = d->H; p.Q = d->W; p.KM = d->C;
    p.w = w; p.w_off = rot_off; p.ws_r = -tapR; p.ws_s = -tapS; p.ws_c = 1; p.ws_k = d->K;
    p.bias = bias; p.resid = nullptr; p.mask = nullptr;
    p.N = d->N; p.stride = 1; p.pad_t = d->R - 1 - d->pad_t; p.pad_l = d->S - 1 - d->pad_l; p.relu = 0; p.relu_in = 0;
    p.band = pick_band(d->N, p.P, 4);
    ctgan_set_last_kernel("fewch_f2m(dgrad)");
    const int rc = launch_f2m(p, d->R, d->S, d->K, st);
    return rc ? rc : 1;
}

static void fw_plan(const ctgan_conv_desc* d, int* band, int* bands, int* n_main, int* n_out) {
    const int rows = d->C <= 4 ? d->P : d->H;              // rows of the many-channel operand
    const int cm = d->C <= 4 ? d->K : d->C;
    const int unit = NT / (cm / 4);                        // one row per lane group
    *band = pick_band(d->N, rows, unit, FW_TASKS);
    *bands = (rows + *band - 1) / *band;
    *n_main = d->R * d->S * d->C * d->K;
    *n_out = *n_main + ((d->C <= 4 ? d->K : d->K) + 3) / 4 * 4;
}

size_t ctgan_fewch_wgrad_workspace(const ctgan_conv_desc* d) {
    if (!ctgan_fewch_handles(d)) return 0;
    int band, bands, n_main, n_out;
    fw_plan(d, &band, &bands, &n_main, &n_out);
    return (size_t)d->N * bands * n_out * sizeof(float);
}

int ctgan_fewch_wgrad(const ctgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                      int relu_x, hipStream_t st) {
    return ctgan_fewch_wgrad2(d, x, dy, d->N, relu_x, db ? 1 : 0, nullptr, nullptr, 0, 0, 0, dw, db, ws, ws_bytes, st);
}

// two (x, dy) pairs of the same geometry and strides summed in one launch (d->N is ignored: N0 + N1 images)
int ctgan_fewch_wgrad2(const ctgan_conv_desc* d0, const float* x, const float* dy, int N0, int relu_x, int bias0, const float* x1,
                       const float* dy1, int N1, int relu_x1, int bias1, float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t st) {
    ctgan_conv_desc dd = *d0;
    dd.N = N0 + N1;
    const ctgan_conv_desc* d = &dd;
    if (!ctgan_fewch_handles(d)) return 0;
    if (N1 > 0 && (!x1 || !dy1 || !al16(x1) || !al16(dy1))) return 0;
    const bool few_in = d->C <= 4;
    FWParams p;
    p.N0 = N0; p.bias1 = bias0; p.bias2 = bias1;
    p.many2 = few_in ? dy1 : x1; p.few2 = few_in ? x1 : dy1;
    p.relu_many2 = few_in ? 0 : relu_x1; p.relu_few2 = few_in ? relu_x1 : 0;
    if (few_in) {
        if (d->ys[1] != 1 || (d->ys[0] | d->ys[2] | d->ys[3]) % 4 || !al16(dy)) return 0;
        p.many = dy; p.ms_n = d->ys[0]; p.ms_h = d->ys[2]; p.ms_w = d->ys[3]; p.MH = d->P; p.MW = d->Q; p.CM = d->K;
        p.few = x; p.fs_n = d->xs[0]; p.fs_c = d->xs[1]; p.fs_h = d->xs[2]; p.fs_w = d->xs[3]; p.FH = d->H; p.FW = d->W;
        p.fst = d->stride;
        for (int r = 0; r < d->R; ++r) p.off_r[r] = r - d->pad_t;
        for (int s = 0; s < d->S; ++s) p.off_s[s] = s - d->pad_l;
        p.relu_many = 0; p.relu_few = relu_x;
    } else {
        if (d->xs[1] != 1 || (d->xs[0] | d->xs[2] | d->xs[3]) % 4 || !al16(x)) return 0;
        p.many = x; p.ms_n = d->xs[0]; p.ms_h = d->xs[2]; p.ms_w = d->xs[3]; p.MH = d->H; p.MW = d->W; p.CM = d->C;
        p.few = dy; p.fs_n = d->ys[0]; p.fs_c = d->ys[1]; p.fs_h = d->ys[2]; p.fs_w = d->ys[3]; p.FH = d->P; p.FW = d->Q;
        p.fst = 1;
        for (int r = 0; r < d->R; ++r) p.off_r[r] = d->pad_t - r;
        for (int s = 0; s < d->S; ++s) p.off_s[s] = d->pad_l - s;
        p.relu_many = relu_x; p.relu_few = 0;
    }
    p.few_in = few_in ? 1 : 0; p.with_bias = db ? 1 : 0; p.N = d->N;
    fw_plan(d, &p.band, &p.bands, &p.n_main, &p.n_out);
    p.total = d->N * p.bands;
    const int blocks = p.total < 256 ? p.total : 256;
    const size_t need = (size_t)blocks * p.n_out * sizeof(float);
    if (!ws || ws_bytes < need) return ctgan_fail(CTGAN_E_BADARG, "conv2d_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
    p.slab = static_cast<float*>(ws);
    const int JS = few_in ? d->C : d->K;
    const int LPP = p.CM / 4, GROUPS = NT / LPP;
    int rspan = 0, sspan = 0;
    for (int r = 0; r < d->R; ++r) for (int r2 = 0; r2 < d->R; ++r2) if (p.off_r[r] - p.off_r[r2] > rspan) rspan = p.off_r[r] - p.off_r[r2];
    for (int s = 0; s < d->S; ++s) for (int s2 = 0; s2 < d->S; ++s2) if (p.off_s[s] - p.off_s[s2] > sspan) sspan = p.off_s[s] - p.off_s[s2];
    const size_t tile_b = (size_t)((p.band - 1) * p.fst + rspan + 1) * ((p.MW - 1) * p.fst + sspan + 1) * 16;
    const size_t red_b = (size_t)(GROUPS / 2) * LPP * (d->R * d->S * JS + 1) * 16;
    const size_t smem = tile_b > red_b ? tile_b : red_b;
    if (smem > 150 * 1024) return 0;
    int rc = 0;
    const dim3 grid(blocks), blk(NT);
    const int use_mfma = 1;
    const size_t red_mfma = (size_t)(64 * 64 + 64) * sizeof(float);
    const size_t smem_m = tile_b > red_mfma ? tile_b : red_mfma;
    if (use_mfma && p.CM == 128 && (p.MW & 1) == 0 && smem_m <= 150 * 1024) {
        bool done = true;
        // two workgroups per CU: the loop is a latency-bound stream (one float4 per lane per two pixels), registers are few
        const int cap = FW_TASKS;
        int blocks_m = p.total < cap ? p.total : cap;
        if ((size_t)blocks_m * p.n_out * sizeof(float) > ws_bytes) blocks_m = blocks;
        const dim3 grid(blocks_m);
        const int blocks = blocks_m;
        switch (tap_case(d->R, d->S, JS)) {
            case 1: rc = set_smem(&fw_wgrad_mfma_kernel<3, 3, 3>, smem_m); if (!rc) hipLaunchKernelGGL((fw_wgrad_mfma_kernel<3, 3, 3>), grid, blk, smem_m, st, p); break;
            case 2: rc = set_smem(&fw_wgrad_mfma_kernel<1, 1, 3>, smem_m); if (!rc) hipLaunchKernelGGL((fw_wgrad_mfma_kernel<1, 1, 3>), grid, blk, smem_m, st, p); break;
            case 3: rc = set_smem(&fw_wgrad_mfma_kernel<5, 5, 1>, smem_m); if (!rc) hipLaunchKernelGGL((fw_wgrad_mfma_kernel<5, 5, 1>), grid, blk, smem_m, st, p); break;
            case 4: rc = set_smem(&fw_wgrad_mfma_kernel<3, 3, 1>, smem_m); if (!rc) hipLaunchKernelGGL((fw_wgrad_mfma_kernel<3, 3, 1>), grid, blk, smem_m, st, p); break;
            default: done = false;
        }
        if (done) {
            if (rc) return rc;
            rc = ctgan_check_launch("fewch_wgrad_mfma");
            if (rc) return rc;
            ctgan_set_last_kernel(few_in ? "fewch_wgrad(few_in)" : "fewch_wgrad(few_out)");
            ctgan_set_last_symbol("fw_wgrad_mfma_kernel<%d, %d, %d>", d->R, d->S, JS);
            const int nb = db ? d->K : 0;
            const int n_tot = db ? p.n_out : p.n_main;
            hipLaunchKernelGGL(fw_reduce_kernel, dim3((n_tot + 63) / 64), dim3(NT), 0, st, p.slab, p.n_out, blocks, dw, p.n_main, db, nb, n_tot);
            rc = ctgan_check_launch("fewch_reduce");
            return rc ? rc : 1;
        }
    }
    switch (tap_case(d->R, d->S, JS)) {
        case 1: rc = set_smem(&fw_wgrad_kernel<3, 3, 3>, smem); if (!rc) hipLaunchKernelGGL((fw_wgrad_kernel<3, 3, 3>), grid, blk, smem, st, p); break;
        case 2: rc = set_smem(&fw_wgrad_kernel<1, 1, 3>, smem); if (!rc) hipLaunchKernelGGL((fw_wgrad_kernel<1, 1, 3>), grid, blk, smem, st, p); break;
        case 3: rc = set_smem(&fw_wgrad_kernel<5, 5, 1>, smem); if (!rc) hipLaunchKernelGGL((fw_wgrad_kernel<5, 5, 1>), grid, blk, smem, st, p); break;
        case 4: rc = set_smem(&fw_wgrad_kernel<3, 3, 1>, smem); if (!rc) hipLaunchKernelGGL((fw_wgrad_kernel<3, 3, 1>), grid, blk, smem, st, p); break;
        default: return 0;
    }
    if (rc) return rc;
    rc = ctgan_check_launch("fewch_wgrad");
    if (rc) return rc;
    ctgan_set_last_kernel(few_in ? "fewch_wgrad(few_in)" : "fewch_wgrad(few_out)");
    ctgan_set_last_symbol("fw_wgrad_kernel<%d, %d, %d>", d->R, d->S, JS);
    const int nb = db ? d->K : 0;
    const int n_tot = db ? p.n_out : p.n_main;
    hipLaunchKernelGGL(fw_reduce_kernel, dim3((n_tot + 63) / 64), dim3(NT), 0, st, p.slab, p.n_out, blocks, dw, p.n_main, db, nb, n_tot);
    rc = ctgan_check_launch("fewch_reduce");
    return rc ? rc : 1;
}
