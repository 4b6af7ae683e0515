// skinny.hip - kernels for the GEMM shapes where a 32x32 MFMA tile would be >90 % padding.
//
// * small-N linear (the critic heads: [n,128]x[128,1], [n,128]x[128,10], DCGAN [n,8192]x[8192,1]):
//   forward = one wave per row with a shuffle reduction, data gradient = one thread per input
//   element, weight/bias gradient = one thread per weight walking the rows in order (deterministic).
//   These are latency-bound (a few KB); the point is a ~3 us launch instead of a ~30 us GEMM tile.
#include <stdint.h>

#include "common.h"

namespace {

constexpr int MAXN = 16;

__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int rows, int C, int K, long long xs_n, long long xs_c,
                                                               long long ys_n, long long ys_k, int relu) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float acc[MAXN];
#pragma unroll
    for (int j = 0; j < MAXN; ++j) acc[j] = 0.f;
    const float* xr = x + (long long)row * xs_n;
    for (int c = lane; c < C; c += 64) {
        const float xv = xr[(long long)c * xs_c];
        const float* wr = w + (long long)c * K;
#pragma unroll
        for (int j = 0; j < MAXN; ++j)
            if (j < K) acc[j] = fmaf(xv, wr[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
        if (j >= K) break;
        float v = acc[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) {
            v += bias ? bias[j] : 0.f;
            if (relu) v = fmaxf(v, 0.f);
            y[(long long)row * ys_n + j * ys_k] = v;
        }
    }
}

// Long single-output rows (the DCGAN critic head [n,8192] x [8192,1], TF/CT_gan_cifar.py:98): one WORKGROUP per row, every thread
// keeps C/1024 independent 16-B loads of x and w in flight (the wave-per-row kernel above walks 128 dependent steps: 57 us for
// 6 MB).  Fixed-order reduction (lane tree, then the 4 wave partials in order).
__global__ __launch_bounds__(256) void linear_gemv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int C, long long xs_n, long long ys_n, int relu) {
    __shared__ float red[4];
    const float* xr = x + (long long)blockIdx.x * xs_n;
    float acc = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const float4 a = *reinterpret_cast<const float4*>(xr + c);
        const float4 b = *reinterpret_cast<const float4*>(w + c);
        acc = fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, acc))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = ((red[0] + red[1]) + (red[2] + red[3])) + (bias ? bias[0] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        y[(long long)blockIdx.x * ys_n] = v;
    }
}

// gx[n,c] = sum_j gy[n,j] * w[c,j]
__global__ void linear_small_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                          const float* __restrict__ bias, float* __restrict__ gx, int rows, int C, int K,
                                          long long gs_n, long long gs_k, long long xs_n, long long xs_c) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)rows * C) return;
    const int c = t % C;
    const long long n = t / C;
    const float* g = gy + n * gs_n;
    const float* wr = w + (long long)c * K;
    float s = bias ? bias[c] : 0.f;
    for (int j = 0; j < K; ++j) s = fmaf(g[j * gs_k], wr[j], s);
    gx[n * xs_n + (long long)c * xs_c] = s;
}

// gw[c,j] = sum_n x[n,c] * gy[n,j] ; gb[j] = sum_n gy[n,j]
// workgroup = 64 channels x 16 row lanes for one output column j (grid.y = K [+1 for the bias]);
// row lanes are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void linear_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                float* __restrict__ gw, float* __restrict__ gb, int rows,
                                                                int C, int K, long long xs_n, long long xs_c, long long gs_n,
                                                                long long gs_k) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int j = blockIdx.y;
    const bool is_bias = j == K;
    float s = 0.f;
    if (is_bias) {
        if (blockIdx.x == 0 && cl < K)
            for (int n = rl; n < rows; n += 16) s += gy[n * gs_n + cl * gs_k];
    } else if (c < C) {
        for (int n = rl; n < rows; n += 16) s = fmaf(x[n * xs_n + (long long)c * xs_c], gy[n * gs_n + j * gs_k], s);
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl != 0) return;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    if (is_bias) { if (blockIdx.x == 0 && cl < K) gb[cl] = t; }
    else if (c < C) gw[(long long)c * K + j] = t;
}

}  // namespace

bool ctgan_is_small_linear(const ctgan_conv_desc* d) {
    return d->R == 1 && d->S == 1 && d->H == 1 && d->W == 1 && d->P == 1 && d->Q == 1 && d->stride == 1 && !d->x_up &&
           d->K <= MAXN && d->N <= 65536;
}

int ctgan_small_linear_fwd(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu,
                           hipStream_t st) {
    if (d->K == 1 && d->xs[1] == 1 && d->C % 4 == 0 && d->C >= 1024 && d->xs[0] % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        hipLaunchKernelGGL(linear_gemv_fwd_kernel, dim3(d->N), dim3(256), 0, st, x, w, bias, y, d->C, (long long)d->xs[0], (long long)d->ys[0], relu);
        return ctgan_check_launch("linear_gemv_fwd");
    }
    hipLaunchKernelGGL(linear_small_fwd_kernel, dim3((d->N + 3) / 4), dim3(256), 0, st, x, w, bias, y, d->N, d->C, d->K,
                       (long long)d->xs[0], (long long)d->xs[1], (long long)d->ys[0], (long long)d->ys[1], relu);
    return ctgan_check_launch("linear_small_fwd");
}

int ctgan_small_linear_dgrad(const ctgan_conv_desc* d, const float* gy, const float* w, const float* bias, float* gx,
                             hipStream_t st) {
    const long long t = (long long)d->N * d->C;
    hipLaunchKernelGGL(linear_small_dgrad_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, st, gy, w, bias, gx, d->N,
                       d->C, d->K, (long long)d->ys[0], (long long)d->ys[1], (long long)d->xs[0], (long long)d->xs[1]);
    return ctgan_check_launch("linear_small_dgrad");
}

int ctgan_small_linear_wgrad(const ctgan_conv_desc* d, const float* x, const float* gy, float* gw, float* gb, hipStream_t st) {
    hipLaunchKernelGGL(linear_small_wgrad_kernel, dim3((d->C + 63) / 64, d->K + (gb ? 1 : 0)), dim3(1024), 0, st, x, gy, gw, gb, d->N,
                       d->C, d->K, (long long)d->xs[0], (long long)d->xs[1], (long long)d->ys[0], (long long)d->ys[1]);
    return ctgan_check_launch("linear_small_wgrad");
}

// ------------------------------------------------------------------------------------------
// im2col / col2im for convolutions with very few input channels (the critics' first conv on the
// 3- or 1-channel image): a 3x3x3 conv has a 27-deep GEMM K axis that the vector
// loaders cannot use (channel runs of 3 floats).  Expanding the image once into a channels-last
// [N,P,Q,Cpad] patch tensor (27 -> 32 columns, 16.7 MB at n=128) turns the conv, its weight gradient
// and its data gradient into 1x1 convs that run on the pipelined MFMA kernels; the expansion is
// reused by forward and weight-gradient.
namespace {

struct ColGeom {
    int N, C, H, W, R, S, stride, pad_t, pad_l, P, Q, Cpad;
    long long xs_n, xs_c, xs_h, xs_w;
};

__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, ColGeom g, long long total) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int RSC = g.R * g.S * g.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int col = i % g.Cpad;
        long long t = i / g.Cpad;
        const int q = t % g.Q; t /= g.Q;
        const int p = t % g.P; const int n = t / g.P;
        float v = 0.f;
        if (col < RSC) {
            const int tap = col / g.C, c = col - tap * g.C, r = tap / g.S, s = tap - r * g.S;
            const int ih = p * g.stride - g.pad_t + r, iw = q * g.stride - g.pad_l + s;
            if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) v = x[n * g.xs_n + c * g.xs_c + ih * g.xs_h + iw * g.xs_w];
        }
        cols[i] = v;
    }
}

// dx[n,c,h,w] = sum over taps (r,s) with p*stride - pad_t + r == h (q likewise) of cols[n,p,q,(r*S+s)*C+c]
__global__ void col2im_kernel(const float* __restrict__ cols, float* __restrict__ dx, ColGeom g, long long total) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        // iterate (n,h,w,c) with c fastest
        const int c = i % g.C;
        long long t = i / g.C;
        const int w = t % g.W; t /= g.W;
        const int h = t % g.H; const int n = t / g.H;
        float acc = 0.f;
        for (int r = 0; r < g.R; ++r) {
            const int ph = h + g.pad_t - r;
            if (ph < 0 || ph % g.stride) continue;
            const int p = ph / g.stride;
            if (p >= g.P) continue;
            for (int s = 0; s < g.S; ++s) {
                const int qw = w + g.pad_l - s;
                if (qw < 0 || qw % g.stride) continue;
                const int q = qw / g.stride;
                if (q >= g.Q) continue;
                acc += cols[(((long long)n * g.P + p) * g.Q + q) * g.Cpad + (r * g.S + s) * g.C + c];
            }
        }
        dx[n * g.xs_n + c * g.xs_c + h * g.xs_h + w * g.xs_w] = acc;
    }
}

// im2col, band form (Cpad % 4 == 0): one workgroup = `band` output rows of one image.  The input rows the band reads sit in LDS as
// [row][iw][c] with the zero padding materialised, so the S*C columns of a filter row are ONE contiguous LDS run starting at the pixel's
// first tap; a thread owns one float4 of the Cpad columns (its four LDS offsets are loop invariants) and walks the band's pixels:
// 16-byte stores, whole 128-byte lines per pixel, no divisions in the loop.  (The per-element kernel above: 19.6 us for config[1]'s
// 256 x 16 x 16 x 96 patch tensor, 1.3 TB/s.)
__global__ __launch_bounds__(256) void im2col_band_kernel(const float* __restrict__ x, float* __restrict__ cols, ColGeom g, int band) {
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [TR][TW * C]
    const int tid = threadIdx.x;
    const int bands = (g.P + band - 1) / band;
    const int n = blockIdx.x / bands, b = blockIdx.x - n * bands;
    const int p0 = b * band, np = min(band, g.P - p0);
    const int TR = (band - 1) * g.stride + g.R, TW = (g.Q - 1) * g.stride + g.S, TWC = TW * g.C;
    const int ih0 = p0 * g.stride - g.pad_t, iw0 = -g.pad_l;
    for (int i = tid; i < g.C * TR * TW; i += 256) {
        const int c = i / (TR * TW), rem = i - c * TR * TW, tr = rem / TW, tw = rem - tr * TW;
        const int ih = ih0 + tr, iw = iw0 + tw;
        float v = 0.f;
        if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) v = x[n * g.xs_n + c * g.xs_c + ih * g.xs_h + iw * g.xs_w];
        tile[tr * TWC + tw * g.C + c] = v;
    }
    const int cq = g.Cpad >> 2, ppw = 256 / cq;
    const int c4 = tid % cq, pl = tid / cq;
    const int SC = g.S * g.C, RSC = g.R * SC;
    int o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = c4 * 4 + k, r = col / SC;
        o[k] = col < RSC ? r * TWC + (col - r * SC) : -1;
    }
    __syncthreads();
    if (pl >= ppw) return;
    const int npx = np * g.Q;
    int pr = 0, q = pl;
    while (q >= g.Q) { q -= g.Q; ++pr; }
    float* out = cols + ((long long)n * g.P + p0) * g.Q * g.Cpad + c4 * 4;
    for (int px = pl; px < npx; px += ppw) {
        const float* base = tile + pr * g.stride * TWC + q * g.stride * g.C;
        float4 v;
        v.x = o[0] >= 0 ? base[o[0]] : 0.f; v.y = o[1] >= 0 ? base[o[1]] : 0.f;
        v.z = o[2] >= 0 ? base[o[2]] : 0.f; v.w = o[3] >= 0 ? base[o[3]] : 0.f;
        *reinterpret_cast<float4*>(out + (long long)px * g.Cpad) = v;
        q += ppw;
        while (q >= g.Q) { q -= g.Q; ++pr; }
    }
}

// col2im, one thread per input pixel (all C <= 4 channels): only the taps whose output position exists are visited - r runs over the
// residue class of (h + pad_t) mod stride - and a tap's C columns are consecutive floats.  (The per-element kernel above visits all R*S
// taps with two modulo tests each: 14.6 us for 64 rows of config[1].)
template <int C>
__global__ __launch_bounds__(256) void col2im_px_kernel(const float* __restrict__ cols, float* __restrict__ dx, ColGeom g, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % g.W);
    const long long t = i / g.W;
    const int h = (int)(t % g.H), n = (int)(t / g.H);
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    const int hh = h + g.pad_t, ww = w + g.pad_l;
    for (int r = hh % g.stride; r < g.R; r += g.stride) {                 // ascending r, s: the summation order of the per-element kernel
        const int p = (hh - r) / g.stride;
        if (hh - r < 0 || p >= g.P) continue;
        for (int s = ww % g.stride; s < g.S; s += g.stride) {
            const int q = (ww - s) / g.stride;
            if (ww - s < 0 || q >= g.Q) continue;
            const float* src = cols + (((long long)n * g.P + p) * g.Q + q) * g.Cpad + (r * g.S + s) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += src[c];
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dx[n * g.xs_n + c * g.xs_c + h * g.xs_h + w * g.xs_w] = acc[c];
}

ColGeom col_geom(const ctgan_conv_desc* d, int cpad) {
    ColGeom g;
    g.N = d->N; g.C = d->C; g.H = d->H; g.W = d->W; g.R = d->R; g.S = d->S; g.stride = d->stride;
    g.pad_t = d->pad_t; g.pad_l = d->pad_l; g.P = d->P; g.Q = d->Q; g.Cpad = cpad;
    g.xs_n = d->xs[0]; g.xs_c = d->xs[1]; g.xs_h = d->xs[2]; g.xs_w = d->xs[3];
    return g;
}

}  // namespace

extern "C" {

int ctgan_im2col(const ctgan_conv_desc* d, const float* x, int32_t cpad, float* cols, ctgan_stream_t stream) {
    if (!d || !x || !cols || cpad < d->R * d->S * d->C || d->x_up) return ctgan_fail(CTGAN_E_BADARG, "im2col: bad argument");
    const long long total = (long long)d->N * d->P * d->Q * cpad;
    if (cpad % 4 == 0 && cpad <= 1024 && (reinterpret_cast<uintptr_t>(cols) & 15) == 0) {
        int band = 1;                                        // the most rows per workgroup that still give >= 1024 workgroups
        for (int b = 8; b > 1; b >>= 1)
            if ((long long)d->N * ((d->P + b - 1) / b) >= 1024 && b <= d->P) { band = b; break; }
        const size_t smem = (size_t)d->C * ((band - 1) * d->stride + d->R) * ((d->Q - 1) * d->stride + d->S) * sizeof(float);
        if (smem <= 48 * 1024) {
            hipLaunchKernelGGL(im2col_band_kernel, dim3((unsigned)(d->N * ((d->P + band - 1) / band))), dim3(256), smem,
                               static_cast<hipStream_t>(stream), x, cols, col_geom(d, cpad), band);
            return ctgan_check_launch("im2col");
        }
    }
    hipLaunchKernelGGL(im2col_kernel, dim3(ctgan_blocks(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, cols,
                       col_geom(d, cpad), total);
    return ctgan_check_launch("im2col");
}

int ctgan_col2im(const ctgan_conv_desc* d, const float* cols, int32_t cpad, float* dx, ctgan_stream_t stream) {
    if (!d || !dx || !cols || cpad < d->R * d->S * d->C || d->x_up) return ctgan_fail(CTGAN_E_BADARG, "col2im: bad argument");
    const long long total = (long long)d->N * d->C * d->H * d->W;
    if (d->C <= 4) {
        const long long px = (long long)d->N * d->H * d->W;
        const dim3 grid((unsigned)((px + 255) / 256)), blk(256);
        hipStream_t st = static_cast<hipStream_t>(stream);
        const ColGeom g = col_geom(d, cpad);
        switch (d->C) {
            case 1: hipLaunchKernelGGL(col2im_px_kernel<1>, grid, blk, 0, st, cols, dx, g, px); break;
            case 2: hipLaunchKernelGGL(col2im_px_kernel<2>, grid, blk, 0, st, cols, dx, g, px); break;
            case 3: hipLaunchKernelGGL(col2im_px_kernel<3>, grid, blk, 0, st, cols, dx, g, px); break;
            default: hipLaunchKernelGGL(col2im_px_kernel<4>, grid, blk, 0, st, cols, dx, g, px); break;
        }
        return ctgan_check_launch("col2im");
    }
    hipLaunchKernelGGL(col2im_kernel, dim3(ctgan_blocks(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), cols, dx,
                       col_geom(d, cpad), total);
    return ctgan_check_launch("col2im");
}

}  // extern "C"
