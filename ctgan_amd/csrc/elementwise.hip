// elementwise.hip - HBM-bound helpers of the CT-WGAN step.
// All are grid-stride kernels; contiguous fp32 streams use 16-byte accesses when aligned.
#include "common.h"

namespace {

constexpr int TPB = 256;

template <typename F>
__global__ void map1_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, F f) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = n >> 2;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
        for (long long v = i; v < n4; v += stride) {
            float4 a = reinterpret_cast<const float4*>(x)[v];
            a.x = f(a.x); a.y = f(a.y); a.z = f(a.z); a.w = f(a.w);
            reinterpret_cast<float4*>(y)[v] = a;
        }
        for (long long t = (n4 << 2) + i; t < n; t += stride) y[t] = f(x[t]);
    } else {
        for (; i < n; i += stride) y[i] = f(x[i]);
    }
}

template <typename F>
__global__ void map2_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ y,
                            long long n, F f) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = n >> 2;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z)) & 15) == 0) {
        for (long long v = i; v < n4; v += stride) {
            float4 a = reinterpret_cast<const float4*>(x)[v];
            const float4 b = reinterpret_cast<const float4*>(z)[v];
            a.x = f(a.x, b.x); a.y = f(a.y, b.y); a.z = f(a.z, b.z); a.w = f(a.w, b.w);
            reinterpret_cast<float4*>(y)[v] = a;
        }
        for (long long t = (n4 << 2) + i; t < n; t += stride) y[t] = f(x[t], z[t]);
    } else {
        for (; i < n; i += stride) y[i] = f(x[i], z[i]);
    }
}

template <typename F>
int launch1(const float* x, float* y, long long n, F f, ctgan_stream_t s, const char* who) {
    if (n < 0 || (n > 0 && (!x || !y))) return ctgan_fail(CTGAN_E_BADARG, "%s: bad argument", who);
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(map1_kernel<F>, dim3(ctgan_blocks((n + 3) / 4, TPB, 2048)), dim3(TPB), 0,
                       static_cast<hipStream_t>(s), x, y, n, f);
    return ctgan_check_launch(who);
}
template <typename F>
int launch2(const float* x, const float* z, float* y, long long n, F f, ctgan_stream_t s, const char* who) {
    if (n < 0 || (n > 0 && (!x || !y || !z))) return ctgan_fail(CTGAN_E_BADARG, "%s: bad argument", who);
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(map2_kernel<F>, dim3(ctgan_blocks((n + 3) / 4, TPB, 2048)), dim3(TPB), 0,
                       static_cast<hipStream_t>(s), x, z, y, n, f);
    return ctgan_check_launch(who);
}

struct LReluF { float a; __device__ float operator()(float x) const { return x > 0.f ? x : a * x; } };
struct LReluB { float a; __device__ float operator()(float g, float r) const { return r > 0.f ? g : a * g; } };
struct LReluBS { float a, sc; __device__ float operator()(float g, float r) const { return (r > 0.f ? g : a * g) * sc; } };
struct MulF { __device__ float operator()(float x, float y) const { return x * y; } };
struct RsqrtF { float eps; __device__ float operator()(float x) const { return 1.f / sqrtf(x + eps); } };
struct DropF { float keep, inv; __device__ float operator()(float x, float u) const { return x * inv * floorf(keep + u); } };
struct TanhF { __device__ float operator()(float x) const { return tanhf(x); } };
struct TanhB { __device__ float operator()(float g, float y) const { return g * (1.f - y * y); } };
struct SigF { __device__ float operator()(float x) const { return 1.f / (1.f + expf(-x)); } };
struct SigB { __device__ float operator()(float g, float y) const { return g * y * (1.f - y); } };
struct ScaleF { float a; __device__ float operator()(float x) const { return a * x; } };
struct AxpbyF { float a, b; __device__ float operator()(float x, float y) const { return a * x + b * y; } };

struct Dims4 { int d[4]; long long xs[4], ys[4]; };

__global__ void copy4d_kernel(const float* __restrict__ x, float* __restrict__ y, Dims4 p, long long n) {
    // iterate in the OUTPUT's fastest-varying order so writes coalesce: order[] sorts dims by ys
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long t = i;
        const int i3 = t % p.d[3]; t /= p.d[3];
        const int i2 = t % p.d[2]; t /= p.d[2];
        const int i1 = t % p.d[1]; const int i0 = t / p.d[1];
        y[i0 * p.ys[0] + i1 * p.ys[1] + i2 * p.ys[2] + i3 * p.ys[3]] =
            x[i0 * p.xs[0] + i1 * p.xs[1] + i2 * p.xs[2] + i3 * p.xs[3]];
    }
}

// y[n,c,p,q] = scale * (x[n,c,2p,2q] + x[n,c,2p+1,2q] + x[n,c,2p,2q+1] + x[n,c,2p+1,2q+1])
// iteration order (n,p,q,c): c fastest, matching channels-last tensors
__global__ void pool2_kernel(const float* __restrict__ x, float* __restrict__ y, Dims4 p, float scale, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int N = p.d[0], C = p.d[1], P = p.d[2], Q = p.d[3];
    (void)N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long t = i;
        const int c = t % C; t /= C;
        const int q = t % Q; t /= Q;
        const int pp = t % P; const int nn = t / P;
        const float* b = x + nn * p.xs[0] + c * p.xs[1] + (2 * pp) * p.xs[2] + (2 * q) * p.xs[3];
        const float v = (b[0] + b[p.xs[2]]) + (b[p.xs[3]] + b[p.xs[2] + p.xs[3]]);
        y[nn * p.ys[0] + c * p.ys[1] + pp * p.ys[2] + q * p.ys[3]] = scale * v;
    }
}

__global__ void upsample2_kernel(const float* __restrict__ x, float* __restrict__ y, Dims4 p, float scale, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int C = p.d[1], H = p.d[2], W = p.d[3];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long t = i;
        const int c = t % C; t /= C;
        const int w = t % W; t /= W;
        const int h = t % H; const int nn = t / H;
        y[nn * p.ys[0] + c * p.ys[1] + h * p.ys[2] + w * p.ys[3]] =
            scale * x[nn * p.xs[0] + c * p.xs[1] + (h >> 1) * p.xs[2] + (w >> 1) * p.xs[3]];
    }
}

// x [n, hw, c] channels-last -> y[n,c] = scale * sum_hw.  Workgroup = 64 channels x 4 position lanes of one
// sample (coalesced 256-B rows, 4 independent accumulation chains), lanes combined through LDS in fixed order.
__global__ __launch_bounds__(256) void spatial_sum_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int hw, int c,
                                                         float scale) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int cc = blockIdx.y * 64 + cl, nn = blockIdx.x;
    float s = 0.f;
    if (cc < c) {
        const float* b = x + (long long)nn * hw * c + cc;
        for (int k = pl; k < hw; k += 4) s += b[(long long)k * c];
    }
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && cc < c) y[(long long)nn * c + cc] = scale * ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

__global__ void spatial_bcast_kernel(const float* __restrict__ g, float* __restrict__ y, int n, int hw, int c, float scale) {
    const long long total = (long long)n * hw * c;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cc = i % c; const long long nn = i / ((long long)hw * c);
        y[i] = scale * g[nn * c + cc];
    }
}

__global__ void real_prep_kernel(const int32_t* __restrict__ xi, const float* __restrict__ noise, float* __restrict__ y,
                                 long long n, float denom) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = 2.f * (((float)xi[i] / denom) - .5f);
        if (noise) v += noise[i];
        y[i] = v;
    }
}

__global__ void interpolate_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                   const float* __restrict__ alpha, float* __restrict__ out, int b, int d) {
    const long long total = (long long)b * d;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float r = real[i];
        out[i] = r + alpha[i / d] * (fake[i] - r);
    }
}

// column sums, two fixed-order stages.  Stage 1: a workgroup = RLANES row lanes x 64 columns walks a
// contiguous row chunk (each wave load = 256 contiguous bytes), lanes are combined through LDS.
// Stage 2: the same shape over the partial rows.
constexpr int CS_COLS = 64, CS_LANES = 4;
__global__ __launch_bounds__(CS_COLS * CS_LANES) void colsum_stage_kernel(const float* __restrict__ x, long long rows, int cols,
                                                                         long long ld, float* __restrict__ out,
                                                                         int rows_per_block) {
    __shared__ float red[CS_LANES][CS_COLS];
    const int cl = threadIdx.x % CS_COLS, rl = threadIdx.x / CS_COLS;
    const int j = blockIdx.y * CS_COLS + cl;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (j < cols) {
        long long r = r0 + rl;
        for (; r + 7 * CS_LANES < r1; r += 8 * CS_LANES) {          // 8 independent loads in flight, summed in the same fixed order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(r + u * CS_LANES) * ld + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < r1; r += CS_LANES) s += x[r * ld + j];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && j < cols) out[(long long)blockIdx.x * cols + j] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

int colsum_plan(long long rows, int* rows_per_block) {
    int nblk = (int)((rows + 255) / 256);
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    *rows_per_block = (int)((rows + nblk - 1) / nblk);
    return (int)((rows + *rows_per_block - 1) / *rows_per_block);
}

Dims4 mk(const int32_t dims[4], const int64_t xs[4], const int64_t ys[4]) {
    Dims4 p;
    for (int i = 0; i < 4; ++i) { p.d[i] = dims[i]; p.xs[i] = xs[i]; p.ys[i] = ys[i]; }
    return p;
}

// Filter of a conv fused with a 2x resampling (see ctgan_filter_spread in the header).
//   spread: out[u,v,c,k] = scale * sum_{a,b in {0,1}} w[u-a, v-b, c, k]        (out is (R+1) x (S+1))
//   flip  : the result is written as out[R-u, S-v, k, c] (rotated, I/O swapped: the transposed-conv filter)
__global__ void filter_spread_kernel(const float* __restrict__ w, float* __restrict__ out, int R, int S, int C, int K,
                                     float scale, int flip) {
    const long long n = (long long)(R + 1) * (S + 1) * C * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K); long long r = i / K;
        const int c = (int)(r % C); r /= C;
        const int v = (int)(r % (S + 1)), u = (int)(r / (S + 1));
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int rr = u - a, ss = v - b;
                if (rr >= 0 && rr < R && ss >= 0 && ss < S) acc += w[(((long long)rr * S + ss) * C + c) * K + k];
            }
        const long long o = flip ? ((((long long)(R - u) * (S + 1) + (S - v)) * K + k) * C + c) : i;
        out[o] = scale * acc;
    }
}
// fold (the adjoint): out[r,s,c,k] = scale * sum_{a,b} W4[r+a, s+b, c, k];  flip: W4[u,v,c,k] = w4[R-u, S-v, k, c]
__global__ void filter_fold_kernel(const float* __restrict__ w4, float* __restrict__ out, int R, int S, int C, int K,
                                   float scale, int flip) {
    const long long n = (long long)R * S * C * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K); long long r = i / K;
        const int c = (int)(r % C); r /= C;
        const int ss = (int)(r % S), rr = (int)(r / S);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int u = rr + a, v = ss + b;
                acc += flip ? w4[(((long long)(R - u) * (S + 1) + (S - v)) * K + k) * C + c]
                            : w4[(((long long)u * (S + 1) + v) * C + c) * K + k];
            }
        out[i] = scale * acc;
    }
}

// several folds in one launch (blockIdx.y = job): the spread-filter gradients of a step are folded after its queued weight
// gradients have been flushed
struct FoldJobs { ctgan_fold_job j[CTGAN_FOLD_BATCH]; };
__global__ void filter_fold_batch_kernel(const FoldJobs t) {
    const ctgan_fold_job& jb = t.j[blockIdx.y];
    const int R = jb.R, S = jb.S, C = jb.C, K = jb.K, flip = jb.flip;
    const float* __restrict__ w4 = jb.src;
    float* __restrict__ out = jb.dst;
    const long long n = (long long)R * S * C * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K); long long r = i / K;
        const int c = (int)(r % C); r /= C;
        const int ss = (int)(r % S), rr = (int)(r / S);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int u = rr + a, v = ss + b;
                acc += flip ? w4[(((long long)(R - u) * (S + 1) + (S - v)) * K + k) * C + c]
                            : w4[(((long long)u * (S + 1) + v) * C + c) * K + k];
            }
        out[i] = jb.scale * acc;
    }
}

// All derived filters of a weight update in one launch (blockIdx.y = job): the rotated / phase-major layouts the
// data-gradient kernels multiply with, the spread filters of the resampled convs, and the data-gradient layouts OF the
// spread filters straight from the parameter (job.pre) - so nothing in the launch depends on another job's output.
// Every output tap plane is a C x K matrix that is a (scaled sum of up to four) source tap plane(s), or the transpose of it:
// one workgroup moves one 32 x 32 tile with coalesced reads (k inner) and coalesced writes (through LDS when transposed).
struct FilterJobs { ctgan_filter_job j[CTGAN_FILTER_BATCH]; };
__global__ __launch_bounds__(256) void filter_batch_kernel(const FilterJobs t) {
    const ctgan_filter_job& jb = t.j[blockIdx.y];
    const int R = jb.R, S = jb.S, C = jb.C, K = jb.K;            // the SOURCE parameter w[R,S,C,K]
    const int pre = jb.pre ? jb.pre : ((jb.kind == CTGAN_FILTER_SPREAD || jb.kind == CTGAN_FILTER_SPREAD_FLIP) ? jb.kind : 0);
    const int kind = (jb.kind == CTGAN_FILTER_SPREAD || jb.kind == CTGAN_FILTER_SPREAD_FLIP) ? -1 : jb.kind;   // -1: materialise E itself
    const float sc = jb.pre ? jb.pre_scale : ((kind == -1) ? jb.scale : 1.f);
    const int Re = pre ? R + 1 : R, Se = pre ? S + 1 : S;        // taps of the effective filter E
    const int tc = (C + 31) >> 5, tk = (K + 31) >> 5, tiles = tc * tk;
    int planes;
    const int Tr = (Re + 1) / 2, Ts = (Se + 1) / 2;
    if (kind == CTGAN_FILTER_PHASES) planes = 4 * Tr * Ts; else planes = Re * Se;
    const int plane = blockIdx.x / tiles;
    if (plane >= planes) return;
    const int tile = blockIdx.x - plane * tiles;
    const int c0 = (tile / tk) * 32, k0 = (tile % tk) * 32;
    // effective tap (u, v) of E this output plane holds, -1 = zero plane
    int u, v;
    if (kind == CTGAN_FILTER_PHASES) {
        const int ph = plane / (Tr * Ts), r = plane - ph * Tr * Ts, tt = r / Ts, vv = r - tt * Ts;
        const int a = ph >> 1, bb = ph & 1;
        u = ((a + jb.pad_t) & 1) + 2 * (Tr - 1 - tt); v = ((bb + jb.pad_l) & 1) + 2 * (Ts - 1 - vv);
        if (u >= Re || v >= Se) u = -1;
    } else if (kind == CTGAN_FILTER_ROTATE) {
        u = Re - 1 - plane / Se; v = Se - 1 - plane % Se;
    } else {
        u = plane / Se; v = plane % Se;
    }
    // E of a flipped spread is the rotated, I/O-swapped spread: tap (u,v) of E = transpose of spread tap (R-u, S-v)
    bool transpose = (kind != -1);                                // the dgrad layouts are [k][c] of E
    if (pre == CTGAN_FILTER_SPREAD_FLIP) { transpose = !transpose; if (u >= 0) { u = R - u; v = S - v; } }
    if (((C | K) & 3) == 0) {                                     // 16-byte path: one float4 per thread each way
        const int row = threadIdx.x >> 3, q4 = threadIdx.x & 7;   // 32 rows x 8 float4
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = c0 + row, k = k0 + q4 * 4;
        if (u >= 0 && c < C && k < K) {
            if (pre) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int rr = u - a, ss = v - bb;
                        if (rr >= 0 && rr < R && ss >= 0 && ss < S) {
                            const float4 w4 = *reinterpret_cast<const float4*>(jb.src + (((long long)rr * S + ss) * C + c) * K + k);
                            acc.x += w4.x; acc.y += w4.y; acc.z += w4.z; acc.w += w4.w;
                        }
                    }
                acc.x = sc * acc.x; acc.y = sc * acc.y; acc.z = sc * acc.z; acc.w = sc * acc.w;
            } else {
                acc = *reinterpret_cast<const float4*>(jb.src + (((long long)u * S + v) * C + c) * K + k);
            }
        }
        float* out4 = jb.dst + (long long)plane * C * K;
        if (!transpose) {
            if (c < C && k < K) *reinterpret_cast<float4*>(out4 + (long long)c * K + k) = acc;
        } else {
            __shared__ float tl4[32][33];
            tl4[row][q4 * 4 + 0] = acc.x; tl4[row][q4 * 4 + 1] = acc.y; tl4[row][q4 * 4 + 2] = acc.z; tl4[row][q4 * 4 + 3] = acc.w;
            __syncthreads();
            const int kk = k0 + row, cc = c0 + q4 * 4;            // now: row = k within the tile, q4 = float4 of c
            if (kk < K && cc < C)
                *reinterpret_cast<float4*>(out4 + (long long)kk * C + cc) =
                    make_float4(tl4[q4 * 4 + 0][row], tl4[q4 * 4 + 1][row], tl4[q4 * 4 + 2][row], tl4[q4 * 4 + 3][row]);
        }
        return;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    float val[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, k = k0 + tx;
        float acc = 0.f;
        if (u >= 0 && c < C && k < K) {
            if (pre) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int rr = u - a, ss = v - bb;
                        if (rr >= 0 && rr < R && ss >= 0 && ss < S) acc += jb.src[(((long long)rr * S + ss) * C + c) * K + k];
                    }
                acc = sc * acc;
            } else {
                acc = jb.src[(((long long)u * S + v) * C + c) * K + k];
            }
        }
        val[i] = acc;
    }
    float* out = jb.dst + (long long)plane * C * K;
    if (!transpose) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, k = k0 + tx;
            if (c < C && k < K) out[(long long)c * K + k] = val[i];
        }
    } else {
        __shared__ float tl[32][33];
#pragma unroll
        for (int i = 0; i < 4; ++i) tl[ty + 8 * i][tx] = val[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, c = c0 + tx;
            if (c < C && k < K) out[(long long)k * C + c] = tl[tx][ty + 8 * i];
        }
    }
}

// ---- per-sample / per-channel primitives of Layernorm (TF/tflib/ops/layernorm.py:6-20) ---------------------------
// x dense [n][m] (any inner layout): y[i] = scale * sum_j x[i][j]   (one workgroup per sample, fixed-order tree)
__global__ __launch_bounds__(256) void sample_sum_kernel(const float* __restrict__ x, float* __restrict__ y, long long m, float scale) {
    __shared__ float sh[4];
    const float* row = x + (long long)blockIdx.x * m;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long j = threadIdx.x;
    for (; j + 768 < m; j += 1024) { s0 += row[j]; s1 += row[j + 256]; s2 += row[j + 512]; s3 += row[j + 768]; }
    for (; j < m; j += 256) s0 += row[j];
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) y[blockIdx.x] = scale * ((sh[0] + sh[1]) + (sh[2] + sh[3]));
}
// y[i][j] = scale * v[i]
__global__ void sample_bcast_kernel(const float* __restrict__ v, float* __restrict__ y, long long m, long long total, float scale) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) y[i] = scale * v[i / m];
}
// channels-last x [rows][c]: y = x * s[c] + o[c]   (o may be NULL)
__global__ void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ of,
                                      float* __restrict__ y, int c, long long total) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % c);
        y[i] = x[i] * sc[ch] + (of ? of[ch] : 0.f);
    }
}

}  // namespace

extern "C" {

int ctgan_lrelu_fwd(const float* x, float* y, int64_t n, float alpha, ctgan_stream_t s) {
    return launch1(x, y, n, LReluF{alpha}, s, "lrelu_fwd");
}
int ctgan_lrelu_bwd(const float* gy, const float* ref, float* gx, int64_t n, float alpha, ctgan_stream_t s) {
    return launch2(gy, ref, gx, n, LReluB{alpha}, s, "lrelu_bwd");
}
int ctgan_lrelu_bwd_scaled(const float* gy, const float* ref, float* gx, int64_t n, float alpha, float scale, ctgan_stream_t s) {
    return launch2(gy, ref, gx, n, LReluBS{alpha, scale}, s, "lrelu_bwd_scaled");
}
int ctgan_dropout(const float* x, const float* u, float* y, int64_t n, float keep, ctgan_stream_t s) {
    if (!(keep > 0.f) || keep > 1.f) return ctgan_fail(CTGAN_E_BADARG, "dropout: keep=%g not in (0,1]", keep);
    return launch2(x, u, y, n, DropF{keep, 1.f / keep}, s, "dropout");
}
int ctgan_tanh_fwd(const float* x, float* y, int64_t n, ctgan_stream_t s) { return launch1(x, y, n, TanhF{}, s, "tanh_fwd"); }
int ctgan_tanh_bwd(const float* gy, const float* y, float* gx, int64_t n, ctgan_stream_t s) {
    return launch2(gy, y, gx, n, TanhB{}, s, "tanh_bwd");
}
int ctgan_sigmoid_fwd(const float* x, float* y, int64_t n, ctgan_stream_t s) { return launch1(x, y, n, SigF{}, s, "sigmoid_fwd"); }
int ctgan_sigmoid_bwd(const float* gy, const float* y, float* gx, int64_t n, ctgan_stream_t s) {
    return launch2(gy, y, gx, n, SigB{}, s, "sigmoid_bwd");
}
int ctgan_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, ctgan_stream_t s) {
    if (!y) return launch1(x, out, n, ScaleF{a}, s, "axpby");
    return launch2(x, y, out, n, AxpbyF{a, b}, s, "axpby");
}

int ctgan_copy4d(const float* x, const int64_t xs[4], float* y, const int64_t ys[4], const int32_t dims[4],
                 ctgan_stream_t s) {
    if (!x || !y || !xs || !ys || !dims) return ctgan_fail(CTGAN_E_BADARG, "copy4d: null");
    // walk in the order that makes the output's unit-stride dim fastest
    int order[4] = {0, 1, 2, 3};
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b)
            if (ys[order[b]] > ys[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    int32_t d2[4]; int64_t xs2[4], ys2[4];
    long long n = 1;
    for (int i = 0; i < 4; ++i) { d2[i] = dims[order[i]]; xs2[i] = xs[order[i]]; ys2[i] = ys[order[i]]; n *= dims[i]; }
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(copy4d_kernel, dim3(ctgan_blocks(n, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), x, y,
                       mk(d2, xs2, ys2), n);
    return ctgan_check_launch("copy4d");
}

int ctgan_pool2(const float* x, const int64_t xs[4], float* y, const int64_t ys[4], const int32_t ydims[4], float scale,
                ctgan_stream_t s) {
    if (!x || !y) return ctgan_fail(CTGAN_E_BADARG, "pool2: null");
    long long n = 1;
    for (int i = 0; i < 4; ++i) n *= ydims[i];
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(pool2_kernel, dim3(ctgan_blocks(n, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), x, y,
                       mk(ydims, xs, ys), scale, n);
    return ctgan_check_launch("pool2");
}

int ctgan_upsample2(const float* x, const int64_t xs[4], float* y, const int64_t ys[4], const int32_t ydims[4],
                    float scale, ctgan_stream_t s) {
    if (!x || !y) return ctgan_fail(CTGAN_E_BADARG, "upsample2: null");
    if (ydims[2] % 2 || ydims[3] % 2) return ctgan_fail(CTGAN_E_BADARG, "upsample2: odd output size");
    long long n = 1;
    for (int i = 0; i < 4; ++i) n *= ydims[i];
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(upsample2_kernel, dim3(ctgan_blocks(n, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), x, y,
                       mk(ydims, xs, ys), scale, n);
    return ctgan_check_launch("upsample2");
}

int ctgan_filter_spread(const float* w, float* out, int32_t R, int32_t S, int32_t C, int32_t K, float scale, int32_t flip,
                        ctgan_stream_t s) {
    if (!w || !out || R <= 0 || S <= 0 || C <= 0 || K <= 0) return ctgan_fail(CTGAN_E_BADARG, "filter_spread: bad argument");
    const long long n = (long long)(R + 1) * (S + 1) * C * K;
    hipLaunchKernelGGL(filter_spread_kernel, dim3(ctgan_blocks(n, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), w, out, R, S,
                       C, K, scale, flip);
    return ctgan_check_launch("filter_spread");
}
int ctgan_filter_fold(const float* w4, float* out, int32_t R, int32_t S, int32_t C, int32_t K, float scale, int32_t flip,
                      ctgan_stream_t s) {
    if (!w4 || !out || R <= 0 || S <= 0 || C <= 0 || K <= 0) return ctgan_fail(CTGAN_E_BADARG, "filter_fold: bad argument");
    const long long n = (long long)R * S * C * K;
    hipLaunchKernelGGL(filter_fold_kernel, dim3(ctgan_blocks(n, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), w4, out, R, S, C,
                       K, scale, flip);
    return ctgan_check_launch("filter_fold");
}

int ctgan_filter_fold_batch(const ctgan_fold_job* jobs, int32_t n, ctgan_stream_t s) {
    if (!jobs || n < 0) return ctgan_fail(CTGAN_E_BADARG, "filter_fold_batch: bad argument");
    for (int32_t base = 0; base < n; base += CTGAN_FOLD_BATCH) {
        FoldJobs t;
        const int m = n - base < CTGAN_FOLD_BATCH ? n - base : CTGAN_FOLD_BATCH;
        long long nmax = 0;
        for (int i = 0; i < m; ++i) {
            const ctgan_fold_job& j = t.j[i] = jobs[base + i];
            if (!j.src || !j.dst || j.R <= 0 || j.S <= 0 || j.C <= 0 || j.K <= 0) return ctgan_fail(CTGAN_E_BADARG, "filter_fold_batch: bad job %d", base + i);
            const long long e = (long long)j.R * j.S * j.C * j.K;
            if (e > nmax) nmax = e;
        }
        hipLaunchKernelGGL(filter_fold_batch_kernel, dim3(ctgan_blocks(nmax, TPB, 1024), m), dim3(TPB), 0, static_cast<hipStream_t>(s), t);
        const int rc = ctgan_check_launch("filter_fold_batch");
        if (rc) return rc;
    }
    return CTGAN_OK;
}

int ctgan_filter_batch(const ctgan_filter_job* jobs, int32_t n, ctgan_stream_t s) {
    if (!jobs || n < 0) return ctgan_fail(CTGAN_E_BADARG, "filter_batch: bad argument");
    for (int32_t base = 0; base < n; base += CTGAN_FILTER_BATCH) {
        FilterJobs t;
        const int m = n - base < CTGAN_FILTER_BATCH ? n - base : CTGAN_FILTER_BATCH;
        int blocks = 0;
        for (int i = 0; i < m; ++i) {
            const ctgan_filter_job& j = t.j[i] = jobs[base + i];
            const bool spread = j.kind == CTGAN_FILTER_SPREAD || j.kind == CTGAN_FILTER_SPREAD_FLIP;
            if (!j.src || !j.dst || j.R <= 0 || j.S <= 0 || j.C <= 0 || j.K <= 0 || j.kind < 0 || j.kind > CTGAN_FILTER_SPREAD_FLIP ||
                (j.pre != 0 && j.pre != CTGAN_FILTER_SPREAD && j.pre != CTGAN_FILTER_SPREAD_FLIP) || (j.pre && spread))
                return ctgan_fail(CTGAN_E_BADARG, "filter_batch: bad job %d", base + i);
            const int Re = (j.pre || spread) ? j.R + 1 : j.R, Se = (j.pre || spread) ? j.S + 1 : j.S;
            const int planes = j.kind == CTGAN_FILTER_PHASES ? 4 * ((Re + 1) / 2) * ((Se + 1) / 2) : Re * Se;
            const int b = planes * ((j.C + 31) / 32) * ((j.K + 31) / 32);
            if (b > blocks) blocks = b;
        }
        hipLaunchKernelGGL(filter_batch_kernel, dim3(blocks, m), dim3(256), 0, static_cast<hipStream_t>(s), t);
        const int rc = ctgan_check_launch("filter_batch");
        if (rc) return rc;
    }
    return CTGAN_OK;
}

int ctgan_mul(const float* x, const float* y, float* out, int64_t n, ctgan_stream_t s) { return launch2(x, y, out, n, MulF{}, s, "mul"); }
int ctgan_rsqrt(const float* x, float* y, int64_t n, float eps, ctgan_stream_t s) { return launch1(x, y, n, RsqrtF{eps}, s, "rsqrt"); }
int ctgan_sample_sum(const float* x, float* y, int32_t n, int64_t m, float scale, ctgan_stream_t s) {
    if (!x || !y || n <= 0 || m <= 0) return ctgan_fail(CTGAN_E_BADARG, "sample_sum: bad argument");
    hipLaunchKernelGGL(sample_sum_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(s), x, y, (long long)m, scale);
    return ctgan_check_launch("sample_sum");
}
int ctgan_sample_bcast(const float* v, float* y, int32_t n, int64_t m, float scale, ctgan_stream_t s) {
    if (!v || !y || n <= 0 || m <= 0) return ctgan_fail(CTGAN_E_BADARG, "sample_bcast: bad argument");
    const long long total = (long long)n * m;
    hipLaunchKernelGGL(sample_bcast_kernel, dim3(ctgan_blocks(total, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), v, y, (long long)m,
                       total, scale);
    return ctgan_check_launch("sample_bcast");
}
int ctgan_channel_affine(const float* x, const float* scale, const float* offset, float* y, int64_t rows, int32_t c, ctgan_stream_t s) {
    if (!x || !scale || !y || rows <= 0 || c <= 0) return ctgan_fail(CTGAN_E_BADARG, "channel_affine: bad argument");
    const long long total = (long long)rows * c;
    hipLaunchKernelGGL(channel_affine_kernel, dim3(ctgan_blocks(total, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), x, scale, offset, y,
                       c, total);
    return ctgan_check_launch("channel_affine");
}

int ctgan_spatial_sum(const float* x, float* y, int32_t n, int32_t hw, int32_t c, float scale, ctgan_stream_t s) {
    if (!x || !y || n <= 0 || hw <= 0 || c <= 0) return ctgan_fail(CTGAN_E_BADARG, "spatial_sum: bad argument");
    hipLaunchKernelGGL(spatial_sum_kernel, dim3(n, (c + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(s), x, y, n, hw, c,
                       scale);
    return ctgan_check_launch("spatial_sum");
}
int ctgan_spatial_bcast(const float* g, float* y, int32_t n, int32_t hw, int32_t c, float scale, ctgan_stream_t s) {
    if (!g || !y || n <= 0 || hw <= 0 || c <= 0) return ctgan_fail(CTGAN_E_BADARG, "spatial_bcast: bad argument");
    hipLaunchKernelGGL(spatial_bcast_kernel, dim3(ctgan_blocks((long long)n * hw * c, TPB)), dim3(TPB), 0,
                       static_cast<hipStream_t>(s), g, y, n, hw, c, scale);
    return ctgan_check_launch("spatial_bcast");
}

int ctgan_real_prep(const int32_t* x_int, const float* noise, float* y, int64_t n, float denom, ctgan_stream_t s) {
    if (!x_int || !y || n < 0 || denom == 0.f) return ctgan_fail(CTGAN_E_BADARG, "real_prep: bad argument");
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(real_prep_kernel, dim3(ctgan_blocks(n, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(s), x_int,
                       noise, y, (long long)n, denom);
    return ctgan_check_launch("real_prep");
}

int ctgan_interpolate(const float* real, const float* fake, const float* alpha, float* out, int32_t b, int32_t d,
                      ctgan_stream_t s) {
    if (!real || !fake || !alpha || !out || b <= 0 || d <= 0) return ctgan_fail(CTGAN_E_BADARG, "interpolate: bad argument");
    hipLaunchKernelGGL(interpolate_kernel, dim3(ctgan_blocks((long long)b * d, TPB)), dim3(TPB), 0,
                       static_cast<hipStream_t>(s), real, fake, alpha, out, b, d);
    return ctgan_check_launch("interpolate");
}

size_t ctgan_colsum_workspace_bytes(int64_t rows, int32_t cols) {
    int rpb;
    const int nblk = colsum_plan(rows, &rpb);
    return (size_t)nblk * cols * sizeof(float);
}

int ctgan_colsum(const float* x, int64_t rows, int32_t cols, int64_t ld, float* out, void* ws, size_t ws_bytes,
                 ctgan_stream_t s) {
    if (!x || !out || rows <= 0 || cols <= 0) return ctgan_fail(CTGAN_E_BADARG, "colsum: bad argument");
    int rpb;
    const int nblk = colsum_plan(rows, &rpb);
    if (!ws || ws_bytes < (size_t)nblk * cols * sizeof(float)) return ctgan_fail(CTGAN_E_BADARG, "colsum: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(s);
    const int cblk = (cols + CS_COLS - 1) / CS_COLS;
    hipLaunchKernelGGL(colsum_stage_kernel, dim3(nblk, cblk), dim3(CS_COLS * CS_LANES), 0, st, x, (long long)rows, cols,
                       (long long)ld, nblk > 1 ? static_cast<float*>(ws) : out, rpb);
    int rc = ctgan_check_launch("colsum_stage1");
    if (rc || nblk == 1) return rc;
    hipLaunchKernelGGL(colsum_stage_kernel, dim3(1, cblk), dim3(CS_COLS * CS_LANES), 0, st, static_cast<const float*>(ws),
                       (long long)nblk, cols, (long long)cols, out, nblk);
    return ctgan_check_launch("colsum_stage2");
}

}  // extern "C"
