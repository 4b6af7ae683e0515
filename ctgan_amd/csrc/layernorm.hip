// layernorm.hip - fused Layernorm forward / backward / double backward (ctgan_layernorm_{fwd,bwd,bwd2}).
//
// TF/tflib/ops/layernorm.py:6-20 (LSUN copy: LS/tflib/ops/layernorm.py): per-sample moments over (C,H,W) (tf.nn.moments,
// biased variance), y = (x - mean) * rsqrt(var + 1e-5) * scale[c] + offset[c].  The layer-normalised critics (config[4],
// the 64x64 GoodDiscriminator) are differentiated TWICE through it by the gradient penalty, so three maps are needed:
//   fwd : y  = xh * scale + offset [then ReLU: `relu`; the backward maps then take y as `ymask` and treat gy as gy * (y > 0)],
//                                                       xh = (x - mean) * r,  r = rsqrt(var + eps)
//   bwd : gx = r * (g - mean(g) - xh * mean(g * xh)),   g = gy * scale;  gscale[c] = sum gy * xh,  goffset[c] = sum gy
//   bwd2: the adjoint of bwd with respect to (gy, x, scale), given the cotangent u of gx.  bwd is linear and SYMMETRIC in g,
//         so  cot_g = r * (u - mean(u) - xh * mean(u * xh)),  cot_gy = cot_g * scale,  cot_scale[c] = sum gy * cot_g;
//         through xh and r:  q = -r * (b * u + m * g)   (a = mean g, b = mean g xh, m = mean u xh),
//         cot_x = r * (q - mean(q)) - xh * (r * mean(q xh) + r^2 * mean(u h)),   h = g - a - xh * b,
//         with mean(q) = -r (b mean(u) + m a), mean(q xh) = -2 r b m, mean(u h) = mean(u g) - a mean(u) - b m:
//         everything follows from FIVE per-sample means (u, u xh, g, g xh, u g).
// Every map is two passes over the tensor: per-(sample, chunk) partial sums (fp32 per thread, fp64 across threads and
// chunks, fixed order => deterministic), then the elementwise pass, whose workgroups first fold the handful of chunk partials
// of their sample.  The per-channel parameter gradients ride the elementwise pass (each thread owns a fixed group of 4
// channels: 1024 % C == 0) as per-workgroup partial rows + one small column reduction.  The composition of ~9 elementwise /
// reduction kernels this replaces (functional.layer_norm_composed) stays as the fallback for other channel counts.
// x / gy / u / outputs: dense, channel fastest ([N,H,W,C] or [N,C]); D = elements per sample.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int CHUNK = 16384;          // elements per workgroup (16 float4 per thread)

__device__ __forceinline__ double block_sum(double v, double* red /* [NT/64] */) {
    // wave reduction through DPP-free shuffles, then across the 4 waves through LDS; result valid in every thread
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// mode 0: sums of x, x^2                               -> part[n][chunk][2]
// mode 1: sums of g, g*xh               (g = gy*scale) -> part[n][chunk][2]
// mode 2: sums of u, u*xh, g, g*xh, u*g                -> part[n][chunk][5]
template <int MODE>
__global__ __launch_bounds__(NT) void ln_partial_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ u,
                                                        const float* __restrict__ scale, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ ymask, long long D, int C,
                                                        double* __restrict__ part) {
    constexpr int NS = MODE == 2 ? 5 : 2;
    __shared__ double red[NT / 64];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const long long e0 = (long long)chunk * CHUNK, e1 = min(D, e0 + CHUNK);
    const float* xs = x + (long long)n * D;
    const float* gs = MODE ? gy + (long long)n * D : nullptr;
    const float* us = MODE == 2 ? u + (long long)n * D : nullptr;
    const float mu = MODE ? mean[n] : 0.f, r = MODE ? rstd[n] : 0.f;
    float s[NS];
    double d0 = 0., d1 = 0.;          // mode 0: E[x^2] - E[x]^2 cancels when |mean| >> std - carry the moments in fp64 throughout
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = 0.f;
    for (long long e = e0 + threadIdx.x * 4; e < e1; e += NT * 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + e);
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { d0 += (double)xa[j]; d1 += (double)xa[j] * (double)xa[j]; }
        } else {
            const float4 gv = *reinterpret_cast<const float4*>(gs + e);
            const float4 sv = *reinterpret_cast<const float4*>(scale + (int)(e % C));
            float ga[4] = {gv.x * sv.x, gv.y * sv.y, gv.z * sv.z, gv.w * sv.w};
            if (ymask) {                      // fused ReLU: the gradient only passes where the forward result is positive
                const float4 yv = *reinterpret_cast<const float4*>(ymask + (long long)n * D + e);
                ga[0] = yv.x > 0.f ? ga[0] : 0.f; ga[1] = yv.y > 0.f ? ga[1] : 0.f; ga[2] = yv.z > 0.f ? ga[2] : 0.f; ga[3] = yv.w > 0.f ? ga[3] : 0.f;
            }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float xh = (xa[j] - mu) * r; s[0] += ga[j]; s[1] += ga[j] * xh; }
            } else {
                const float4 uv = *reinterpret_cast<const float4*>(us + e);
                const float ua[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xa[j] - mu) * r;
                    s[0] += ua[j]; s[1] += ua[j] * xh; s[2] += ga[j]; s[3] += ga[j] * xh; s[4] += ua[j] * ga[j];
                }
            }
        }
    }
    double* o = part + ((long long)n * gridDim.x + chunk) * NS;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double t = block_sum(MODE == 0 ? (k == 0 ? d0 : d1) : (double)s[k], red);
        if (threadIdx.x == 0) o[k] = t;
    }
}

template <int NS>
__device__ __forceinline__ void fold_partials(const double* __restrict__ part, int n, int chunks, double (&tot)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) tot[k] = 0.;
    const double* p = part + (long long)n * chunks * NS;
    for (int c = 0; c < chunks; ++c)
#pragma unroll
        for (int k = 0; k < NS; ++k) tot[k] += p[c * NS + k];
}

__global__ __launch_bounds__(NT) void ln_fwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
                                                          const double* __restrict__ part, long long D, int C, float eps, int relu,
                                                          float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    double t[2];
    fold_partials<2>(part, n, gridDim.x, t);
    const double m = t[0] / (double)D;
    double var = t[1] / (double)D - m * m;
    if (var < 0.) var = 0.;
    const float mu = (float)m, r = (float)(1.0 / sqrt(var + (double)eps));
    if (chunk == 0 && threadIdx.x == 0) { mean[n] = mu; rstd[n] = r; }
    const long long e0 = (long long)chunk * CHUNK, e1 = min(D, e0 + CHUNK);
    const float* xs = x + (long long)n * D;
    float* ys = y + (long long)n * D;
    for (long long e = e0 + threadIdx.x * 4; e < e1; e += NT * 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + e);
        const int c = (int)(e % C);
        const float4 sv = *reinterpret_cast<const float4*>(scale + c);
        const float4 ov = *reinterpret_cast<const float4*>(offset + c);
        float4 o;
        o.x = (xv.x - mu) * r * sv.x + ov.x; o.y = (xv.y - mu) * r * sv.y + ov.y;
        o.z = (xv.z - mu) * r * sv.z + ov.z; o.w = (xv.w - mu) * r * sv.w + ov.w;
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(ys + e) = o;
    }
}

// per-workgroup, per-channel partial sums: every thread owns the 4 channels (e % C) of all the float4s it visits
// (NT*4 % C == 0); threads that own the same channels are NT*4/C apart ... combine through LDS, one row per workgroup.
template <int NV>
__device__ __forceinline__ void channel_rows(const float (&acc)[NV][4], int C, float* lds /* [NV][NT*4] */, float* __restrict__ rows,
                                             long long row, int nrows_stride) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[v * NT * 4 + threadIdx.x * 4 + j] = acc[v][j];
    __syncthreads();
    // element i of the NT*4-wide image has channel i % C
    for (int c = threadIdx.x; c < C; c += NT) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float s = 0.f;
            for (int i = c; i < NT * 4; i += C) s += lds[v * NT * 4 + i];
            rows[((long long)v * nrows_stride + row) * C + c] = s;
        }
    }
}

__global__ __launch_bounds__(NT) void ln_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ ymask,
                                                          const double* __restrict__ part, long long D, int C, float* __restrict__ gx,
                                                          float* __restrict__ rows /* [2][N*chunks][C] or null */) {
    __shared__ float lds[2 * NT * 4];
    const int n = blockIdx.y, chunk = blockIdx.x;
    double t[2];
    fold_partials<2>(part, n, gridDim.x, t);
    const float a = (float)(t[0] / (double)D), b = (float)(t[1] / (double)D);
    const float mu = mean[n], r = rstd[n];
    const long long e0 = (long long)chunk * CHUNK, e1 = min(D, e0 + CHUNK);
    const float* xs = x + (long long)n * D;
    const float* gs = gy + (long long)n * D;
    float* os = gx + (long long)n * D;
    float acc[2][4] = {};
    for (long long e = e0 + threadIdx.x * 4; e < e1; e += NT * 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + e);
        const float4 gv = *reinterpret_cast<const float4*>(gs + e);
        const float4 sv = *reinterpret_cast<const float4*>(scale + (int)(e % C));
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, sa[4] = {sv.x, sv.y, sv.z, sv.w};
        float gya[4] = {gv.x, gv.y, gv.z, gv.w};
        if (ymask) {
            const float4 yv = *reinterpret_cast<const float4*>(ymask + (long long)n * D + e);
            gya[0] = yv.x > 0.f ? gya[0] : 0.f; gya[1] = yv.y > 0.f ? gya[1] : 0.f; gya[2] = yv.z > 0.f ? gya[2] : 0.f; gya[3] = yv.w > 0.f ? gya[3] : 0.f;
        }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xa[j] - mu) * r;
            o[j] = r * (gya[j] * sa[j] - a - xh * b);
            acc[0][j] += gya[j] * xh; acc[1][j] += gya[j];
        }
        *reinterpret_cast<float4*>(os + e) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (rows) channel_rows<2>(acc, C, lds, rows, (long long)n * gridDim.x + chunk, gridDim.x * gridDim.y);
}

__global__ __launch_bounds__(NT) void ln_bwd2_apply_kernel(const float* __restrict__ u, const float* __restrict__ gy, const float* __restrict__ x,
                                                           const float* __restrict__ scale, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ ymask,
                                                           const double* __restrict__ part, long long D, int C,
                                                           float* __restrict__ cot_gy, float* __restrict__ cot_x,
                                                           float* __restrict__ rows /* [1][N*chunks][C] or null */) {
    __shared__ float lds[NT * 4];
    const int n = blockIdx.y, chunk = blockIdx.x;
    double t[5];
    fold_partials<5>(part, n, gridDim.x, t);
    const double inv = 1.0 / (double)D;
    const float mu_u = (float)(t[0] * inv), m = (float)(t[1] * inv), a = (float)(t[2] * inv), b = (float)(t[3] * inv), ug = (float)(t[4] * inv);
    const float mu = mean[n], r = rstd[n];
    const float mean_q = -r * (b * mu_u + m * a);
    const float k_xh = r * (-2.f * r * b * m) + r * r * (ug - a * mu_u - b * m);     // r*mean(q xh) + r^2*mean(u h)
    const long long e0 = (long long)chunk * CHUNK, e1 = min(D, e0 + CHUNK);
    const float* xs = x + (long long)n * D;
    const float* gs = gy + (long long)n * D;
    const float* us = u + (long long)n * D;
    float acc[1][4] = {};
    for (long long e = e0 + threadIdx.x * 4; e < e1; e += NT * 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + e);
        const float4 gv = *reinterpret_cast<const float4*>(gs + e);
        const float4 uv = *reinterpret_cast<const float4*>(us + e);
        const float4 sv = *reinterpret_cast<const float4*>(scale + (int)(e % C));
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ua[4] = {uv.x, uv.y, uv.z, uv.w}, sa[4] = {sv.x, sv.y, sv.z, sv.w};
        float gya[4] = {gv.x, gv.y, gv.z, gv.w}, mk[4] = {1.f, 1.f, 1.f, 1.f};
        if (ymask) {
            const float4 yv = *reinterpret_cast<const float4*>(ymask + (long long)n * D + e);
            mk[0] = yv.x > 0.f ? 1.f : 0.f; mk[1] = yv.y > 0.f ? 1.f : 0.f; mk[2] = yv.z > 0.f ? 1.f : 0.f; mk[3] = yv.w > 0.f ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) gya[j] *= mk[j];
        }
        float og[4], ox[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xa[j] - mu) * r;
            const float g = gya[j] * sa[j];
            const float cg = r * (ua[j] - mu_u - xh * m);
            og[j] = cg * sa[j] * mk[j];
            acc[0][j] += gya[j] * cg;
            const float q = -r * (b * ua[j] + m * g);
            ox[j] = r * (q - mean_q) - xh * k_xh;
        }
        if (cot_gy) *reinterpret_cast<float4*>(cot_gy + (long long)n * D + e) = make_float4(og[0], og[1], og[2], og[3]);
        if (cot_x) *reinterpret_cast<float4*>(cot_x + (long long)n * D + e) = make_float4(ox[0], ox[1], ox[2], ox[3]);
    }
    if (rows) channel_rows<1>(acc, C, lds, rows, (long long)n * gridDim.x + chunk, gridDim.x * gridDim.y);
}

// out[v][c] = sum over rows of rows[v][row][c]: a workgroup owns 64 channels x 16 row lanes (256-B coalesced row segments, four
// independent loads in flight per lane), fixed order => deterministic
constexpr int RC = 64, RLN = 16;
__global__ __launch_bounds__(RC * RLN) void ln_rows_reduce_kernel(const float* __restrict__ rows, long long nrows, int C, float* __restrict__ out0,
                                                                  float* __restrict__ out1) {
    __shared__ double red[RLN][RC];
    const int cl = threadIdx.x % RC, rl = threadIdx.x / RC;
    const int c = blockIdx.x * RC + cl, v = blockIdx.y;
    double s = 0.;
    if (c < C) {
        const float* base = rows + (long long)v * nrows * C + c;
        long long r = rl;
        for (; r + 3 * RLN < nrows; r += 4 * RLN) {
            const float v0 = base[r * C], v1 = base[(r + RLN) * C], v2 = base[(r + 2 * RLN) * C], v3 = base[(r + 3 * RLN) * C];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; r < nrows; r += RLN) s += (double)base[r * C];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.;
#pragma unroll
        for (int k = 0; k < RLN; ++k) t += red[k][cl];
        (v == 0 ? out0 : out1)[c] = (float)t;
    }
}

int chunks_of(long long D) { return (int)((D + CHUNK - 1) / CHUNK); }

bool ln_ok(long long D, int C) { return C > 0 && C % 4 == 0 && (NT * 4) % C == 0 && D % C == 0; }

}  // namespace

extern "C" {

int ctgan_layernorm_supported(int64_t D, int32_t C) { return ln_ok(D, C) ? 1 : 0; }

size_t ctgan_layernorm_workspace_bytes(int32_t N, int64_t D, int32_t C) {
    const size_t ch = (size_t)chunks_of(D);
    return (size_t)N * ch * 5 * sizeof(double) + (size_t)2 * N * ch * C * sizeof(float);
}

int ctgan_layernorm_fwd(const float* x, const float* scale, const float* offset, float* y, float* mean, float* rstd, int32_t N,
                        int64_t D, int32_t C, float eps, int32_t relu, void* ws, size_t ws_bytes, ctgan_stream_t stream) {
    if (!x || !scale || !offset || !y || !mean || !rstd || N <= 0) return ctgan_fail(CTGAN_E_BADARG, "layernorm_fwd: bad argument");
    if (!ln_ok(D, C)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "layernorm_fwd: D=%lld C=%d outside the fused kernels", (long long)D, C);
    if (ws_bytes < ctgan_layernorm_workspace_bytes(N, D, C)) return ctgan_fail(CTGAN_E_BADARG, "layernorm_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(chunks_of(D), N);
    double* part = (double*)ws;
    hipLaunchKernelGGL(ln_partial_kernel<0>, grid, dim3(NT), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (long long)D, C, part);
    hipLaunchKernelGGL(ln_fwd_apply_kernel, grid, dim3(NT), 0, st, x, scale, offset, part, (long long)D, C, eps, (int)relu, y, mean, rstd);
    return ctgan_check_launch("layernorm_fwd");
}

int ctgan_layernorm_bwd(const float* gy, const float* x, const float* scale, const float* mean, const float* rstd, const float* ymask,
                        float* gx, float* gscale, float* goffset, int32_t N, int64_t D, int32_t C, void* ws, size_t ws_bytes,
                        ctgan_stream_t stream) {
    if (!gy || !x || !scale || !mean || !rstd || !gx || N <= 0 || (!gscale) != (!goffset)) return ctgan_fail(CTGAN_E_BADARG, "layernorm_bwd: bad argument");
    if (!ln_ok(D, C)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "layernorm_bwd: D=%lld C=%d outside the fused kernels", (long long)D, C);
    if (ws_bytes < ctgan_layernorm_workspace_bytes(N, D, C)) return ctgan_fail(CTGAN_E_BADARG, "layernorm_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int ch = chunks_of(D);
    const dim3 grid(ch, N);
    double* part = (double*)ws;
    float* rows = gscale ? (float*)(part + (size_t)N * ch * 5) : nullptr;
    hipLaunchKernelGGL(ln_partial_kernel<1>, grid, dim3(NT), 0, st, x, gy, nullptr, scale, mean, rstd, ymask, (long long)D, C, part);
    hipLaunchKernelGGL(ln_bwd_apply_kernel, grid, dim3(NT), 0, st, gy, x, scale, mean, rstd, ymask, part, (long long)D, C, gx, rows);
    if (rows) hipLaunchKernelGGL(ln_rows_reduce_kernel, dim3((C + RC - 1) / RC, 2), dim3(RC * RLN), 0, st, rows, (long long)N * ch, C, gscale, goffset);
    return ctgan_check_launch("layernorm_bwd");
}

int ctgan_layernorm_bwd2(const float* u, const float* gy, const float* x, const float* scale, const float* mean, const float* rstd,
                         const float* ymask, float* cot_gy, float* cot_x, float* cot_scale, int32_t N, int64_t D, int32_t C, void* ws,
                         size_t ws_bytes, ctgan_stream_t stream) {
    if (!u || !gy || !x || !scale || !mean || !rstd || N <= 0) return ctgan_fail(CTGAN_E_BADARG, "layernorm_bwd2: bad argument");
    if (!ln_ok(D, C)) return ctgan_fail(CTGAN_E_UNSUPPORTED, "layernorm_bwd2: D=%lld C=%d outside the fused kernels", (long long)D, C);
    if (ws_bytes < ctgan_layernorm_workspace_bytes(N, D, C)) return ctgan_fail(CTGAN_E_BADARG, "layernorm_bwd2: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int ch = chunks_of(D);
    const dim3 grid(ch, N);
    double* part = (double*)ws;
    float* rows = cot_scale ? (float*)(part + (size_t)N * ch * 5) : nullptr;
    hipLaunchKernelGGL(ln_partial_kernel<2>, grid, dim3(NT), 0, st, x, gy, u, scale, mean, rstd, ymask, (long long)D, C, part);
    hipLaunchKernelGGL(ln_bwd2_apply_kernel, grid, dim3(NT), 0, st, u, gy, x, scale, mean, rstd, ymask, part, (long long)D, C, cot_gy, cot_x, rows);
    if (rows) hipLaunchKernelGGL(ln_rows_reduce_kernel, dim3((C + RC - 1) / RC, 1), dim3(RC * RLN), 0, st, rows, (long long)N * ch, C, cot_scale, cot_scale);
    return ctgan_check_launch("layernorm_bwd2");
}

}  // extern "C"
