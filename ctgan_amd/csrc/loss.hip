// loss.hip - fused loss heads of the CT-WGAN critic/generator objectives.
// Tiny tensors (B=64 rows): one workgroup per sample for the row reductions, one workgroup for the
// batch mean; fixed-order tree reductions (deterministic).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// blockDim = 256; returns the block sum in every thread
__device__ __forceinline__ float block_sum(float v, float* sh /*[4]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// slopes[b] = ||g[b,:]||
__global__ __launch_bounds__(256) void gp_slopes_kernel(const float* __restrict__ g, int d, float* __restrict__ slopes) {
    __shared__ float sh[4];
    const float* row = g + (long long)blockIdx.x * d;
    float s = 0.f;
    for (int i = threadIdx.x; i < d; i += 256) { const float v = row[i]; s += v * v; }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) slopes[blockIdx.x] = sqrtf(s);
}
__global__ __launch_bounds__(256) void gp_mean_kernel(const float* __restrict__ slopes, int b, float lambda, float* __restrict__ gp) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < b; i += 256) { const float t = slopes[i] - 1.f; s += t * t; }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) gp[0] = lambda * s / (float)b;
}
__global__ void gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ slopes, const float* __restrict__ gout,
                              int b, int d, float lambda, float* __restrict__ gg) {
    const long long total = (long long)b * d;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float go = gout[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float s = slopes[i / d];
        // d/dg [ lambda/B * (s-1)^2 ] = lambda/B * 2 (s-1) * g/s   (s==0 => 0, as TF's sqrt-grad would give nan; guard)
        const float coef = s > 0.f ? go * lambda * 2.f * (s - 1.f) / (s * (float)b) : 0.f;
        gg[i] = coef * g[i];
    }
}

// CT_i = l2*(d-d_)^2 + 0.1*l2*mean_j (f-f_)^2
__global__ __launch_bounds__(256) void ct_rows_kernel(const float* __restrict__ d, const float* __restrict__ d_,
                                                      const float* __restrict__ f, const float* __restrict__ f_, int nf,
                                                      float l2, float* __restrict__ ct_i) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < nf; j += 256) { const float t = f[(long long)i * nf + j] - f_[(long long)i * nf + j]; s += t * t; }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) { const float t = d[i] - d_[i]; ct_i[i] = l2 * t * t + l2 * 0.1f * (s / (float)nf); }
}
__global__ __launch_bounds__(256) void ct_mean_kernel(const float* __restrict__ ct_i, int b, float M, float* __restrict__ ct) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < b; i += 256) s += fmaxf(ct_i[i] - M, 0.f);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) ct[0] = s / (float)b;
}
__global__ void ct_bwd_kernel(const float* __restrict__ d, const float* __restrict__ d_, const float* __restrict__ f,
                              const float* __restrict__ f_, const float* __restrict__ ct_i, const float* __restrict__ gout,
                              int b, int nf, float l2, float M, float* __restrict__ gd, float* __restrict__ gd_,
                              float* __restrict__ gf, float* __restrict__ gf_) {
    const long long total = (long long)b * (nf + 1);
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float go = gout[0] / (float)b;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int i = t / (nf + 1), j = t - (long long)i * (nf + 1);
        // tf.maximum(a, 0*a): gradient flows to the first argument where a >= 0*a ... TF's
        // MaximumGrad routes the gradient to x where x >= y; at CT_i - M == 0 both args are equal
        // and x gets it.  For M=0 and CT_i>=0 that is always "on".
        const float on = (ct_i[i] - M >= 0.f) ? go : 0.f;
        if (j == nf) {
            const float v = on * l2 * 2.f * (d[i] - d_[i]);
            gd[i] = v; gd_[i] = -v;
        } else {
            const long long o = (long long)i * nf + j;
            const float v = on * l2 * 0.1f * 2.f * (f[o] - f_[o]) / (float)nf;
            gf[o] = v; gf_[o] = -v;
        }
    }
}

// one thread per row (b <= few hundred, ncls = 10)
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                             int b, int ncls, float* __restrict__ probs,
                                                             float* __restrict__ loss, float* __restrict__ ncorrect) {
    __shared__ float sh[4];
    float l = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < b; i += 256) {
        const float* z = logits + (long long)i * ncls;
        float mx = z[0]; int am = 0;
        for (int k = 1; k < ncls; ++k) if (z[k] > mx) { mx = z[k]; am = k; }
        float se = 0.f;
        for (int k = 0; k < ncls; ++k) se += expf(z[k] - mx);
        const float lse = logf(se);
        for (int k = 0; k < ncls; ++k) probs[(long long)i * ncls + k] = expf(z[k] - mx - lse);
        const int lab = labels[i];
        l += (mx + lse) - z[lab];
        c += (am == lab) ? 1.f : 0.f;
    }
    l = block_sum(l, sh);
    c = block_sum(c, sh);
    if (threadIdx.x == 0) { loss[0] = l / (float)b; if (ncorrect) ncorrect[0] = c; }
}
__global__ void softmax_ce_bwd_kernel(const float* __restrict__ probs, const int32_t* __restrict__ labels,
                                      const float* __restrict__ gout, int b, int ncls, float* __restrict__ gl) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * ncls) return;
    const int i = t / ncls, k = t - i * ncls;
    gl[t] = gout[0] / (float)b * (probs[t] - (labels[i] == k ? 1.f : 0.f));
}

__global__ __launch_bounds__(256) void mean_diff_fwd_kernel(const float* __restrict__ x, int na, int nb, float sa, float sb,
                                                            float* __restrict__ out) {
    __shared__ float sh[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < na; i += 256) a += x[i];
    for (int i = threadIdx.x; i < nb; i += 256) b += x[na + i];
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0) out[0] = (na ? sa * a / (float)na : 0.f) + (nb ? sb * b / (float)nb : 0.f);
}
__global__ void mean_diff_bwd_kernel(const float* __restrict__ gout, int na, int nb, float sa, float sb, float* __restrict__ gx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na + nb) return;
    gx[i] = gout[0] * (i < na ? sa / (float)na : sb / (float)nb);
}


// ---- all critic loss heads of one D step in one launch (forward) / one launch (backward) ------------------
// d [3B], f [3B,nf], a [3B,ncls] are the outputs of the batched dropout passes: rows [0,B) real pass 1,
// [B,2B) fake pass 1, [2B,3B) real pass 2.
//   wgan  = mean(d[B:2B]) - mean(d[0:B])                                               (:244)
//   CT_i  = l2*(d_i - d_{2B+i})^2 + 0.1*l2*mean_j (f_ij - f_{2B+i,j})^2;  ct = mean_i max(CT_i - M, 0)   (:288-291)
//   acgan = mean_i softmax-CE(a[i], labels[i]),  i < B                                  (:246-248)
//   out = {wgan + ct + scale*acgan, wgan, ct, acgan}
__global__ __launch_bounds__(256) void critic_heads_fwd_kernel(const float* __restrict__ d, const float* __restrict__ f,
                                                               const float* __restrict__ a, const int32_t* __restrict__ labels,
                                                               const float* __restrict__ gp, int B, int nf, int ncls, float l2,
                                                               float M, float scale, float* __restrict__ ct_i,
                                                               float* __restrict__ probs, float* __restrict__ out) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = w; i < B; i += 4) {                       // one wave per row
        float s = 0.f;
        for (int j = lane; j < nf; j += 64) { const float t = f[(long long)i * nf + j] - f[(long long)(2 * B + i) * nf + j]; s += t * t; }
        s = wave_sum(s);
        if (lane == 0) { const float t = d[i] - d[2 * B + i]; ct_i[i] = l2 * t * t + l2 * 0.1f * (s / (float)nf); }
    }
    __syncthreads();
    float sr = 0.f, sf = 0.f, sc = 0.f, sl = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        sr += d[i]; sf += d[B + i];
        sc += fmaxf(ct_i[i] - M, 0.f);
        if (a) {
            const float* z = a + (long long)i * ncls;
            float mx = z[0];
            for (int k = 1; k < ncls; ++k) mx = fmaxf(mx, z[k]);
            float se = 0.f;
            for (int k = 0; k < ncls; ++k) se += expf(z[k] - mx);
            const float lse = logf(se);
            for (int k = 0; k < ncls; ++k) probs[(long long)i * ncls + k] = expf(z[k] - mx - lse);
            sl += (mx + lse) - z[labels[i]];
        }
    }
    sr = block_sum(sr, sh); sf = block_sum(sf, sh); sc = block_sum(sc, sh); sl = block_sum(sl, sh);
    if (threadIdx.x == 0) {
        const float wgan = sf / (float)B - sr / (float)B, ct = sc / (float)B, ac = a ? sl / (float)B : 0.f;
        const float pen = gp ? gp[0] : 0.f;                    // gradient penalty of the step (:284-286), computed elsewhere
        out[0] = ((wgan + ct) + pen) + scale * ac; out[1] = wgan; out[2] = ct; out[3] = ac; out[4] = (wgan + ct) + pen;
    }
}
// gradients w.r.t. d [3B], f [3B,nf], a [3B,ncls] in one pass (every element written, zeros included);
// gout[4] = upstream gradients of {cost, wgan, ct, acgan}
__global__ void critic_heads_bwd_kernel(const float* __restrict__ d, const float* __restrict__ f, const float* __restrict__ probs,
                                        const int32_t* __restrict__ labels, const float* __restrict__ ct_i,
                                        const float* __restrict__ gout, int n_gout, int B, int nf, int ncls, float l2, float M,
                                        float scale, float* __restrict__ gd, float* __restrict__ gf, float* __restrict__ ga) {
    const float g0 = gout[0], g1 = n_gout > 1 ? gout[1] : 0.f, g2 = n_gout > 1 ? gout[2] : 0.f, g3 = n_gout > 1 ? gout[3] : 0.f;
    const float cw = (g0 + g1) / (float)B, cc = (g0 + g2) / (float)B, ca = (g0 * scale + g3) / (float)B;
    const long long n_f = 3LL * B * nf, n_a = ga ? 3LL * B * ncls : 0, total = n_f + n_a + 3LL * B;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        if (t < n_f) {
            const int row = (int)(t / nf), j = (int)(t - (long long)row * nf);
            float v = 0.f;
            if (row < B || row >= 2 * B) {
                const int i = row < B ? row : row - 2 * B;
                // tf.maximum(CT_i - M, 0): the gradient goes to the first argument where it is >= the second (TF MaximumGrad)
                const float on = (ct_i[i] - M >= 0.f) ? cc : 0.f;
                v = on * l2 * 0.1f * 2.f * (f[(long long)i * nf + j] - f[(long long)(2 * B + i) * nf + j]) / (float)nf;
                if (row >= 2 * B) v = -v;
            }
            gf[t] = v;
        } else if (t < n_f + n_a) {
            const long long q = t - n_f;
            const int row = (int)(q / ncls), k = (int)(q - (long long)row * ncls);
            ga[q] = row < B ? ca * (probs[q] - (labels[row] == k ? 1.f : 0.f)) : 0.f;
        } else {
            const int row = (int)(t - n_f - n_a);
            float v;
            if (row < B || row >= 2 * B) {
                const int i = row < B ? row : row - 2 * B;
                const float on = (ct_i[i] - M >= 0.f) ? cc : 0.f;
                v = on * l2 * 2.f * (d[i] - d[2 * B + i]);
                v = row < B ? v - cw : -v;
            } else {
                v = cw;
            }
            gd[row] = v;
        }
    }
}
// accuracies of the clean pass (:249-266): logits [2B, ncls] (real rows then fake rows), labels [B]; acc[0] real, acc[1] fake
__global__ __launch_bounds__(256) void accuracy2_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels, int B,
                                                        int ncls, float* __restrict__ acc) {
    __shared__ float sh[4];
    float cr = 0.f, cf = 0.f;
    for (int i = threadIdx.x; i < 2 * B; i += 256) {
        const float* z = logits + (long long)i * ncls;
        float mx = z[0]; int am = 0;
        for (int k = 1; k < ncls; ++k) if (z[k] > mx) { mx = z[k]; am = k; }
        const float hit = (am == labels[i < B ? i : i - B]) ? 1.f : 0.f;
        if (i < B) cr += hit; else cf += hit;
    }
    cr = block_sum(cr, sh); cf = block_sum(cf, sh);
    if (threadIdx.x == 0) { acc[0] = cr / (float)B; acc[1] = cf / (float)B; }
}

}  // namespace

extern "C" {

int ctgan_gp_fwd(const float* g, int32_t b, int32_t d, float lambda, float* slopes, float* gp, ctgan_stream_t s) {
    if (!g || !slopes || !gp || b <= 0 || d <= 0) return ctgan_fail(CTGAN_E_BADARG, "gp_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(s);
    hipLaunchKernelGGL(gp_slopes_kernel, dim3(b), dim3(256), 0, st, g, d, slopes);
    int rc = ctgan_check_launch("gp_slopes");
    if (rc) return rc;
    hipLaunchKernelGGL(gp_mean_kernel, dim3(1), dim3(256), 0, st, slopes, b, lambda, gp);
    return ctgan_check_launch("gp_mean");
}
int ctgan_gp_bwd(const float* g, const float* slopes, const float* gout, int32_t b, int32_t d, float lambda, float* gg,
                 ctgan_stream_t s) {
    if (!g || !slopes || !gout || !gg || b <= 0 || d <= 0) return ctgan_fail(CTGAN_E_BADARG, "gp_bwd: bad argument");
    hipLaunchKernelGGL(gp_bwd_kernel, dim3(ctgan_blocks((long long)b * d, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(s), g, slopes, gout, b, d, lambda, gg);
    return ctgan_check_launch("gp_bwd");
}

int ctgan_ct_fwd(const float* d, const float* d_, const float* f, const float* f_, int32_t b, int32_t nf, float lambda2,
                 float M, float* ct_i, float* ct, ctgan_stream_t s) {
    if (!d || !d_ || !f || !f_ || !ct_i || !ct || b <= 0 || nf <= 0) return ctgan_fail(CTGAN_E_BADARG, "ct_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(s);
    hipLaunchKernelGGL(ct_rows_kernel, dim3(b), dim3(256), 0, st, d, d_, f, f_, nf, lambda2, ct_i);
    int rc = ctgan_check_launch("ct_rows");
    if (rc) return rc;
    hipLaunchKernelGGL(ct_mean_kernel, dim3(1), dim3(256), 0, st, ct_i, b, M, ct);
    return ctgan_check_launch("ct_mean");
}
int ctgan_ct_bwd(const float* d, const float* d_, const float* f, const float* f_, const float* ct_i, const float* gout,
                 int32_t b, int32_t nf, float lambda2, float M, float* gd, float* gd_, float* gf, float* gf_,
                 ctgan_stream_t s) {
    if (!d || !d_ || !f || !f_ || !ct_i || !gout || !gd || !gd_ || !gf || !gf_ || b <= 0 || nf <= 0)
        return ctgan_fail(CTGAN_E_BADARG, "ct_bwd: bad argument");
    hipLaunchKernelGGL(ct_bwd_kernel, dim3(ctgan_blocks((long long)b * (nf + 1), 256)), dim3(256), 0,
                       static_cast<hipStream_t>(s), d, d_, f, f_, ct_i, gout, b, nf, lambda2, M, gd, gd_, gf, gf_);
    return ctgan_check_launch("ct_bwd");
}

int ctgan_softmax_ce_fwd(const float* logits, const int32_t* labels, int32_t b, int32_t ncls, float* probs, float* loss,
                         float* n_correct, ctgan_stream_t s) {
    if (!logits || !labels || !probs || !loss || b <= 0 || ncls <= 0) return ctgan_fail(CTGAN_E_BADARG, "softmax_ce_fwd: bad argument");
    hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(s), logits, labels, b, ncls,
                       probs, loss, n_correct);
    return ctgan_check_launch("softmax_ce_fwd");
}
int ctgan_softmax_ce_bwd(const float* probs, const int32_t* labels, const float* gout, int32_t b, int32_t ncls,
                         float* glogits, ctgan_stream_t s) {
    if (!probs || !labels || !gout || !glogits || b <= 0 || ncls <= 0) return ctgan_fail(CTGAN_E_BADARG, "softmax_ce_bwd: bad argument");
    hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3((b * ncls + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(s), probs,
                       labels, gout, b, ncls, glogits);
    return ctgan_check_launch("softmax_ce_bwd");
}

int ctgan_critic_heads_fwd(const float* d, const float* f, const float* a, const int32_t* labels, const float* gp, int32_t B,
                           int32_t nf, int32_t ncls, float lambda2, float M, float acgan_scale, float* ct_i, float* probs,
                           float* out, ctgan_stream_t s) {
    if (!d || !f || !ct_i || !out || B <= 0 || nf <= 0 || (a && (!labels || !probs || ncls <= 0)))
        return ctgan_fail(CTGAN_E_BADARG, "critic_heads_fwd: bad argument");
    hipLaunchKernelGGL(critic_heads_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(s), d, f, a, labels, gp, B, nf, ncls,
                       lambda2, M, acgan_scale, ct_i, probs, out);
    return ctgan_check_launch("critic_heads_fwd");
}
int ctgan_critic_heads_bwd(const float* d, const float* f, const float* probs, const int32_t* labels, const float* ct_i,
                           const float* gout, int32_t n_gout, int32_t B, int32_t nf, int32_t ncls, float lambda2, float M,
                           float acgan_scale, float* gd, float* gf, float* ga, ctgan_stream_t s) {
    if (!d || !f || !ct_i || !gout || (n_gout != 1 && n_gout != 4) || !gd || !gf || B <= 0 || nf <= 0 || (ga && (!probs || !labels || ncls <= 0)))
        return ctgan_fail(CTGAN_E_BADARG, "critic_heads_bwd: bad argument");
    const long long total = 3LL * B * nf + (ga ? 3LL * B * ncls : 0) + 3LL * B;
    hipLaunchKernelGGL(critic_heads_bwd_kernel, dim3(ctgan_blocks(total, 256)), dim3(256), 0, static_cast<hipStream_t>(s), d, f, probs,
                       labels, ct_i, gout, n_gout, B, nf, ncls, lambda2, M, acgan_scale, gd, gf, ga);
    return ctgan_check_launch("critic_heads_bwd");
}
int ctgan_accuracy2(const float* logits, const int32_t* labels, int32_t B, int32_t ncls, float* acc, ctgan_stream_t s) {
    if (!logits || !labels || !acc || B <= 0 || ncls <= 0) return ctgan_fail(CTGAN_E_BADARG, "accuracy2: bad argument");
    hipLaunchKernelGGL(accuracy2_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(s), logits, labels, B, ncls, acc);
    return ctgan_check_launch("accuracy2");
}

int ctgan_mean_diff_fwd(const float* x, int32_t na, int32_t nb, float sa, float sb, float* out, ctgan_stream_t s) {
    if (!x || !out || na < 0 || nb < 0 || na + nb <= 0) return ctgan_fail(CTGAN_E_BADARG, "mean_diff_fwd: bad argument");
    hipLaunchKernelGGL(mean_diff_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(s), x, na, nb, sa, sb, out);
    return ctgan_check_launch("mean_diff_fwd");
}
int ctgan_mean_diff_bwd(const float* gout, int32_t na, int32_t nb, float sa, float sb, float* gx, ctgan_stream_t s) {
    if (!gout || !gx || na < 0 || nb < 0 || na + nb <= 0) return ctgan_fail(CTGAN_E_BADARG, "mean_diff_bwd: bad argument");
    hipLaunchKernelGGL(mean_diff_bwd_kernel, dim3((na + nb + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(s), gout, na,
                       nb, sa, sb, gx);
    return ctgan_check_launch("mean_diff_bwd");
}

}  // extern "C"
