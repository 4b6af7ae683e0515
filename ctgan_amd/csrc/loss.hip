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

// gp_bwd_kernel + the penalty's value: block 0 also takes gp = lambda * mean((slopes - 1)^2) (b values) and adds it into out5[0] (cost) and
// out5[4] (wgan + ct + gp) - the two sums of ctgan_tail_critic_heads_fwd that contain the penalty.  For the hand-scheduled critic step
// (round 5), whose loss heads run BEFORE the penalty's gradient exists: no extra launch for a number only the log reads.
__global__ void gp_bwd_mean_kernel(const float* __restrict__ g, const float* __restrict__ slopes, const float* __restrict__ gout,
                                   int b, int d, float lambda, float* __restrict__ gg, float* __restrict__ gp, float* __restrict__ out5) {
    const long long total = (long long)b * d;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float go = gout[0];
    if (blockIdx.x == 0) {
        __shared__ float sh[4];
        float s = 0.f;
        for (int i = threadIdx.x; i < b; i += 256) { const float t = slopes[i] - 1.f; s += t * t; }
        s = block_sum(s, sh);
        if (threadIdx.x == 0) {
            const float v = lambda * s / (float)b;
            if (gp) gp[0] = v;
            if (out5) { out5[0] += v; out5[4] += v; }
        }
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float s = slopes[i / d];
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
// CT_i for wide feature rows (the DCGAN critics' 8192 features): one workgroup per sample, float4 loads
__global__ __launch_bounds__(256) void critic_heads_ct_rows_kernel(const float* __restrict__ d, const float* __restrict__ f, int B, int nf,
                                                                   float l2, float* __restrict__ ct_i) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const float* fa = f + (long long)i * nf;
    const float* fb = f + (long long)(2 * B + i) * nf;
    float s = 0.f;
    if ((nf & 3) == 0) {
        for (int j = threadIdx.x * 4; j < nf; j += 1024) {
            const float4 p = *reinterpret_cast<const float4*>(fa + j), q = *reinterpret_cast<const float4*>(fb + j);
            const float t0 = p.x - q.x, t1 = p.y - q.y, t2 = p.z - q.z, t3 = p.w - q.w;
            s += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
        }
    } else {
        for (int j = threadIdx.x; j < nf; j += 256) { const float t = fa[j] - fb[j]; s += t * t; }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) { const float t = d[i] - d[2 * B + i]; ct_i[i] = l2 * t * t + l2 * 0.1f * (s / (float)nf); }
}
__global__ __launch_bounds__(256) void critic_heads_fwd_kernel(const float* __restrict__ d, const float* __restrict__ f,
                                                               const float* __restrict__ a, const int32_t* __restrict__ labels,
                                                               const float* __restrict__ gp, int B, int nf, int ncls, float l2,
                                                               float M, float scale, int have_ct, float* __restrict__ ct_i,
                                                               float* __restrict__ probs, float* __restrict__ out) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = w; i < B && !have_ct; i += 4) {           // one wave per row
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


// ---- the critic's output head fused around the loss heads (CIFAR ResNet critic, TF/CT_gan_cifar_resnet.py:179-186) -----
// y = relu(dropout(.)) output of the last residual block, physical [n][hw][nf] (nf % 4 == 0, nf <= 1024).  Per row:
//   f[row,:] = mean_hw y [relu'd first if `relu`]   (:179-180 relu, reduce_mean over axes 2,3)
//   d[row]   = f . w_out + b_out                    (:181 Linear nf -> 1)
//   a[row,:] = f . w_ac + b_ac                      (:183 Linear nf -> ncls)
// replaces [relu +] spatial_sum + two linear launches.  blockDim = 256.
__device__ __forceinline__ void head_row(const float* __restrict__ yr, int hw, int nf, int relu, const float* __restrict__ w_out,
                                         const float* __restrict__ b_out, const float* __restrict__ w_ac, const float* __restrict__ b_ac,
                                         int ncls, float* part /*[<=1024]*/, float* fs /*[nf]*/, float* f_row, float* d_row, float* a_row,
                                         float* d_sh) {
    const int tid = threadIdx.x;
    const int nf4 = nf >> 2;
    const int S = 256 / nf4;                          // position slices (nf4 <= 256)
    const int s = tid / nf4, j4 = tid - s * nf4;
    if (s < S) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int p = s; p < hw; p += S) {
            float4 v = *reinterpret_cast<const float4*>(yr + (long long)p * nf + j4 * 4);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(part + s * nf + j4 * 4) = acc;
    }
    __syncthreads();
    const float inv = 1.f / (float)hw;
    for (int j = tid; j < nf; j += 256) {
        float acc = part[j];
        for (int q = 1; q < S; ++q) acc += part[q * nf + j];
        acc *= inv;
        fs[j] = acc;
        f_row[j] = acc;
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6;
    const int nout = (w_out ? 1 : 0) + (w_ac ? ncls : 0);
    for (int o = w; o < nout; o += 4) {
        const bool is_d = w_out && o == 0;
        const int k = o - (w_out ? 1 : 0);
        float acc = 0.f;
        if (is_d) { for (int j = lane; j < nf; j += 64) acc += fs[j] * w_out[j]; }
        else      { for (int j = lane; j < nf; j += 64) acc += fs[j] * w_ac[(long long)j * ncls + k]; }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (is_d) { const float v = acc + (b_out ? b_out[0] : 0.f); d_row[0] = v; if (d_sh) d_sh[0] = v; }
            else a_row[k] = acc + (b_ac ? b_ac[k] : 0.f);
        }
    }
    __syncthreads();
}

// pair_B == 0: one workgroup per row.  pair_B == B > 0 (critic step, rows = real pass 1 | fake pass 1 | real pass 2):
// workgroup r < B handles rows r and 2B + r and also writes the per-sample loss terms of ctgan_critic_heads_fwd
// (ct_i[r], probs[r,:], ce_i[r] = softmax-CE of a[r] against labels[r]); workgroup r in [B, 2B) handles row r.
__global__ __launch_bounds__(256) void tail_heads_rows_kernel(const float* __restrict__ y, int hw, int nf, int relu,
                                                              const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                              const float* __restrict__ w_ac, const float* __restrict__ b_ac, int ncls,
                                                              float* __restrict__ f, float* __restrict__ d, float* __restrict__ a,
                                                              int pair_B, const int32_t* __restrict__ labels, float l2,
                                                              float* __restrict__ ct_i, float* __restrict__ probs, float* __restrict__ ce_i,
                                                              const float* __restrict__ y2, int n_main, int relu2,
                                                              float* __restrict__ f2, float* __restrict__ a2) {
    __shared__ __attribute__((aligned(16))) float part[1024];
    __shared__ float fs0[1024], fs1[1024];
    __shared__ float dsh[2];
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const long long row_elems = (long long)hw * nf;
    if (y2 && r >= n_main) {        // rows of a second tensor that only feed the class head (the clean pass of the accuracies, :249-266)
        const int q = r - n_main;
        head_row(y2 + q * row_elems, hw, nf, relu2, nullptr, nullptr, w_ac, b_ac, ncls, part, fs0, f2 + (long long)q * nf, nullptr,
                 a2 + (long long)q * ncls, dsh);
        return;
    }
    head_row(y + r * row_elems, hw, nf, relu, w_out, b_out, w_ac, b_ac, ncls, part, fs0, f + (long long)r * nf, d ? d + r : nullptr,
             a ? a + (long long)r * ncls : nullptr, dsh);
    if (pair_B == 0 || r >= pair_B) return;
    const int r2 = 2 * pair_B + r;
    head_row(y + r2 * row_elems, hw, nf, relu, w_out, b_out, w_ac, b_ac, ncls, part, fs1, f + (long long)r2 * nf, d ? d + r2 : nullptr,
             a ? a + (long long)r2 * ncls : nullptr, dsh + 1);
    float sq = 0.f;
    for (int j = threadIdx.x; j < nf; j += 256) { const float t = fs0[j] - fs1[j]; sq += t * t; }
    sq = block_sum(sq, sh);
    if (threadIdx.x == 0) {
        const float t = dsh[0] - dsh[1];
        ct_i[r] = l2 * t * t + l2 * 0.1f * (sq / (float)nf);                     // :288-290
        if (a) {
            const float* z = a + (long long)r * ncls;                            // written above by this workgroup
            float mx = z[0];
            for (int k = 1; k < ncls; ++k) mx = fmaxf(mx, z[k]);
            float se = 0.f;
            for (int k = 0; k < ncls; ++k) se += expf(z[k] - mx);
            const float lse = logf(se);
            for (int k = 0; k < ncls; ++k) probs[(long long)r * ncls + k] = expf(z[k] - mx - lse);
            ce_i[r] = (mx + lse) - z[labels[r]];                                 // :246-248
        }
    }
}
// the batch means over those per-sample terms: out[5] as ctgan_critic_heads_fwd
// slopes != NULL: the gradient penalty's batch mean is taken here (gp_mean_kernel's arithmetic) and WRITTEN to gp[0] before it is used;
// a_clean != NULL: the accuracies of the clean pass (accuracy2_kernel's arithmetic) are written to acc[2].
__global__ __launch_bounds__(256) void critic_heads_final_kernel(const float* __restrict__ d, const float* __restrict__ ct_i,
                                                                 const float* __restrict__ ce_i, float* gp, int B,
                                                                 float M, float scale, float* __restrict__ out,
                                                                 const float* __restrict__ slopes, float gp_lambda,
                                                                 const float* __restrict__ a_clean, const int32_t* __restrict__ labels,
                                                                 int ncls, float* __restrict__ acc) {
    __shared__ float sh[4];
    if (slopes) {
        float s = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) { const float t = slopes[i] - 1.f; s += t * t; }
        s = block_sum(s, sh);
        if (threadIdx.x == 0) gp[0] = gp_lambda * s / (float)B;
        __syncthreads();
    }
    if (a_clean) {
        float cr = 0.f, cf = 0.f;
        for (int i = threadIdx.x; i < 2 * B; i += 256) {
            const float* z = a_clean + (long long)i * ncls;
            float mx = z[0]; int am = 0;
            for (int k = 1; k < ncls; ++k) if (z[k] > mx) { mx = z[k]; am = k; }
            const float hit = (am == labels[i < B ? i : i - B]) ? 1.f : 0.f;
            if (i < B) cr += hit; else cf += hit;
        }
        cr = block_sum(cr, sh); cf = block_sum(cf, sh);
        if (threadIdx.x == 0) { acc[0] = cr / (float)B; acc[1] = cf / (float)B; }
    }
    float sr = 0.f, sf = 0.f, sc = 0.f, sl = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        sr += d[i]; sf += d[B + i];
        sc += fmaxf(ct_i[i] - M, 0.f);
        if (ce_i) sl += ce_i[i];
    }
    sr = block_sum(sr, sh); sf = block_sum(sf, sh); sc = block_sum(sc, sh); sl = block_sum(sl, sh);
    if (threadIdx.x == 0) {
        const float wgan = sf / (float)B - sr / (float)B, ct = sc / (float)B, ac = ce_i ? sl / (float)B : 0.f;
        const float pen = gp ? gp[0] : 0.f;
        out[0] = ((wgan + ct) + pen) + scale * ac; out[1] = wgan; out[2] = ct; out[3] = ac; out[4] = (wgan + ct) + pen;
    }
}

// per-row upstream gradients of the critic loss heads (the formulas of critic_heads_bwd_kernel)
struct HeadCoef { float cw, cc, ca; };
__device__ __forceinline__ HeadCoef head_coef(const float* gout, int n_gout, int B, float scale) {
    const float g0 = gout[0], g1 = n_gout > 1 ? gout[1] : 0.f, g2 = n_gout > 1 ? gout[2] : 0.f, g3 = n_gout > 1 ? gout[3] : 0.f;
    HeadCoef c;
    c.cw = (g0 + g1) / (float)B; c.cc = (g0 + g2) / (float)B; c.ca = (g0 * scale + g3) / (float)B;
    return c;
}
__device__ __forceinline__ float head_gd(const float* d, const float* ct_i, int row, int B, float l2, float M, HeadCoef c) {
    if (row < B || row >= 2 * B) {
        const int i = row < B ? row : row - 2 * B;
        const float on = (ct_i[i] - M >= 0.f) ? c.cc : 0.f;
        const float v = on * l2 * 2.f * (d[i] - d[2 * B + i]);
        return row < B ? v - c.cw : -v;
    }
    return c.cw;
}

// Backward of {rows kernel + loss heads + relu/dropout mask of the last conv} in one launch.
//   workgroups [0, 3B): gy[row,hw,j] = y > 0 ? (gf[j] + gd*w_out[j] + sum_k ga[k]*w_ac[j,k]) / hw * mask_scale : 0
//   workgroups [3B, ..): gw_out[j] = sum_row f[row,j]*gd[row], gb_out = sum gd, gw_ac[j,k] = sum_row f[row,j]*ga[row,k], gb_ac
// (fixed summation order over rows => deterministic).  blockDim = 256; dynamic shared: max(nf, 3B) floats.
__global__ __launch_bounds__(256) void tail_heads_bwd_kernel(const float* __restrict__ y, const float* __restrict__ d,
                                                             const float* __restrict__ f, const float* __restrict__ probs,
                                                             const int32_t* __restrict__ labels, const float* __restrict__ ct_i,
                                                             const float* __restrict__ gout, int n_gout, int B, int hw, int nf, int ncls,
                                                             float l2, float M, float scale, float mask_scale,
                                                             const float* __restrict__ w_out, const float* __restrict__ w_ac,
                                                             float* __restrict__ gy, float* __restrict__ gw_out, float* __restrict__ gb_out,
                                                             float* __restrict__ gw_ac, float* __restrict__ gb_ac,
                                                             const float* __restrict__ y_gp, int n_gp, float* __restrict__ gz_gp) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    const HeadCoef c = head_coef(gout, n_gout, B, scale);
    if ((int)blockIdx.x >= 3 * B && (int)blockIdx.x < 3 * B + n_gp) {
        // rows of the gradient-penalty pass (gp_head_grad_kernel's formula): gz = y > 0 ? w_out / hw * mask_scale : 0 - the seed of their
        // backward, written by the launch that seeds the dropout passes' (the hand-scheduled step runs both chains as one)
        const long long base = (long long)((int)blockIdx.x - 3 * B) * hw * nf;
        const int n4 = (hw * nf) >> 2;
        const float sc = mask_scale / (float)hw;
        for (int q = tid; q < n4; q += 256) {
            const int j = (q * 4) % nf;
            const float4 yv = *reinterpret_cast<const float4*>(y_gp + base + (long long)q * 4);
            const float4 wv = *reinterpret_cast<const float4*>(w_out + j);
            float4 o;
            o.x = yv.x > 0.f ? wv.x * sc : 0.f; o.y = yv.y > 0.f ? wv.y * sc : 0.f;
            o.z = yv.z > 0.f ? wv.z * sc : 0.f; o.w = yv.w > 0.f ? wv.w * sc : 0.f;
            *reinterpret_cast<float4*>(gz_gp + base + (long long)q * 4) = o;
        }
        return;
    }
    if ((int)blockIdx.x < 3 * B) {
        const int row = blockIdx.x;
        const float gd = head_gd(d, ct_i, row, B, l2, M, c);
        float* t = sm;                                   // [nf]
        for (int j = tid; j < nf; j += 256) {
            float v = 0.f;
            if (row < B || row >= 2 * B) {
                const int i = row < B ? row : row - 2 * B;
                const float on = (ct_i[i] - M >= 0.f) ? c.cc : 0.f;
                v = on * l2 * 0.1f * 2.f * (f[(long long)i * nf + j] - f[(long long)(2 * B + i) * nf + j]) / (float)nf;
                if (row >= 2 * B) v = -v;
            }
            v += gd * w_out[j];
            if (w_ac && row < B) {
                const int lab = labels[row];
                for (int k = 0; k < ncls; ++k)
                    v += c.ca * (probs[(long long)row * ncls + k] - (lab == k ? 1.f : 0.f)) * w_ac[(long long)j * ncls + k];
            }
            t[j] = v * (1.f / (float)hw);
        }
        __syncthreads();
        const long long base = (long long)row * hw * nf;
        const int n4 = (hw * nf) >> 2;                   // nf % 4 == 0
        for (int q = tid; q < n4; q += 256) {
            const int j = (q * 4) % nf;
            const float4 yv = *reinterpret_cast<const float4*>(y + base + (long long)q * 4);
            float4 o;
            o.x = yv.x > 0.f ? t[j] * mask_scale : 0.f;     o.y = yv.y > 0.f ? t[j + 1] * mask_scale : 0.f;
            o.z = yv.z > 0.f ? t[j + 2] * mask_scale : 0.f; o.w = yv.w > 0.f ? t[j + 3] * mask_scale : 0.f;
            *reinterpret_cast<float4*>(gy + base + (long long)q * 4) = o;
        }
        return;
    }
    float* gds = sm;                                     // [3B]
    for (int r = tid; r < 3 * B; r += 256) gds[r] = head_gd(d, ct_i, r, B, l2, M, c);
    __syncthreads();
    const int nac = w_ac ? ncls : 0;
    const int n_w = nf * (1 + nac), n_all = n_w + 1 + nac;
    const int lane = tid & 63;
    const int idx = ((int)blockIdx.x - 3 * B - n_gp) * 4 + (tid >> 6);   // one wave per output, lanes over rows (fixed order)
    if (idx >= n_all) return;
    float acc = 0.f;
    if (idx < nf) {
        for (int r = lane; r < 3 * B; r += 64) acc += f[(long long)r * nf + idx] * gds[r];
    } else if (idx < n_w) {
        const int q = idx - nf, k = q / nf, j = q - k * nf;
        for (int r = lane; r < B; r += 64)
            acc += f[(long long)r * nf + j] * (c.ca * (probs[(long long)r * ncls + k] - (labels[r] == k ? 1.f : 0.f)));
    } else if (idx == n_w) {
        for (int r = lane; r < 3 * B; r += 64) acc += gds[r];
    } else {
        const int k = idx - n_w - 1;
        for (int r = lane; r < B; r += 64) acc += c.ca * (probs[(long long)r * ncls + k] - (labels[r] == k ? 1.f : 0.f));
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (idx < nf) gw_out[idx] = acc;
        else if (idx < n_w) { const int q = idx - nf, k = q / nf, j = q - k * nf; gw_ac[(long long)j * ncls + k] = acc; }
        else if (idx == n_w) gb_out[0] = acc;
        else gb_ac[idx - n_w - 1] = acc;
    }
}

// ---- generator-step loss on the critic's outputs (TF/CT_gan_cifar_resnet.py:321-330): cost = -mean(d) + scale * CE(a, labels) ---------
// forward: one workgroup (batch means; the softmax probabilities are kept for the backward)
__global__ __launch_bounds__(256) void gen_heads_loss_kernel(const float* __restrict__ d, const float* __restrict__ a,
                                                             const int32_t* __restrict__ labels, int n, int ncls, float scale,
                                                             float* __restrict__ probs, float* __restrict__ out) {
    __shared__ float sh[4];
    float sd = 0.f, sl = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        sd += d[i];
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
    sd = block_sum(sd, sh); sl = block_sum(sl, sh);
    if (threadIdx.x == 0) out[0] = -(sd / (float)n) + (a ? scale * (sl / (float)n) : 0.f);
}
// backward, one workgroup per sample: gradient w.r.t. the last conv's result through both Linear layers, the spatial mean and
// the relu/dropout mask:  gy[row,hw,j] = y > 0 ? gout * (-w_out[j] + scale * sum_k (p_k - 1[k = label]) w_ac[j,k]) / n / hw * mask_scale : 0
__global__ __launch_bounds__(256) void gen_heads_bwd_kernel(const float* __restrict__ y, const float* __restrict__ probs,
                                                            const int32_t* __restrict__ labels, const float* __restrict__ gout, int n,
                                                            int hw, int nf, int ncls, float scale, float mask_scale,
                                                            const float* __restrict__ w_out, const float* __restrict__ w_ac,
                                                            float* __restrict__ gy) {
    extern __shared__ float sm[];                          // t [nf]
    const int row = blockIdx.x, tid = threadIdx.x;
    const float g0 = gout[0] / (float)n;
    for (int j = tid; j < nf; j += 256) {
        float v = -g0 * w_out[j];
        if (w_ac) {
            const int lab = labels[row];
            for (int k = 0; k < ncls; ++k)
                v += g0 * scale * (probs[(long long)row * ncls + k] - (lab == k ? 1.f : 0.f)) * w_ac[(long long)j * ncls + k];
        }
        sm[j] = v * (1.f / (float)hw);
    }
    __syncthreads();
    const long long base = (long long)row * hw * nf;
    const int n4 = (hw * nf) >> 2;
    for (int q = tid; q < n4; q += 256) {
        const int j = (q * 4) % nf;
        const float4 yv = *reinterpret_cast<const float4*>(y + base + (long long)q * 4);
        float4 o;
        o.x = yv.x > 0.f ? sm[j] * mask_scale : 0.f;     o.y = yv.y > 0.f ? sm[j + 1] * mask_scale : 0.f;
        o.z = yv.z > 0.f ? sm[j + 2] * mask_scale : 0.f; o.w = yv.w > 0.f ? sm[j + 3] * mask_scale : 0.f;
        *reinterpret_cast<float4*>(gy + base + (long long)q * 4) = o;
    }
}

// Gradient-penalty branch: dD/dy of D = mean_hw(y) . w_out for the last block's output y = relu(dropout(z)), taken w.r.t. z:
//   gz[row,hw,j] = y > 0 ? w_out[j] / hw * mask_scale : 0      (the value D(x_hat) itself is never needed, :284)
__global__ void gp_head_grad_kernel(const float* __restrict__ y, const float* __restrict__ w_out, long long n4, int nf, float s,
                                    float* __restrict__ gz) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4) return;
    const int j = (int)((q * 4) % nf);
    const float4 yv = *reinterpret_cast<const float4*>(y + q * 4);
    const float4 wv = *reinterpret_cast<const float4*>(w_out + j);
    float4 o;
    o.x = yv.x > 0.f ? wv.x * s : 0.f; o.y = yv.y > 0.f ? wv.y * s : 0.f;
    o.z = yv.z > 0.f ? wv.z * s : 0.f; o.w = yv.w > 0.f ? wv.w * s : 0.f;
    *reinterpret_cast<float4*>(gz + q * 4) = o;
}
// its adjoint w.r.t. w_out (the double backward): gw[j] = s * sum_{row,hw : y > 0} gg[row,hw,j].  Two fixed-order stages:
// GPW_SLICES workgroups each reduce a contiguous slice of the (row, hw) axis with coalesced 16-B loads over all channels,
// a second launch sums the slice partials (a single-stage kernel over 4 channels per workgroup reads 16 B out of every 512 B
// line: 18 us for 4 MB).
constexpr int GPW_SLICES = 64;
__global__ __launch_bounds__(256) void gp_head_wgrad_stage1_kernel(const float* __restrict__ gg, const float* __restrict__ y, long long rows,
                                                                   int nf, float* __restrict__ part /*[GPW_SLICES][nf]*/) {
    __shared__ float4 red[256];
    const int c4n = nf >> 2, rls = 256 / c4n;                     // (nf/4) channel lanes x row lanes; nf/4 divides 256 (host check)
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    const long long per = (rows + GPW_SLICES - 1) / GPW_SLICES;
    const long long r0 = (long long)blockIdx.x * per, r1 = min(rows, r0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += rls) {
        const float4 g = *reinterpret_cast<const float4*>(gg + r * nf + c4 * 4);
        const float4 yv = *reinterpret_cast<const float4*>(y + r * nf + c4 * 4);
        acc.x += yv.x > 0.f ? g.x : 0.f; acc.y += yv.y > 0.f ? g.y : 0.f;
        acc.z += yv.z > 0.f ? g.z : 0.f; acc.w += yv.w > 0.f ? g.w : 0.f;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        float4 t = red[c4];
        for (int r = 1; r < rls; ++r) { const float4 v = red[r * c4n + c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(part + (long long)blockIdx.x * nf + c4 * 4) = t;
    }
}
__global__ __launch_bounds__(256) void gp_head_wgrad_stage2_kernel(const float* __restrict__ part, int nf, float s, float* __restrict__ gw,
                                                                   int accumulate) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nf) return;
    float acc = 0.f;
    for (int k0 = 0; k0 < GPW_SLICES; k0 += 8) {                  // 8 independent loads in flight, fixed summation order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k0 + u) * nf + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    gw[j] = accumulate ? gw[j] + acc * s : acc * s;
}

// End of the penalty's first backward (hand-scheduled critic step): dD/dx_hat = ga + scale * upsample2(gs) - the gradient through the first
// conv plus the gradient through the pooled 1x1 shortcut (TF/CT_gan_cifar_resnet.py:146-153) - written over ga, and slopes[b] = ||.||_2 of
// the sample: one workgroup per sample instead of an upsample, an add and the slopes launch.
__global__ __launch_bounds__(256) void gp_finish_kernel(float* __restrict__ ga, const float* __restrict__ gs, long long ss_n, long long ss_c,
                                                        long long ss_h, long long ss_w, int C, int H, int W, float scale,
                                                        float* __restrict__ slopes) {
    __shared__ float sh[4];
    const int d = C * H * W;
    float* row = ga + (long long)blockIdx.x * d;
    const float* srow = gs + (long long)blockIdx.x * ss_n;
    float s = 0.f;
    for (int i = threadIdx.x; i < d; i += 256) {
        const int c = i / (H * W), rem = i - c * H * W, h = rem / W, w = rem - h * W;
        const float v = row[i] + scale * srow[c * ss_c + (h >> 1) * ss_h + (w >> 1) * ss_w];
        row[i] = v;
        s += v * v;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) slopes[blockIdx.x] = sqrtf(s);
}

}  // namespace

extern "C" {

int ctgan_gp_fwd(const float* g, int32_t b, int32_t d, float lambda, float* slopes, float* gp, ctgan_stream_t s) {
    if (!g || !slopes || b <= 0 || d <= 0) return ctgan_fail(CTGAN_E_BADARG, "gp_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(s);
    hipLaunchKernelGGL(gp_slopes_kernel, dim3(b), dim3(256), 0, st, g, d, slopes);
    int rc = ctgan_check_launch("gp_slopes");
    if (rc || !gp) return rc;                        // gp == NULL: the mean is taken by ctgan_tail_critic_heads_fwd2 (from slopes)
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

int ctgan_gp_bwd_mean(const float* g, const float* slopes, const float* gout, int32_t b, int32_t d, float lambda, float* gg, float* gp,
                      float* out5, ctgan_stream_t s) {
    if (!g || !slopes || !gout || !gg || b <= 0 || d <= 0) return ctgan_fail(CTGAN_E_BADARG, "gp_bwd_mean: bad argument");
    hipLaunchKernelGGL(gp_bwd_mean_kernel, dim3(ctgan_blocks((long long)b * d, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(s), g, slopes, gout, b, d, lambda, gg, gp, out5);
    return ctgan_check_launch("gp_bwd_mean");
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
    const int wide = (long long)B * nf >= 65536 && !(reinterpret_cast<uintptr_t>(f) & 15);     // wide rows: CT_i by one workgroup per sample
    if (wide)
        hipLaunchKernelGGL(critic_heads_ct_rows_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(s), d, f, B, nf, lambda2, ct_i);
    hipLaunchKernelGGL(critic_heads_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(s), d, f, a, labels, gp, B, nf, ncls,
                       lambda2, M, acgan_scale, wide, ct_i, probs, out);
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
int ctgan_tail_heads_fwd(const float* y, int32_t n, int32_t hw, int32_t nf, int32_t relu, const float* w_out, const float* b_out,
                         const float* w_ac, const float* b_ac, int32_t ncls, float* f, float* d, float* a, ctgan_stream_t s) {
    if (!y || !f || n <= 0 || hw <= 0 || nf <= 0 || (nf & 3) || nf > 1024 || (w_out && !d) || (w_ac && (!a || ncls <= 0)) ||
        (reinterpret_cast<uintptr_t>(y) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "tail_heads_fwd: bad argument");
    hipLaunchKernelGGL(tail_heads_rows_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(s), y, hw, nf, relu, w_out, b_out, w_ac,
                       b_ac, ncls, f, d, a, 0, (const int32_t*)nullptr, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (const float*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr);
    return ctgan_check_launch("tail_heads_rows");
}
int ctgan_tail_critic_heads_fwd(const float* y, int32_t B, int32_t hw, int32_t nf, const float* w_out, const float* b_out,
                                const float* w_ac, const float* b_ac, int32_t ncls, const int32_t* labels, const float* gp,
                                float lambda2, float M, float acgan_scale, float* f, float* d, float* a, float* ct_i, float* probs,
                                float* ce_i, float* out, ctgan_stream_t s) {
    return ctgan_tail_critic_heads_fwd2(y, B, hw, nf, w_out, b_out, w_ac, b_ac, ncls, labels, const_cast<float*>(gp), nullptr, 0.f, nullptr, 0,
                                        nullptr, nullptr, nullptr, lambda2, M, acgan_scale, f, d, a, ct_i, probs, ce_i, out, s);
}
int ctgan_tail_critic_heads_fwd2(const float* y, int32_t B, int32_t hw, int32_t nf, const float* w_out, const float* b_out,
                                 const float* w_ac, const float* b_ac, int32_t ncls, const int32_t* labels, float* gp,
                                 const float* slopes, float gp_lambda, const float* y_clean, int32_t clean_relu, float* f_clean,
                                 float* a_clean, float* acc, float lambda2, float M, float acgan_scale, float* f, float* d, float* a,
                                 float* ct_i, float* probs, float* ce_i, float* out, ctgan_stream_t s) {
    if (!y || !f || !d || !w_out || !ct_i || !out || B <= 0 || hw <= 0 || nf <= 0 || (nf & 3) || nf > 1024 ||
        (w_ac && (!a || !labels || !probs || !ce_i || ncls <= 0)) || (reinterpret_cast<uintptr_t>(y) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "tail_critic_heads_fwd: bad argument");
    if (slopes && !gp) return ctgan_fail(CTGAN_E_BADARG, "tail_critic_heads_fwd2: slopes without a gp slot");
    if (y_clean && (!w_ac || !f_clean || !a_clean || !acc || !labels || (reinterpret_cast<uintptr_t>(y_clean) & 15)))
        return ctgan_fail(CTGAN_E_BADARG, "tail_critic_heads_fwd2: clean rows need the class head, f_clean, a_clean, acc, labels");
    hipStream_t st = static_cast<hipStream_t>(s);
    const int n_clean = y_clean ? 2 * B : 0;
    hipLaunchKernelGGL(tail_heads_rows_kernel, dim3(2 * B + n_clean), dim3(256), 0, st, y, hw, nf, 0, w_out, b_out, w_ac, b_ac, ncls, f, d,
                       w_ac ? a : (float*)nullptr, B, labels, lambda2, ct_i, probs, ce_i, y_clean, 2 * B, clean_relu, f_clean, a_clean);
    int rc = ctgan_check_launch("tail_heads_rows");
    if (rc) return rc;
    hipLaunchKernelGGL(critic_heads_final_kernel, dim3(1), dim3(256), 0, st, d, ct_i, w_ac ? ce_i : (const float*)nullptr, gp, B, M,
                       acgan_scale, out, slopes, gp_lambda, y_clean ? a_clean : (const float*)nullptr, labels, ncls, acc);
    return ctgan_check_launch("critic_heads_final");
}
int ctgan_tail_heads_bwd_gp(const float* y, const float* d, const float* f, const float* probs, const int32_t* labels, const float* ct_i,
                            const float* gout, int32_t n_gout, int32_t B, int32_t hw, int32_t nf, int32_t ncls, float lambda2, float M,
                            float acgan_scale, float mask_scale, const float* w_out, const float* w_ac, float* gy, float* gw_out,
                            float* gb_out, float* gw_ac, float* gb_ac, const float* y_gp, int32_t n_gp, float* gz_gp, ctgan_stream_t s) {
    if (!y || !d || !f || !ct_i || !gout || (n_gout != 1 && n_gout != 4) || !w_out || !gy || !gw_out || !gb_out || B <= 0 || hw <= 0 ||
        nf <= 0 || (nf & 3) || (w_ac && (!probs || !labels || !gw_ac || !gb_ac || ncls <= 0)) || n_gp < 0 || (n_gp > 0 && (!y_gp || !gz_gp)))
        return ctgan_fail(CTGAN_E_BADARG, "tail_heads_bwd: bad argument");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y_gp) | reinterpret_cast<uintptr_t>(gz_gp) |
         (n_gp > 0 ? reinterpret_cast<uintptr_t>(w_out) : 0)) & 15)
        return ctgan_fail(CTGAN_E_BADARG, "tail_heads_bwd: unaligned");
    const int nac = w_ac ? ncls : 0;
    const int wblocks = (nf * (1 + nac) + 1 + nac + 3) / 4;               // one wave per output
    const size_t sh = (size_t)(nf > 3 * B ? nf : 3 * B) * sizeof(float);
    hipLaunchKernelGGL(tail_heads_bwd_kernel, dim3(3 * B + n_gp + wblocks), dim3(256), sh, static_cast<hipStream_t>(s), y, d, f, probs, labels, ct_i,
                       gout, n_gout, B, hw, nf, ncls, lambda2, M, acgan_scale, mask_scale, w_out, w_ac, gy, gw_out, gb_out, gw_ac, gb_ac,
                       y_gp, n_gp, gz_gp);
    return ctgan_check_launch("tail_heads_bwd");
}
int ctgan_tail_heads_bwd(const float* y, const float* d, const float* f, const float* probs, const int32_t* labels, const float* ct_i,
                         const float* gout, int32_t n_gout, int32_t B, int32_t hw, int32_t nf, int32_t ncls, float lambda2, float M,
                         float acgan_scale, float mask_scale, const float* w_out, const float* w_ac, float* gy, float* gw_out,
                         float* gb_out, float* gw_ac, float* gb_ac, ctgan_stream_t s) {
    return ctgan_tail_heads_bwd_gp(y, d, f, probs, labels, ct_i, gout, n_gout, B, hw, nf, ncls, lambda2, M, acgan_scale, mask_scale, w_out, w_ac, gy,
                                   gw_out, gb_out, gw_ac, gb_ac, nullptr, 0, nullptr, s);
}
int ctgan_gen_heads_fwd(const float* y, int32_t n, int32_t hw, int32_t nf, const float* w_out, const float* b_out, const float* w_ac,
                        const float* b_ac, int32_t ncls, const int32_t* labels, float ac_scale, float* f, float* d, float* a, float* probs,
                        float* out, ctgan_stream_t s) {
    if (!y || !f || !d || !w_out || !out || n <= 0 || hw <= 0 || nf <= 0 || (nf & 3) || nf > 1024 ||
        (w_ac && (!a || !labels || !probs || ncls <= 0)) || (reinterpret_cast<uintptr_t>(y) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "gen_heads_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(s);
    hipLaunchKernelGGL(tail_heads_rows_kernel, dim3(n), dim3(256), 0, st, y, hw, nf, 0, w_out, b_out, w_ac, b_ac, ncls, f, d,
                       w_ac ? a : (float*)nullptr, 0, (const int32_t*)nullptr, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (const float*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr);
    int rc = ctgan_check_launch("tail_heads_rows");
    if (rc) return rc;
    hipLaunchKernelGGL(gen_heads_loss_kernel, dim3(1), dim3(256), 0, st, d, w_ac ? a : (const float*)nullptr, labels, n, ncls, ac_scale, probs, out);
    return ctgan_check_launch("gen_heads_loss");
}
int ctgan_gen_heads_bwd(const float* y, const float* probs, const int32_t* labels, const float* gout, int32_t n, int32_t hw, int32_t nf,
                        int32_t ncls, float ac_scale, float mask_scale, const float* w_out, const float* w_ac, float* gy, ctgan_stream_t s) {
    if (!y || !gout || !w_out || !gy || n <= 0 || hw <= 0 || nf <= 0 || (nf & 3) || (w_ac && (!probs || !labels || ncls <= 0)) ||
        ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gy)) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "gen_heads_bwd: bad argument");
    hipLaunchKernelGGL(gen_heads_bwd_kernel, dim3(n), dim3(256), (size_t)nf * sizeof(float), static_cast<hipStream_t>(s), y, probs, labels, gout,
                       n, hw, nf, ncls, ac_scale, mask_scale, w_out, w_ac, gy);
    return ctgan_check_launch("gen_heads_bwd");
}
int ctgan_gp_head_grad(const float* y, const float* w_out, int32_t n, int32_t hw, int32_t nf, float mask_scale, float* gz,
                       ctgan_stream_t s) {
    if (!y || !w_out || !gz || n <= 0 || hw <= 0 || nf <= 0 || (nf & 3)) return ctgan_fail(CTGAN_E_BADARG, "gp_head_grad: bad argument");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gz) | reinterpret_cast<uintptr_t>(w_out)) & 15)
        return ctgan_fail(CTGAN_E_BADARG, "gp_head_grad: unaligned");
    const long long n4 = (long long)n * hw * nf / 4;
    hipLaunchKernelGGL(gp_head_grad_kernel, dim3(ctgan_blocks(n4, 256)), dim3(256), 0, static_cast<hipStream_t>(s), y, w_out, n4, nf,
                       mask_scale / (float)hw, gz);
    return ctgan_check_launch("gp_head_grad");
}
int ctgan_gp_head_wgrad(const float* gg, const float* y, int32_t n, int32_t hw, int32_t nf, float mask_scale, float* gw, float* ws,
                        ctgan_stream_t s) {
    const int c4n = nf >> 2;
    if (!gg || !y || !gw || !ws || n <= 0 || hw <= 0 || nf <= 0 || (nf & 3) || c4n > 256 || (256 % c4n))
        return ctgan_fail(CTGAN_E_BADARG, "gp_head_wgrad: bad argument");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gg) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return ctgan_fail(CTGAN_E_BADARG, "gp_head_wgrad: unaligned");
    hipStream_t st = static_cast<hipStream_t>(s);
    hipLaunchKernelGGL(gp_head_wgrad_stage1_kernel, dim3(GPW_SLICES), dim3(256), 0, st, gg, y, (long long)n * hw, nf, ws);
    int rc = ctgan_check_launch("gp_head_wgrad_stage1");
    if (rc) return rc;
    hipLaunchKernelGGL(gp_head_wgrad_stage2_kernel, dim3((nf + 255) / 256), dim3(256), 0, st, ws, nf, mask_scale / (float)hw, gw, 0);
    return ctgan_check_launch("gp_head_wgrad_stage2");
}
int ctgan_gp_head_wgrad_acc(const float* gg, const float* y, int32_t n, int32_t hw, int32_t nf, float mask_scale, float* gw, float* ws,
                            ctgan_stream_t s) {
    const int c4n = nf >> 2;
    if (!gg || !y || !gw || !ws || n <= 0 || hw <= 0 || nf <= 0 || (nf & 3) || c4n > 256 || (256 % c4n))
        return ctgan_fail(CTGAN_E_BADARG, "gp_head_wgrad_acc: bad argument");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gg) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return ctgan_fail(CTGAN_E_BADARG, "gp_head_wgrad_acc: unaligned");
    hipStream_t st = static_cast<hipStream_t>(s);
    hipLaunchKernelGGL(gp_head_wgrad_stage1_kernel, dim3(GPW_SLICES), dim3(256), 0, st, gg, y, (long long)n * hw, nf, ws);
    int rc = ctgan_check_launch("gp_head_wgrad_stage1");
    if (rc) return rc;
    hipLaunchKernelGGL(gp_head_wgrad_stage2_kernel, dim3((nf + 255) / 256), dim3(256), 0, st, ws, nf, mask_scale / (float)hw, gw, 1);
    return ctgan_check_launch("gp_head_wgrad_stage2");
}
int ctgan_gp_finish(float* ga, const float* gs, const int64_t* gs_strides, int32_t b, int32_t c, int32_t h, int32_t w, float scale, float* slopes,
                    ctgan_stream_t s) {
    if (!ga || !gs || !gs_strides || !slopes || b <= 0 || c <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1))
        return ctgan_fail(CTGAN_E_BADARG, "gp_finish: bad argument");
    hipLaunchKernelGGL(gp_finish_kernel, dim3(b), dim3(256), 0, static_cast<hipStream_t>(s), ga, gs, (long long)gs_strides[0], (long long)gs_strides[1],
                       (long long)gs_strides[2], (long long)gs_strides[3], c, h, w, scale, slopes);
    return ctgan_check_launch("gp_finish");
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
