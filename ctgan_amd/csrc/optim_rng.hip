// optim_rng.hip - TF-form Adam on flat buffers (K21) and Philox4x32-10 random streams (K12/K19).
#include "common.h"
#include "philox.h"

namespace {
using namespace ctgan_philox;

// tf.train.AdamOptimizer:  lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m=b1 m+(1-b1) g; v=b2 v+(1-b2) g^2;
// theta -= lr_t*m/(sqrt(v)+eps)    (eps outside the bias correction, unlike torch.optim.Adam)
// one element of the update; contraction off, so that every kernel that inlines it (plain, step-end, packed; scalar or 4-wide)
// rounds identically - the fused forms are tested bit-for-bit against the separate launches
__device__ __forceinline__ void adam_elem(float& th, float& m, float& v, float graw, float gscale, float b1, float b2, float eps, float lr_t,
                                          float* skipped) {
#pragma clang fp contract(off)
    const float gi = graw * gscale;
    // A gradient element that is not finite (an overflow of the fp16 matrix-core mode under its fixed loss scale; a degenerate input)
    // leaves ITS weight and slots untouched: one inf would otherwise sit in m and v for good and turn theta into NaN - also at a learning
    // rate of 0, as in the warm-up passes of a graph capture (0 * inf).  Finite gradients take the unchanged path.
    // The skip is COUNTED (state[3], a float accumulator: exact up to 2^24 elements, only touched on this branch) so that an overflowing
    // or diverged run does not look healthy: FlatAdam.skipped() / bench.py / the training log read it back.
    if (!(fabsf(gi) <= 3.0e38f)) { atomicAdd(skipped, 1.0f); return; }
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi; v = vi;
    th = th - lr_t * mi / (sqrtf(vi) + eps);
}
__global__ void adam_kernel(float* __restrict__ th, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float* state, float b1, float b2,
                            float eps, float gscale) {
    const float lr = state[0], b1p = state[1], b2p = state[2];
    const float lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        adam_elem(th[i], m[i], v[i], g[i], gscale, b1, b2, eps, lr_t, state + 3);
}
__global__ void adam_advance_kernel(float* state, float b1, float b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { state[1] *= b1; state[2] *= b2; }
}
// End of a step: the beta powers of one optimizer and the Philox step counter advance in ONE one-thread launch (they used to be two
// dependent launches, ~5 us each on the critical path of a captured step).  Doing it inside the update kernel itself - the workgroup
// that finishes last, found with a device-scope fence + atomic per workgroup - was measured: 142 us for the 3456-workgroup update
// (the XCDs' L2s are not coherent: every release fence writes back), against 6.6 us without; removed.
__global__ void step_advance_kernel(float* state, float b1, float b2, uint64_t* ctr, uint64_t by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[1] *= b1; state[2] *= b2;
        if (ctr) ctr[0] += by;
    }
}

// Gather separately allocated gradient tensors into the flat bucket in ONE launch.  The pointer table travels
// by value in the kernel arguments (no device table, no host->device copy => hipGraph-capture safe: the
// arguments are baked into the graph node and the graph's allocations are static).  grid.y = tensor index.
constexpr int PACK_MAX = 64;
struct PackTable { const float* src[PACK_MAX]; long long dst_off[PACK_MAX]; long long n[PACK_MAX]; };
__global__ void pack_kernel(const PackTable t, float* __restrict__ flat) {
    const float* src = t.src[blockIdx.y];
    const long long off = t.dst_off[blockIdx.y], n = t.n[blockIdx.y];
    const long long stride = (long long)gridDim.x * blockDim.x;
    // 16-B copies when source, destination and length allow (the few multi-megabyte filters dominate the bytes)
    if (((n | off) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(flat) & 15) == 0) {
        const long long n4 = n >> 2;
        float4* dst4 = reinterpret_cast<float4*>(flat + off);
        const float4* src4 = reinterpret_cast<const float4*>(src);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
            dst4[i] = src ? src4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        flat[off + i] = src ? src[i] : 0.f;
}

// pack_kernel + adam_kernel in one launch (single-rank steps: nothing happens between the gather and the update).
// The flat bucket is still written - it is the gradient the caller reports and the tests read.  Same per-element arithmetic as
// adam_kernel on the packed bucket (bit-identical results).
__global__ void adam_packed_kernel(const PackTable t, float* __restrict__ flat, float* __restrict__ th, float* __restrict__ m,
                                   float* __restrict__ v, float* state, float b1, float b2, float eps, float gscale) {
    const float lr = state[0], b1p = state[1], b2p = state[2];
    const float lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const float* src = t.src[blockIdx.y];
    const long long off = t.dst_off[blockIdx.y], n = t.n[blockIdx.y];
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (((n | off) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {        // (the flat buffers are 16-B aligned: checked by the host)
        const long long n4 = n >> 2;
        const float4* src4 = reinterpret_cast<const float4*>(src);
        float4* f4 = reinterpret_cast<float4*>(flat + off);
        float4* th4 = reinterpret_cast<float4*>(th + off);
        float4* m4 = reinterpret_cast<float4*>(m + off);
        float4* v4 = reinterpret_cast<float4*>(v + off);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const float4 gr = src ? src4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            f4[i] = gr;
            float4 mm = m4[i], vv = v4[i], tt = th4[i];
            const float gv[4] = {gr.x, gr.y, gr.z, gr.w};
            float* mp = &mm.x; float* vp = &vv.x; float* tp = &tt.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) adam_elem(tp[k], mp[k], vp[k], gv[k], gscale, b1, b2, eps, lr_t, state + 3);
            m4[i] = mm; v4[i] = vv; th4[i] = tt;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const float gr = src ? src[i] : 0.f;
            flat[off + i] = gr;
            adam_elem(th[off + i], m[off + i], v[off + i], gr, gscale, b1, b2, eps, lr_t, state + 3);
        }
    }
}

// ---- Philox4x32-10 (Salmon et al., SC'11); constants of the Random123 reference

__global__ void rng_uniform_kernel(float* __restrict__ out, long long n, uint64_t seed, uint32_t sid,
                                   const uint64_t* __restrict__ ctr, float lo, float hi) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long nblk = (n + 3) >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t c[4];
        draw4(seed, sid, step, (uint32_t)b, c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = b * 4 + k;
            if (i < n) out[i] = lo + (hi - lo) * u01(c[k]);
        }
    }
}
// tf.nn.dropout with the uniform draw generated in the kernel: y = x/keep * floor(keep + u), u = element i of the
// Philox stream (seed, sid, step) - exactly the value rng_uniform_kernel would have written at physical index i.
__global__ void dropout_rng_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float keep, float inv,
                                   uint64_t seed, uint32_t sid, const uint64_t* __restrict__ ctr) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long nblk = (n + 3) >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t c[4];
        draw4(seed, sid, step, (uint32_t)b, c);
        const long long i = b * 4;
        if (vec && i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            float4 o;
            o.x = v.x * inv * floorf(keep + u01(c[0])); o.y = v.y * inv * floorf(keep + u01(c[1]));
            o.z = v.z * inv * floorf(keep + u01(c[2])); o.w = v.w * inv * floorf(keep + u01(c[3]));
            *reinterpret_cast<float4*>(y + i) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n) y[i + k] = x[i + k] * inv * floorf(keep + u01(c[k]));
        }
    }
}
// LeakyReLU and dropout in ONE pass (the DCGAN critics' `dropout(LeakyReLU(conv))`, TF/CT_gan_cifar.py:84-98): y = x * slope(ref) / keep *
// floor(keep + u), slope(r) = r > 0 ? 1 : alpha.  Forward: ref = x (y = dropout(lrelu(x))).  Backward and double backward: x = the arriving
// gradient, ref = the forward RESULT - where the mask kept the value its sign is the pre-activation's, where it dropped it the product is 0.
// (n1, sid2: elements [n1, n) draw stream sid2, indexed from n1 - two tensors' dropouts in one launch, each with the draws a launch of
// its own would make: the hand-scheduled DCGAN critic step keeps the dropout-pass rows and the penalty rows in one tensor.  n1 % 4 == 0.)
__global__ void lrelu_dropout_rng_kernel(const float* __restrict__ x, const float* __restrict__ ref, float* __restrict__ y, long long n,
                                         float alpha, float keep, float inv, uint64_t seed, uint32_t sid, const uint64_t* __restrict__ ctr,
                                         long long n1, uint32_t sid2) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long nblk = (n + 3) >> 2, blk1 = n1 >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ref)) & 15) == 0;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t c[4];
        if (b < blk1) draw4(seed, sid, step, (uint32_t)b, c);
        else draw4(seed, sid2, step, (uint32_t)(b - blk1), c);
        const long long i = b * 4;
        if (vec && i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            const float4 r = *reinterpret_cast<const float4*>(ref + i);
            float4 o;
            o.x = v.x * (r.x > 0.f ? 1.f : alpha) * inv * floorf(keep + u01(c[0])); o.y = v.y * (r.y > 0.f ? 1.f : alpha) * inv * floorf(keep + u01(c[1]));
            o.z = v.z * (r.z > 0.f ? 1.f : alpha) * inv * floorf(keep + u01(c[2])); o.w = v.w * (r.w > 0.f ? 1.f : alpha) * inv * floorf(keep + u01(c[3]));
            *reinterpret_cast<float4*>(y + i) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n) y[i + k] = x[i + k] * (ref[i + k] > 0.f ? 1.f : alpha) * inv * floorf(keep + u01(c[k]));
        }
    }
}
// dropout_rng_kernel followed by the ReLU mask of lrelu_bwd (alpha 0) in one pass: y = dropout(x) (optional), ym = y where ref > 0
// else 0.  The double backward of a data gradient whose result is masked, added to and dropped needs both (functional.ConvDgradFn).
__global__ void dropout_rng_mask_kernel(const float* __restrict__ x, const float* __restrict__ ref, float* __restrict__ y,
                                        float* __restrict__ ym, long long n, float keep, float inv, uint64_t seed, uint32_t sid,
                                        const uint64_t* __restrict__ ctr) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long nblk = (n + 3) >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ym) |
                       reinterpret_cast<uintptr_t>(ref)) & 15) == 0;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t c[4];
        draw4(seed, sid, step, (uint32_t)b, c);
        const long long i = b * 4;
        if (vec && i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            const float4 r = *reinterpret_cast<const float4*>(ref + i);
            float4 o;
            o.x = v.x * inv * floorf(keep + u01(c[0])); o.y = v.y * inv * floorf(keep + u01(c[1]));
            o.z = v.z * inv * floorf(keep + u01(c[2])); o.w = v.w * inv * floorf(keep + u01(c[3]));
            if (y) *reinterpret_cast<float4*>(y + i) = o;
            o.x = r.x > 0.f ? o.x : 0.f; o.y = r.y > 0.f ? o.y : 0.f; o.z = r.z > 0.f ? o.z : 0.f; o.w = r.w > 0.f ? o.w : 0.f;
            *reinterpret_cast<float4*>(ym + i) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n) {
                    const float o = x[i + k] * inv * floorf(keep + u01(c[k]));
                    if (y) y[i + k] = o;
                    ym[i + k] = ref[i + k] > 0.f ? o : 0.f;
                }
        }
    }
}
__global__ void rng_normal_kernel(float* __restrict__ out, long long n, uint64_t seed, uint32_t sid,
                                  const uint64_t* __restrict__ ctr) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long nblk = (n + 3) >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t c[4];
        draw4(seed, sid, step, (uint32_t)b, c);
        float z[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {   // Box-Muller on (c[2k], c[2k+1])
            const float u1 = ((float)(c[2 * k] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
            const float u2 = u01(c[2 * k + 1]);
            const float r = sqrtf(-2.f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * k] = r * cs; z[2 * k + 1] = r * sn;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = b * 4 + k;
            if (i < n) out[i] = z[k];
        }
    }
}
__global__ void rng_labels_kernel(int32_t* __restrict__ out, long long n, int nlab, uint64_t seed, uint32_t sid,
                                  const uint64_t* __restrict__ ctr) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long nblk = (n + 3) >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t c[4];
        draw4(seed, sid, step, (uint32_t)b, c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = b * 4 + k;
            if (i < n) out[i] = (int32_t)(u01(c[k]) * (float)nlab);   // tf.cast(float->int32) truncates
        }
    }
}
// ---- critic-step input preparation in one launch (TF/CT_gan_cifar_resnet.py:201-202,226,277-283) ---------------------------
//   real   = 2*(x/denom - .5) + U[lo,hi)          (dequantisation noise = element i of stream sid_deq)
//   interp = real + alpha*(fake - real)            (alpha[row] = element row of stream sid_alpha, U[0,1))
//   rf     = [real ; fake]                         (the batch of the two dropout passes)
// bit-identical draws to rng_uniform_kernel on [b,d] / [b,1] tensors followed by real_prep / interpolate / concat.  d % 4 == 0.
__global__ void critic_prep_kernel(const int32_t* __restrict__ xi, const float* __restrict__ fake, long long n4, int d, uint64_t seed,
                                   uint32_t sid_deq, uint32_t sid_alpha, const uint64_t* __restrict__ ctr, float lo, float hi, float denom,
                                   float* __restrict__ rf, float* __restrict__ interp) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4) return;
    const long long i = q * 4;
    const int row = (int)(i / d);
    uint32_t c[4], ca[4];
    draw4(seed, sid_deq, step, (uint32_t)q, c);
    draw4(seed, sid_alpha, step, (uint32_t)(row >> 2), ca);
    const float alpha = 0.f + (1.f - 0.f) * u01(ca[row & 3]);
    const int4 xv = *reinterpret_cast<const int4*>(xi + i);
    const float4 fv = *reinterpret_cast<const float4*>(fake + i);
    float4 r, o;
    r.x = 2.f * (((float)xv.x / denom) - .5f); r.x += lo + (hi - lo) * u01(c[0]);
    r.y = 2.f * (((float)xv.y / denom) - .5f); r.y += lo + (hi - lo) * u01(c[1]);
    r.z = 2.f * (((float)xv.z / denom) - .5f); r.z += lo + (hi - lo) * u01(c[2]);
    r.w = 2.f * (((float)xv.w / denom) - .5f); r.w += lo + (hi - lo) * u01(c[3]);
    o.x = r.x + alpha * (fv.x - r.x); o.y = r.y + alpha * (fv.y - r.y);
    o.z = r.z + alpha * (fv.z - r.z); o.w = r.w + alpha * (fv.w - r.w);
    *reinterpret_cast<float4*>(rf + i) = r;
    *reinterpret_cast<float4*>(rf + n4 * 4 + i) = fv;
    *reinterpret_cast<float4*>(interp + i) = o;
}

// ---- [x ; x[0:n_extra]] with tf.nn.dropout on the result, one launch (the input of the critic tail for the two dropout passes,
// pass 2 on the real half only, :226-227,288-291): dst row r < n_src reads src row r, row n_src + r' reads src row r'.
// Dropout draws = element index of dst in stream sid (as dropout_rng_kernel on the concatenated tensor); keep >= 1: plain concat.
__global__ void rows_cat_dropout_kernel(const float* __restrict__ src, long long row4, long long n4_src, long long n4_dst, float keep,
                                        float inv, uint64_t seed, uint32_t sid, const uint64_t* __restrict__ ctr, float* __restrict__ dst) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4_dst) return;
    const long long sq = q < n4_src ? q : q - n4_src;
    float4 v = *reinterpret_cast<const float4*>(src + sq * 4);
    if (keep < 1.f) {
        uint32_t c[4];
        draw4(seed, sid, step, (uint32_t)q, c);
        v.x = v.x * inv * floorf(keep + u01(c[0])); v.y = v.y * inv * floorf(keep + u01(c[1]));
        v.z = v.z * inv * floorf(keep + u01(c[2])); v.w = v.w * inv * floorf(keep + u01(c[3]));
    }
    *reinterpret_cast<float4*>(dst + q * 4) = v;
    (void)row4;
}
// General form: dst = concatenation of row segments of src, each with its own dropout (or none); the Philox element index
// of a dst element is taken relative to dst row `index_row0` of its segment - so a group of segments can reproduce exactly the
// dropout of "its own" concatenated tensor.  One launch builds the input of every critic tail of a step.
struct RowSegs { int n; long long dst_end4[CTGAN_ROW_SEGMENTS]; long long src_off4[CTGAN_ROW_SEGMENTS]; long long idx_off4[CTGAN_ROW_SEGMENTS];
                 float keep[CTGAN_ROW_SEGMENTS]; unsigned sid[CTGAN_ROW_SEGMENTS]; };
__global__ void rows_gather_dropout_kernel(const float* __restrict__ src, const RowSegs t, long long n4_dst, uint64_t seed,
                                           const uint64_t* __restrict__ ctr, float* __restrict__ dst) {
    const uint64_t step = ctr ? ctr[0] : 0;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4_dst) return;
    int sg = 0;
#pragma unroll
    for (int i = 1; i < CTGAN_ROW_SEGMENTS; ++i)
        if (i < t.n && q >= t.dst_end4[i - 1]) sg = i;
    const long long dst0 = sg ? t.dst_end4[sg - 1] : 0;
    float4 v = *reinterpret_cast<const float4*>(src + (t.src_off4[sg] + (q - dst0)) * 4);
    const float keep = t.keep[sg];
    if (keep < 1.f) {
        uint32_t c[4];
        draw4(seed, t.sid[sg], step, (uint32_t)(q - t.idx_off4[sg]), c);
        const float inv = 1.f / keep;
        v.x = v.x * inv * floorf(keep + u01(c[0])); v.y = v.y * inv * floorf(keep + u01(c[1]));
        v.z = v.z * inv * floorf(keep + u01(c[2])); v.w = v.w * inv * floorf(keep + u01(c[3]));
    }
    *reinterpret_cast<float4*>(dst + q * 4) = v;
}
// adjoint of the concat: gsrc[r] = g[r] + (r < n_extra ? g[n_src + r] : 0)
// the same with n_pass rows behind the concat that pass straight through: g = [a (n_src) ; a' (n_extra) ; c (n_pass)] -> gsrc = [a + a' ; c]
// (the merged backward of a critic step: rows [real, fake | real' | x_hat] of the tail -> rows [real, fake, x_hat] of the trunk)
__global__ void rows_cat_bwd2_kernel(const float* __restrict__ g, long long n4_src, long long n4_extra, long long n4_pass, float* __restrict__ gsrc) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4_src + n4_pass) return;
    float4 v;
    if (q < n4_src) {
        v = *reinterpret_cast<const float4*>(g + q * 4);
        if (q < n4_extra) {
            const float4 w = *reinterpret_cast<const float4*>(g + (n4_src + q) * 4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
    } else {
        v = *reinterpret_cast<const float4*>(g + (n4_extra + q) * 4);
    }
    *reinterpret_cast<float4*>(gsrc + q * 4) = v;
}
__global__ void rows_cat_bwd_kernel(const float* __restrict__ g, long long n4_src, long long n4_extra, float* __restrict__ gsrc) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4_src) return;
    float4 v = *reinterpret_cast<const float4*>(g + q * 4);
    if (q < n4_extra) {
        const float4 w = *reinterpret_cast<const float4*>(g + (n4_src + q) * 4);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    *reinterpret_cast<float4*>(gsrc + q * 4) = v;
}

__global__ void rng_advance_kernel(uint64_t* ctr, uint64_t by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += by;
}

}  // namespace

extern "C" {

int ctgan_adam_step(float* theta, const float* g, float* m, float* v, int64_t n, float* state, float beta1,
                    float beta2, float eps, float grad_scale, ctgan_stream_t s) {
    if (!theta || !g || !m || !v || !state || n < 0) return ctgan_fail(CTGAN_E_BADARG, "adam_step: bad argument");
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(ctgan_blocks(n, 256, 2048)), dim3(256), 0, static_cast<hipStream_t>(s), theta, g, m,
                       v, (long long)n, state, beta1, beta2, eps, grad_scale);
    return ctgan_check_launch("adam_step");
}
int ctgan_pack(const float* const* srcs, const int64_t* dst_offs, const int64_t* counts, int32_t n_tensors, float* flat,
               ctgan_stream_t s) {
    if (!srcs || !dst_offs || !counts || !flat || n_tensors <= 0) return ctgan_fail(CTGAN_E_BADARG, "pack: bad argument");
    for (int base = 0; base < n_tensors; base += PACK_MAX) {
        PackTable t;
        const int cnt = n_tensors - base < PACK_MAX ? n_tensors - base : PACK_MAX;
        long long mx = 1;
        for (int i = 0; i < cnt; ++i) {
            t.src[i] = srcs[base + i]; t.dst_off[i] = dst_offs[base + i]; t.n[i] = counts[base + i];
            if (t.n[i] > mx) mx = t.n[i];
        }
        hipLaunchKernelGGL(pack_kernel, dim3(ctgan_blocks(mx, 1024, 512), cnt), dim3(256), 0, static_cast<hipStream_t>(s), t, flat);
        int rc = ctgan_check_launch("pack");
        if (rc) return rc;
    }
    return CTGAN_OK;
}
int ctgan_step_advance(float* state, float beta1, float beta2, uint64_t* rng_ctr, uint64_t rng_by, ctgan_stream_t s) {
    if (!state) return ctgan_fail(CTGAN_E_BADARG, "step_advance: null");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(s), state, beta1, beta2, rng_ctr, rng_by);
    return ctgan_check_launch("step_advance");
}
int ctgan_adam_step_packed(const float* const* srcs, const int64_t* dst_offs, const int64_t* counts, int32_t n_tensors, float* flat,
                           float* theta, float* m, float* v, float* state, float beta1, float beta2, float eps, float grad_scale,
                           ctgan_stream_t s) {
    if (!srcs || !dst_offs || !counts || !flat || !theta || !m || !v || !state || n_tensors <= 0)
        return ctgan_fail(CTGAN_E_BADARG, "adam_step_packed: bad argument");
    if (n_tensors > PACK_MAX) return ctgan_fail(CTGAN_E_UNSUPPORTED, "adam_step_packed: more than %d tensors", PACK_MAX);
    if ((reinterpret_cast<uintptr_t>(flat) | reinterpret_cast<uintptr_t>(theta) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return ctgan_fail(CTGAN_E_UNSUPPORTED, "adam_step_packed: flat buffers must be 16-byte aligned");
    PackTable t;
    long long mx = 1;
    for (int i = 0; i < n_tensors; ++i) {
        if (counts[i] < 0 || dst_offs[i] < 0) return ctgan_fail(CTGAN_E_BADARG, "adam_step_packed: negative extent");
        t.src[i] = srcs[i]; t.dst_off[i] = dst_offs[i]; t.n[i] = counts[i];
        if (t.n[i] > mx) mx = t.n[i];
    }
    hipLaunchKernelGGL(adam_packed_kernel, dim3(ctgan_blocks(mx, 1024, 512), n_tensors), dim3(256), 0, static_cast<hipStream_t>(s), t,
                       flat, theta, m, v, state, beta1, beta2, eps, grad_scale);
    return ctgan_check_launch("adam_step_packed");
}
int ctgan_adam_advance(float* state, float beta1, float beta2, ctgan_stream_t s) {
    if (!state) return ctgan_fail(CTGAN_E_BADARG, "adam_advance: null");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(s), state, beta1, beta2);
    return ctgan_check_launch("adam_advance");
}

int ctgan_rng_uniform(float* out, int64_t n, uint64_t seed, uint64_t stream_id, const uint64_t* ctr, float lo, float hi,
                      ctgan_stream_t s) {
    if (!out || n < 0 || n >= (1LL << 34)) return ctgan_fail(CTGAN_E_BADARG, "rng_uniform: bad argument");
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(rng_uniform_kernel, dim3(ctgan_blocks((n + 3) / 4, 256, 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(s), out, (long long)n, seed, (uint32_t)stream_id, ctr, lo, hi);
    return ctgan_check_launch("rng_uniform");
}
int ctgan_dropout_rng(const float* x, float* y, int64_t n, float keep, uint64_t seed, uint64_t stream_id, const uint64_t* ctr,
                      ctgan_stream_t s) {
    if (!x || !y || n < 0 || n >= (1LL << 34)) return ctgan_fail(CTGAN_E_BADARG, "dropout_rng: bad argument");
    if (!(keep > 0.f) || keep > 1.f) return ctgan_fail(CTGAN_E_BADARG, "dropout_rng: keep=%g not in (0,1]", keep);
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(dropout_rng_kernel, dim3(ctgan_blocks((n + 3) / 4, 256, 2048)), dim3(256), 0, static_cast<hipStream_t>(s), x, y,
                       (long long)n, keep, 1.f / keep, seed, (uint32_t)stream_id, ctr);
    return ctgan_check_launch("dropout_rng");
}
int ctgan_lrelu_dropout_rng(const float* x, const float* ref, float* y, int64_t n, float alpha, float keep, uint64_t seed, uint64_t stream_id,
                            const uint64_t* ctr, ctgan_stream_t s) {
    if (!x || !ref || !y || n < 0 || n >= (1LL << 34)) return ctgan_fail(CTGAN_E_BADARG, "lrelu_dropout_rng: bad argument");
    if (!(keep > 0.f) || keep > 1.f) return ctgan_fail(CTGAN_E_BADARG, "lrelu_dropout_rng: keep=%g not in (0,1]", keep);
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(lrelu_dropout_rng_kernel, dim3(ctgan_blocks((n + 3) / 4, 256, 2048)), dim3(256), 0, static_cast<hipStream_t>(s), x, ref, y,
                       (long long)n, alpha, keep, 1.f / keep, seed, (uint32_t)stream_id, ctr, ((long long)n + 3) & ~3LL, 0u);      // one stream: every block
    return ctgan_check_launch("lrelu_dropout_rng");
}
int ctgan_lrelu_dropout_rng2(const float* x, const float* ref, float* y, int64_t n, int64_t n1, float alpha, float keep, uint64_t seed,
                             uint64_t stream_id, uint64_t stream_id2, const uint64_t* ctr, ctgan_stream_t s) {
    if (!x || !ref || !y || n < 0 || n >= (1LL << 34) || n1 < 0 || n1 > n || (n1 & 3)) return ctgan_fail(CTGAN_E_BADARG, "lrelu_dropout_rng2: bad argument");
    if (!(keep > 0.f) || keep > 1.f) return ctgan_fail(CTGAN_E_BADARG, "lrelu_dropout_rng2: keep=%g not in (0,1]", keep);
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(lrelu_dropout_rng_kernel, dim3(ctgan_blocks((n + 3) / 4, 256, 2048)), dim3(256), 0, static_cast<hipStream_t>(s), x, ref, y,
                       (long long)n, alpha, keep, 1.f / keep, seed, (uint32_t)stream_id, ctr, (long long)n1, (uint32_t)stream_id2);
    return ctgan_check_launch("lrelu_dropout_rng2");
}
int ctgan_dropout_rng_mask(const float* x, const float* ref, float* y, float* y_masked, int64_t n, float keep, uint64_t seed,
                           uint64_t stream_id, const uint64_t* ctr, ctgan_stream_t s) {
    if (!x || !ref || !y_masked || n < 0 || n >= (1LL << 34)) return ctgan_fail(CTGAN_E_BADARG, "dropout_rng_mask: bad argument");
    if (!(keep > 0.f) || keep > 1.f) return ctgan_fail(CTGAN_E_BADARG, "dropout_rng_mask: keep=%g not in (0,1]", keep);
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(dropout_rng_mask_kernel, dim3(ctgan_blocks((n + 3) / 4, 256, 2048)), dim3(256), 0, static_cast<hipStream_t>(s), x, ref,
                       y, y_masked, (long long)n, keep, 1.f / keep, seed, (uint32_t)stream_id, ctr);
    return ctgan_check_launch("dropout_rng_mask");
}
int ctgan_rng_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, const uint64_t* ctr, ctgan_stream_t s) {
    if (!out || n < 0 || n >= (1LL << 34)) return ctgan_fail(CTGAN_E_BADARG, "rng_normal: bad argument");
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(rng_normal_kernel, dim3(ctgan_blocks((n + 3) / 4, 256, 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(s), out, (long long)n, seed, (uint32_t)stream_id, ctr);
    return ctgan_check_launch("rng_normal");
}
int ctgan_rng_labels(int32_t* out, int64_t n, int32_t nlab, uint64_t seed, uint64_t stream_id, const uint64_t* ctr,
                     ctgan_stream_t s) {
    if (!out || n < 0 || nlab <= 0) return ctgan_fail(CTGAN_E_BADARG, "rng_labels: bad argument");
    if (n == 0) return CTGAN_OK;
    hipLaunchKernelGGL(rng_labels_kernel, dim3(ctgan_blocks((n + 3) / 4, 256, 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(s), out, (long long)n, nlab, seed, (uint32_t)stream_id, ctr);
    return ctgan_check_launch("rng_labels");
}
int ctgan_critic_prep(const int32_t* x_int, const float* fake, int32_t b, int32_t d, uint64_t seed, uint64_t sid_deq, uint64_t sid_alpha,
                      const uint64_t* ctr, float lo, float hi, float denom, float* rf, float* interp, ctgan_stream_t s) {
    if (!x_int || !fake || !rf || !interp || b <= 0 || d <= 0 || (d & 3) ||
        ((reinterpret_cast<uintptr_t>(x_int) | reinterpret_cast<uintptr_t>(fake) | reinterpret_cast<uintptr_t>(rf) | reinterpret_cast<uintptr_t>(interp)) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "critic_prep: bad argument");
    const long long n4 = (long long)b * d / 4;
    hipLaunchKernelGGL(critic_prep_kernel, dim3(ctgan_blocks(n4, 256, 1 << 20)), dim3(256), 0, static_cast<hipStream_t>(s), x_int, fake, n4,
                       d, seed, (uint32_t)sid_deq, (uint32_t)sid_alpha, ctr, lo, hi, denom, rf, interp);
    return ctgan_check_launch("critic_prep");
}
int ctgan_rows_cat_dropout(const float* src, int64_t n_src, int64_t n_extra, int64_t row_elems, float keep, uint64_t seed,
                           uint64_t stream_id, const uint64_t* ctr, float* dst, ctgan_stream_t s) {
    if (!src || !dst || n_src <= 0 || n_extra < 0 || n_extra > n_src || row_elems <= 0 || (row_elems & 3) || !(keep > 0.f) ||
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) || (n_src + n_extra) * row_elems >= (1LL << 34))
        return ctgan_fail(CTGAN_E_BADARG, "rows_cat_dropout: bad argument");
    const long long n4_src = n_src * row_elems / 4, n4_dst = (n_src + n_extra) * row_elems / 4;
    hipLaunchKernelGGL(rows_cat_dropout_kernel, dim3(ctgan_blocks(n4_dst, 256, 1 << 20)), dim3(256), 0, static_cast<hipStream_t>(s), src,
                       (long long)row_elems / 4, n4_src, n4_dst, keep, 1.f / keep, seed, (uint32_t)stream_id, ctr, dst);
    return ctgan_check_launch("rows_cat_dropout");
}
int ctgan_rows_gather_dropout(const float* src, const ctgan_row_segment* segs, int32_t nseg, int64_t row_elems, uint64_t seed,
                              const uint64_t* ctr, float* dst, ctgan_stream_t s) {
    if (!src || !segs || !dst || nseg < 1 || nseg > CTGAN_ROW_SEGMENTS || row_elems <= 0 || (row_elems & 3) ||
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "rows_gather_dropout: bad argument");
    RowSegs t;
    t.n = nseg;
    const long long r4 = row_elems / 4;
    long long row = 0;
    for (int i = 0; i < CTGAN_ROW_SEGMENTS; ++i) {
        if (i < nseg) {
            const ctgan_row_segment& g = segs[i];
            if (g.rows <= 0 || g.src_row0 < 0 || g.index_row0 < 0 || g.index_row0 > row || !(g.keep > 0.f))
                return ctgan_fail(CTGAN_E_BADARG, "rows_gather_dropout: bad segment %d", i);
            t.src_off4[i] = g.src_row0 * r4; t.idx_off4[i] = g.index_row0 * r4;
            t.keep[i] = g.keep < 1.f ? g.keep : 1.f; t.sid[i] = (unsigned)g.stream_id;
            row += g.rows;
        } else {
            t.src_off4[i] = 0; t.idx_off4[i] = 0; t.keep[i] = 1.f; t.sid[i] = 0;
        }
        t.dst_end4[i] = row * r4;
    }
    const long long n4 = row * r4;
    if (n4 * 4 >= (1LL << 34)) return ctgan_fail(CTGAN_E_BADARG, "rows_gather_dropout: too large");
    hipLaunchKernelGGL(rows_gather_dropout_kernel, dim3(ctgan_blocks(n4, 256, 1 << 20)), dim3(256), 0, static_cast<hipStream_t>(s), src, t, n4,
                       seed, ctr, dst);
    return ctgan_check_launch("rows_gather_dropout");
}
int ctgan_rows_cat_bwd2(const float* g, int64_t n_src, int64_t n_extra, int64_t n_pass, int64_t row_elems, float* gsrc, ctgan_stream_t s) {
    if (!g || !gsrc || n_src <= 0 || n_extra < 0 || n_extra > n_src || n_pass < 0 || row_elems <= 0 || (row_elems & 3) ||
        ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gsrc)) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "rows_cat_bwd2: bad argument");
    const long long n4_src = n_src * row_elems / 4, n4_extra = n_extra * row_elems / 4, n4_pass = n_pass * row_elems / 4;
    hipLaunchKernelGGL(rows_cat_bwd2_kernel, dim3(ctgan_blocks(n4_src + n4_pass, 256, 1 << 20)), dim3(256), 0, static_cast<hipStream_t>(s), g,
                       n4_src, n4_extra, n4_pass, gsrc);
    return ctgan_check_launch("rows_cat_bwd2");
}
int ctgan_rows_cat_bwd(const float* g, int64_t n_src, int64_t n_extra, int64_t row_elems, float* gsrc, ctgan_stream_t s) {
    if (!g || !gsrc || n_src <= 0 || n_extra < 0 || n_extra > n_src || row_elems <= 0 || (row_elems & 3) ||
        ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gsrc)) & 15))
        return ctgan_fail(CTGAN_E_BADARG, "rows_cat_bwd: bad argument");
    const long long n4_src = n_src * row_elems / 4, n4_extra = n_extra * row_elems / 4;
    hipLaunchKernelGGL(rows_cat_bwd_kernel, dim3(ctgan_blocks(n4_src, 256, 1 << 20)), dim3(256), 0, static_cast<hipStream_t>(s), g, n4_src,
                       n4_extra, gsrc);
    return ctgan_check_launch("rows_cat_bwd");
}
int ctgan_rng_advance(uint64_t* ctr, uint64_t by, ctgan_stream_t s) {
    if (!ctr) return ctgan_fail(CTGAN_E_BADARG, "rng_advance: null");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(s), ctr, by);
    return ctgan_check_launch("rng_advance");
}

}  // extern "C"
