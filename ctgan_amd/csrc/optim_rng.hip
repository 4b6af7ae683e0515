// optim_rng.hip - TF-form Adam on flat buffers (K21) and Philox4x32-10 random streams (K12/K19).
#include "common.h"
#include "philox.h"

namespace {
using namespace ctgan_philox;

// tf.train.AdamOptimizer:  lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m=b1 m+(1-b1) g; v=b2 v+(1-b2) g^2;
// theta -= lr_t*m/(sqrt(v)+eps)    (eps outside the bias correction, unlike torch.optim.Adam)
__global__ void adam_kernel(float* __restrict__ th, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, const float* __restrict__ state, float b1, float b2,
                            float eps, float gscale) {
    const float lr = state[0], b1p = state[1], b2p = state[2];
    const float lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        th[i] = th[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}
__global__ void adam_advance_kernel(float* state, float b1, float b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { state[1] *= b1; state[2] *= b2; }
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
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        flat[off + i] = src ? src[i] : 0.f;
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
__global__ void rng_advance_kernel(uint64_t* ctr, uint64_t by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += by;
}

}  // namespace

extern "C" {

int ctgan_adam_step(float* theta, const float* g, float* m, float* v, int64_t n, const float* state, float beta1,
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
        hipLaunchKernelGGL(pack_kernel, dim3(ctgan_blocks(mx, 256, 64), cnt), dim3(256), 0, static_cast<hipStream_t>(s), t, flat);
        int rc = ctgan_check_launch("pack");
        if (rc) return rc;
    }
    return CTGAN_OK;
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
int ctgan_rng_advance(uint64_t* ctr, uint64_t by, ctgan_stream_t s) {
    if (!ctr) return ctgan_fail(CTGAN_E_BADARG, "rng_advance: null");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(s), ctr, by);
    return ctgan_check_launch("rng_advance");
}

}  // extern "C"
