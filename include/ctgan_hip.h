/* ctgan_hip.h - C-ABI of libctgan_hip.so: the MI355X (gfx950) kernels behind the CT-WGAN
 * adversarial step of biuyq/CT-GAN.
 *
 * The reference has no native code and no FFI: its operator library (the tflib/ops modules) bottoms out
 * in TensorFlow-1.2.1 graph nodes.  Each entry point below replaces one of those TF call sites
 * (cited per function; TF/ = CT-GANs/tensorflow_generative_model/).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - All tensors are fp32 unless stated; activations are logically NCHW like the reference but
 *    described with explicit element strides, so the physical layout may be NHWC (channels-last,
 *    what the fast kernels want) or NCHW (image inputs / sample outputs).
 *  - The library never allocates or frees caller-visible memory; workspaces are caller-provided
 *    (size from ctgan_conv2d_workspace_bytes).
 *  - Every call is asynchronous on the caller's hipStream_t (`stream`, passed as void*); no
 *    internal device synchronisation; safe under hipGraph stream capture.
 *  - Return 0 on success, negative CTGAN_E_* on failure; ctgan_last_error() gives a thread-local
 *    message.  No C++ exception crosses the ABI.
 */
#ifndef CTGAN_HIP_H
#define CTGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTGAN_OK 0
#define CTGAN_E_BADARG (-1)
#define CTGAN_E_UNSUPPORTED (-2)
#define CTGAN_E_LAUNCH (-3)

#define CTGAN_ABI_VERSION 1

typedef void* ctgan_stream_t; /* hipStream_t */

/* Geometry of one TF 'SAME' convolution  y[n,k,p,q] = sum_{r,s,c} x[n,c,p*stride-pad_t+r,
 * q*stride-pad_l+s] * w[r,s,c,k]  (w is HWIO, contiguous: tflib `name.Filters`).
 * The same descriptor drives the forward op, its data gradient and its weight gradient.
 * A Deconv2D is described by the descriptor of the strided conv it is the adjoint of
 * (x side = the large / output side of the deconv). */
typedef struct ctgan_conv_desc {
    int32_t N, C, H, W;   /* x: batch, channels, height, width                                */
    int32_t K, R, S;      /* y channels, filter height, filter width                           */
    int32_t P, Q;         /* y height, width ( = ceil(H/stride), ceil(W/stride) under SAME )   */
    int32_t stride;       /* 1 or 2                                                            */
    int32_t pad_t, pad_l; /* leading SAME pads (trailing pads are implied by P,Q)              */
    int32_t x_up;         /* 1: x is read through a nearest-neighbour 2x upsample (H,W are the */
                          /*    upsampled sizes; the physical tensor is H/2 x W/2)  (K10)      */
    int32_t reserved;
    int64_t xs[4];        /* x element strides for (n, c, h, w)                                */
    int64_t ys[4];        /* y element strides for (n, k, p, q)                                */
} ctgan_conv_desc;

enum { CTGAN_CONV_FWD = 0, CTGAN_CONV_DGRAD = 1, CTGAN_CONV_WGRAD = 2 };
enum { CTGAN_EPI_RELU = 1, CTGAN_IN_RELU = 2,    /* relu on the result / on the gathered input (conv(relu(x))) */
       CTGAN_RESID_UP = 8 };                      /* fwd: `resid` is the dense channels-last [N,K,P/2,Q/2] tensor and is added
                                                    * through a nearest-2x upsample (UpsampleConv shortcut, :100-107,130) */
enum { CTGAN_DGRAD_W_REPACKED = 1 };

/* ---- library ------------------------------------------------------------------------------ */
int ctgan_version(void);
const char* ctgan_last_error(void);
/* which kernel variant the last conv call on this thread dispatched to (for tests/profiles)   */
const char* ctgan_last_kernel(void);
/* ... and its device symbol as rocprofv3 prints it (e.g. "conv16_kernel<3, 2, 2, 32, false, false>"); the variant name when the
   launcher does not record one.  bench.py keys its per-kernel roofline table by it (cross-checked against profiles/).            */
const char* ctgan_last_symbol(void);
/* (test / A-B switches and launch introspection live in ctgan_hip_debug.h - not part of this ABI)                              */

/* Optional epilogue extension of ctgan_conv2d_fwd / ctgan_conv2d_dgrad: tf.nn.dropout (:173-177) applied to the
 * RESULT inside the kernel, y *= floor(keep + u)/keep, where u is what ctgan_rng_uniform(.., seed, stream_id, ctr)
 * writes at the same physical offset (i.e. exactly ctgan_dropout_rng of the result, without the extra pass).  The
 * forward uses it for a dropout that follows a conv; the backward for the dropout mask a data gradient is multiplied
 * with.  drop_keep outside (0,1) = no dropout.  CTGAN_E_UNSUPPORTED unless the pipelined kernel with the 16-B
 * epilogue serves the call (caller then applies ctgan_dropout_rng itself).
 * Row ranges (forward only, dense channels-last result): n_ranges > 0 splits the batch into consecutive sample ranges
 * [0, range_end[0]), [range_end[0], range_end[1]), ... each with its own keep probability (outside (0,1) = none) and
 * stream id, the draws indexed from the range's first element - exactly the dropout each range would get as a tensor of
 * its own.  Lets several passes that use the same weights (dropout passes, clean pass, gradient-penalty pass) share one
 * launch per layer.                                                                                                      */
#define CTGAN_DROP_RANGES 3
typedef struct ctgan_epilogue_ext {
    float drop_keep;
    uint64_t drop_seed, drop_stream_id;
    const uint64_t* drop_ctr;
    int32_t n_ranges;
    int32_t range_end[CTGAN_DROP_RANGES];
    float range_keep[CTGAN_DROP_RANGES];
    uint64_t range_stream_id[CTGAN_DROP_RANGES];
    /* forward only (ctgan_conv2d_fwd_ex, ctgan_conv2d16_fwd_ex): the result is kept only where out_mask > 0 (strides of y; after the
     * bias, before resid - the order of the dgrad epilogue's mask).  Used by the double backward of the gradient penalty, where the
     * conv that follows multiplies its input with a constant ReLU mask (TF/CT_gan_cifar_resnet.py:284-286 through :109-141).  NULL = none. */
    const float* out_mask;
    /* 16-bit slice kernels only (ctgan_conv2d16_fwd_ex / ctgan_conv2d16_dgrad_ex, mma = CTGAN_MMA_BF16 / CTGAN_MMA_F16, dense channels-last
     * result): act = 1 applies the LeakyReLU + dropout pair of the DCGAN critics (TF/CT_gan_cifar.py:86-98, TF/CT_gan_mnist.py:94-106:
     * tf.nn.dropout(LeakyReLU(conv))) to the result r after bias / mask / resid / relu:
     *     out = r * (ref > 0 ? 1 : act_alpha) * floor(keep + u) / keep,   ref = act_ref ? act_ref[same offset] : r
     * with keep / Philox stream from drop_keep / drop_stream_id or, with n_ranges > 0, from the sample range of the row (draws indexed
     * from the range's first element) - bit for bit ctgan_lrelu_dropout_rng(2) run on the plain result.  act_ref = the forward result
     * gives the pair's backward (the data gradient's epilogue) and its application to a cotangent (the penalty's double backward).      */
    int32_t act;
    float act_alpha;
    const float* act_ref;
    /* ctgan_conv2d_fwd_ex, many -> few convs on the one-pixel-per-lane kernel only (3x3, <= 4 output channels, 32-pixel rows; the generator's
     * output stage tanh(Conv2D(relu(Batchnorm(h)))), TF/CT_gan_cifar_resnet.py:164-166, evaluated without a tape): training-mode batch norm of
     * the INPUT applied while it is staged - x' = (x - in_bn_mean[g][c]) * in_bn_rstd[g][c] * in_bn_scale[c] + in_bn_offset[c], g = sample /
     * (N / in_bn_groups), then CTGAN_IN_RELU if set; the SAME zero padding stays zero - and out_tanh: tanh of the result.  The arithmetic of
     * ctgan_bn_apply followed by the conv and ctgan_tanh_fwd.  in_bn_mean == NULL and out_tanh == 0: off.                                   */
    const float* in_bn_mean;
    const float* in_bn_rstd;
    const float* in_bn_scale;
    const float* in_bn_offset;
    int32_t in_bn_groups;
    int32_t out_tanh;
    /* ... and on ctgan_conv2d16_fwd_ex (mma = CTGAN_MMA_F32X3, stride 1, launches the fragment-streaming halo kernel takes with tiles inside one image,
     * CTGAN_IN_RELU set): the same batch norm while the halo patch is staged - the generator's Conv2 layers (TF/CT_gan_cifar_resnet.py:134-141) without
     * a tape.  in_bn_labels (int32 per sample, or NULL): conditional batch norm - in_bn_scale / in_bn_offset are [n_labels][C] tables.  */
    const int32_t* in_bn_labels;
} ctgan_epilogue_ext;
int ctgan_conv2d_fwd_ex(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias, const float* resid,
                        float* y, int flags, const ctgan_epilogue_ext* ext, ctgan_stream_t stream);
int ctgan_conv2d_dgrad_ex(const ctgan_conv_desc* d, const float* dy, const float* w, const float* bias,
                          const float* mask, const float* resid, float* dx, void* ws, size_t ws_bytes, int flags,
                          const ctgan_epilogue_ext* ext, ctgan_stream_t stream);
/* Weight gradient of ONE filter over several (x, dy) pairs of the same geometry and strides (the uses of the filter in
 * different passes of a step: the dropout passes and the gradient-penalty double backward,
 * TF/CT_gan_cifar_resnet.py:284,335-336 sum them in tf.gradients): dw = sum_s wgrad(xs[s], dys[s]) in one launch + one
 * fixed-order reduction.  d->N is ignored (Ns[s] rows per segment); seg_flags[s] = CTGAN_IN_RELU (x -> relu(x)) |
 * CTGAN_WGRAD_SEG_BIAS (dys[s] contributes to db).  Returns CTGAN_E_UNSUPPORTED for shapes outside the pipelined
 * kernel (caller falls back to ctgan_conv2d_wgrad per segment).                                               */
#define CTGAN_WGRAD_MAX_SEGS 3
#define CTGAN_WGRAD_SEG_BIAS 4
size_t ctgan_conv2d_wgrad_multi_workspace_bytes(const ctgan_conv_desc* d, int32_t nseg, const int32_t* Ns);
int ctgan_conv2d_wgrad_multi(const ctgan_conv_desc* d, int32_t nseg, const float* const* xs, const float* const* dys,
                             const int32_t* Ns, const int32_t* seg_flags, float* dw, float* db, void* ws,
                             size_t ws_bytes, ctgan_stream_t stream);
/* Several such weight gradients (different filters / geometries) at once: the deferred weight gradients of a step are
 * independent of each other, so they share ONE launch per tile configuration (workgroup -> problem table in the kernel
 * arguments, at most CTGAN_WGRAD_GROUP_MAX problems per launch) and ONE launch for all their split-K reductions.
 * Validates every group before the first launch: CTGAN_E_UNSUPPORTED means nothing ran (fall back per group).   */
#define CTGAN_WGRAD_GROUP_MAX 8      /* problems per grouped kernel launch */
#define CTGAN_WGRAD_GROUP_LIMIT 32   /* groups per call */
#define CTGAN_REDUCE_BATCH 16        /* reductions per launch */
typedef struct ctgan_wgrad_group {
    ctgan_conv_desc d;               /* d.N ignored, as in ctgan_conv2d_wgrad_multi */
    int32_t nseg;
    int32_t Ns[CTGAN_WGRAD_MAX_SEGS];
    int32_t seg_flags[CTGAN_WGRAD_MAX_SEGS];
    const float* xs[CTGAN_WGRAD_MAX_SEGS];
    const float* dys[CTGAN_WGRAD_MAX_SEGS];
    float* dw;
    float* db;                       /* NULL unless a segment carries CTGAN_WGRAD_SEG_BIAS */
    /* optional finished addends (grouped launch only): dw = (sum over segments) + add_dw, db likewise - the weight gradient another
       launch already produced for the same filter.  May alias dw / db (in-place accumulation).  add_db is ignored when db is NULL. */
    const float* add_dw;
    const float* add_db;
} ctgan_wgrad_group;
size_t ctgan_conv2d_wgrad_group_workspace_bytes(const ctgan_wgrad_group* groups, int32_t n);
int ctgan_conv2d_wgrad_group(const ctgan_wgrad_group* groups, int32_t n, void* ws, size_t ws_bytes,
                             ctgan_stream_t stream);
/* The two phases of the grouped launch separately (bench.py times the GEMM launches alone, as rocprofv3 reports them): the grouped
   weight-gradient kernels write the split-K slabs, the batched reduction sums them into dw / db in a fixed order.             */
enum { CTGAN_WGRAD_GROUP_GEMM = 1, CTGAN_WGRAD_GROUP_REDUCE = 2,
       /* optional: restrict the GEMM phase to the launch of one tile configuration (bit CTGAN_WGRAD_GROUP_TILE0 << t, t = 0..3) */
       CTGAN_WGRAD_GROUP_TILE0 = 16, CTGAN_WGRAD_GROUP_TILE_MASK = 16 | 32 | 64 | 128 };
/* The same grouped call on the 16-bit family - mma = CTGAN_MMA_F32X3 (fp32 accuracy on the bf16 matrix cores at 6/16 of the fp32 MFMA's
   cost: the hybrid fp32 mode's weight gradients) or CTGAN_MMA_BF16 / CTGAN_MMA_F16 (the small weight gradients of the mixed-precision configs):
   every segment of every group is one problem of ONE launch of 128x128-tile workgroups with a common pixels-per-split, then one batched
   reduction (add_dw / add_db as above).  Members: C and K multiples of 128, power-of-two pixel grid (split mode; Q % 4 == 0 otherwise), unit channel stride of x, dense
   channels-last dy, no x_up; workspace_bytes returns 0 and the call CTGAN_E_UNSUPPORTED (nothing launched) if a member does not
   qualify - the caller then uses ctgan_conv2d_wgrad_group for it.  phases as CTGAN_WGRAD_GROUP_GEMM | CTGAN_WGRAD_GROUP_REDUCE; with
   CTGAN_WGRAD_GROUP_TILE0 << 0 / << 1 the GEMM phase launches only the filter-column / only the slice kernel.                  */
size_t ctgan_conv2d16_wgrad_group_workspace_bytes(const ctgan_wgrad_group* groups, int32_t n, int mma);
int ctgan_conv2d16_wgrad_group(const ctgan_wgrad_group* groups, int32_t n, int mma, void* ws, size_t ws_bytes, int phases,
                               ctgan_stream_t stream);
/* the launch (0..3) of a grouped call that problem g rides in; -1 if it does not qualify */
int ctgan_conv2d_wgrad_group_tile(const ctgan_wgrad_group* g);
int ctgan_conv2d_wgrad_group_ex(const ctgan_wgrad_group* groups, int32_t n, void* ws, size_t ws_bytes, int phases,
                                ctgan_stream_t stream);
/* ---- convolution family  (replaces tf.nn.conv2d TF/tflib/ops/conv2d.py:106-112,
 *      tf.nn.conv2d_transpose TF/tflib/ops/deconv2d.py:97-103, tf.matmul
 *      TF/tflib/ops/linear.py:132-137 (as 1x1 conv on a 1x1 image), and the conv gradient nodes
 *      tf.gradients / compute_gradients generate, TF/CT_gan_cifar_resnet.py:284,335-336) ------- */
size_t ctgan_conv2d_workspace_bytes(const ctgan_conv_desc* d, int op);
/* y = conv(x,w) [+ bias[k]] [+ resid (same strides as y)] [relu if flags&CTGAN_EPI_RELU]       */
int ctgan_conv2d_fwd(const ctgan_conv_desc* d, const float* x, const float* w, const float* bias,
                     const float* resid, float* y, int flags, ctgan_stream_t stream);
/* dx = adjoint of conv w.r.t. x, applied to dy [+ bias[c]] (dx has the strides d->xs; x_up must
 * be 0).  With bias this is Deconv2D's forward (TF/tflib/ops/deconv2d.py:97-110).               */
/* epilogue: dx = (conv^T(dy,w) + bias) [kept only where mask > 0: the ReLU backward of conv(relu(x)),
 * mask = x] [+ resid].  mask and resid have the strides of dx.                                    */
int ctgan_conv2d_dgrad(const ctgan_conv_desc* d, const float* dy, const float* w, const float* bias,
                       const float* mask, const float* resid, float* dx, void* ws, size_t ws_bytes,
                       int flags, ctgan_stream_t stream);
/* wt[r',s',k,c] = w[R-1-r',S-1-s',c,k]: the rotated, I/O-swapped filter the data gradient multiplies
 * with.  Callers that run several dgrads per weight update repack once and pass
 * CTGAN_DGRAD_W_REPACKED (then `w` is wt and no workspace is needed).                            */
int ctgan_conv2d_repack_filter(const ctgan_conv_desc* d, const float* w, float* wt, ctgan_stream_t stream);
/* ---- Layernorm, fused (csrc/layernorm.hip) ----------------------------------------------------------------------------------
 * TF/tflib/ops/layernorm.py:6-20 (tf.nn.moments over the non-batch axes + tf.nn.batch_normalization, eps 1e-5) for dense
 * tensors with the channel axis fastest ([N,H,W,C] or [N,C]); D = elements per sample, C = channels (scale / offset size).
 *   fwd : y = (x - mean) * rstd * scale[c] + offset[c]; mean / rstd [N] are returned for the backward passes
 *   bwd : gx (and, when gscale / goffset != NULL, the parameter gradients) from the gradient gy of y
 *   bwd2: the adjoint of bwd - the gradient penalty differentiates the critic twice (LS/wgan_LSUN_Bedrooms128.py:256-262):
 *         given the cotangent u of gx returns the cotangents of gy, x and scale (NULL = not wanted)
 * Supported when C % 4 == 0 and 1024 % C == 0 (ctgan_layernorm_supported); otherwise CTGAN_E_UNSUPPORTED and the caller
 * composes the operator from ctgan_sample_sum / ctgan_mul / ctgan_rsqrt / ctgan_channel_affine.  ws: scratch of
 * ctgan_layernorm_workspace_bytes(N, D, C) bytes.                                                                       */
int ctgan_layernorm_supported(int64_t D, int32_t C);
size_t ctgan_layernorm_workspace_bytes(int32_t N, int64_t D, int32_t C);
/* relu != 0: y = relu(Layernorm(x)) (the critics' Normalize -> nonlinearity, LS/wgan_LSUN_Bedrooms128.py:122-128) in the same
 * pass; the backward maps then take that y as `ymask` (NULL = no fused ReLU) and read gy as gy * (y > 0).                   */
int ctgan_layernorm_fwd(const float* x, const float* scale, const float* offset, float* y, float* mean, float* rstd, int32_t N,
                        int64_t D, int32_t C, float eps, int32_t relu, void* ws, size_t ws_bytes, ctgan_stream_t stream);
int ctgan_layernorm_bwd(const float* gy, const float* x, const float* scale, const float* mean, const float* rstd,
                        const float* ymask, float* gx, float* gscale, float* goffset, int32_t N, int64_t D, int32_t C, void* ws,
                        size_t ws_bytes, ctgan_stream_t stream);
int ctgan_layernorm_bwd2(const float* u, const float* gy, const float* x, const float* scale, const float* mean,
                         const float* rstd, const float* ymask, float* cot_gy, float* cot_x, float* cot_scale, int32_t N,
                         int64_t D, int32_t C, void* ws, size_t ws_bytes, ctgan_stream_t stream);
/* ---- 16-bit matrix-core family (csrc/igemm16.hip): BASELINE.json configs[1] "bf16" and configs[4] "fp16 MFMA conv" ----
 * The same three operators (tf.nn.conv2d TF/tflib/ops/conv2d.py:106-112, tf.nn.conv2d_transpose / the data gradient
 * TF/tflib/ops/deconv2d.py:91-103, the filter gradient tf.gradients derives) computed as mixed precision: operands rounded
 * (nearest-even) to bf16 or fp16 on their way into LDS, v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulation; activations,
 * gradients and master weights stay fp32 in caller memory.  The filter operand is a packed 16-bit image built once per
 * weight version by ctgan_conv2d16_pack_filter (FWD: [K][(r,s,c)]; DGRAD: per output-parity phase [C][(t,u,k)] with the taps
 * of that phase only, rotated).  `mma` = CTGAN_MMA_BF16 | CTGAN_MMA_F16.  Shapes outside the family (few-channel layers, odd
 * extents, x_up) return CTGAN_E_UNSUPPORTED: the caller uses the fp32 entry points above for those layers.
 * Epilogues as the fp32 family: fwd  y = [relu](conv(x | relu(x)) + bias + resid);  dgrad  dx = (conv^T(dy) + bias)
 * [kept where mask > 0] [+ resid].  flags = CTGAN_EPI_RELU | CTGAN_IN_RELU.  The bias gradient is not fused here
 * (ctgan_colsum).                                                                                                     */
/* CTGAN_MMA_F32X3: fp32 arithmetic on the bf16 matrix cores.  Every fp32 operand is split exactly into three bf16 terms
 * (x = h + m + l, nearest-even at each level: 24 significand bits) while it is staged, and a product is the six bf16 MFMAs
 * hh + hm + mh + mm + hl + lh accumulated in fp32 - the dropped terms are below 2^-24 relative, i.e. the result carries fp32
 * rounding accuracy (measured against an fp64 reference it is as close as the fp32 MFMA family), at 6/16 of the fp32 MFMA's
 * cycle cost.  The packed filter holds the three planes one after the other.                                               */
enum { CTGAN_MMA_BF16 = 1, CTGAN_MMA_F16 = 2, CTGAN_MMA_F32X3 = 3 };
int ctgan_conv2d16_supported(const ctgan_conv_desc* d, int op, int mma);        /* op = CTGAN_CONV_{FWD,DGRAD,WGRAD}; 1 / 0 */
/* 1 where CTGAN_MMA_F32X3 is the faster fp32 path for this launch (stride-1 convs / data gradients on whole-row or whole-image
 * 128x128 tiles, stride-2 convs / data gradients, weight gradients over >= 32k pixels - each only when the launch fills the chip):
 * the caller may route such layers through ctgan_conv2d16_{fwd,dgrad,wgrad} with mma = CTGAN_MMA_F32X3 and keep every other layer
 * on the fp32 MFMA entry points - the results carry fp32 accuracy either way.  A forward launch with prefers = 1 and stride 1 also
 * accepts CTGAN_RESID_UP.                                                                                                    */
int ctgan_conv2d16_x3_prefers(const ctgan_conv_desc* d, int op);
/* 1 when the filter-column weight-gradient kernel (csrc/wgrad16c.hip) takes this problem over `rows` samples in mode mma - geometry, the
 * strides of x in d->xs (images dense in memory) and the 32-bit offset limits; callers that queue weight gradients for
 * ctgan_conv2d16_wgrad_group use it to tell the problems the grouped launch runs on that kernel from those it would run on the slice tiles. */
int ctgan_conv2d16_wgrad_col_takes(const ctgan_conv_desc* d, int mma, int32_t rows);
size_t ctgan_conv2d16_filter_elems(const ctgan_conv_desc* d, int op, int mma);  /* 16-bit elements of the packed filter     */
int ctgan_conv2d16_pack_filter(const ctgan_conv_desc* d, int op, int mma, const float* w, void* wp, ctgan_stream_t stream);
/* n packs in one launch (all images of one weight version): descs[i] / ops[i] (CTGAN_CONV_FWD | CTGAN_CONV_DGRAD) / ws[i] -> wps[i] */
int ctgan_conv2d16_pack_batch(const ctgan_conv_desc* descs, const int32_t* ops, int32_t n, int mma, const float* const* ws,
                              void* const* wps, ctgan_stream_t stream);
/* ws (optional, ctgan_conv2d16_workspace_bytes(d, op) bytes): slabs for a K split of launches whose pixel x channel tiles cannot
 * fill the chip (8x8 / 4x4 layers at batch 64); NULL = never split.                                                      */
size_t ctgan_conv2d16_workspace_bytes(const ctgan_conv_desc* d, int op);
int ctgan_conv2d16_fwd(const ctgan_conv_desc* d, int mma, const float* x, const void* wp, const float* bias, const float* resid,
                       float* y, int flags, void* ws, size_t ws_bytes, ctgan_stream_t stream);
/* forward with the epilogue dropout of ctgan_conv2d_fwd_ex (same ext, same draws): mma = CTGAN_MMA_F32X3 on launches the halo-patch
 * kernel takes (ctgan_conv2d16_x3_prefers, stride 1); CTGAN_E_UNSUPPORTED otherwise - the caller then uses ctgan_conv2d_fwd_ex.   */
int ctgan_conv2d16_fwd_ex(const ctgan_conv_desc* d, int mma, const float* x, const void* wp, const float* bias, const float* resid,
                          float* y, int flags, const ctgan_epilogue_ext* ext, void* ws, size_t ws_bytes, ctgan_stream_t stream);
int ctgan_conv2d16_dgrad(const ctgan_conv_desc* d, int mma, const float* dy, const void* wp, const float* bias,
                         const float* mask, const float* resid, float* dx, int flags, void* ws, size_t ws_bytes,
                         ctgan_stream_t stream);
/* data gradient with the epilogue dropout of ctgan_conv2d_dgrad_ex (dx multiplied by the mask of ext's dropout, same draws at the same
 * physical offsets): mma = CTGAN_MMA_F32X3, stride 1, dense channels-last dx, launches the halo-patch kernels take; CTGAN_E_UNSUPPORTED
 * otherwise - the caller then uses ctgan_conv2d_dgrad_ex.                                                                          */
int ctgan_conv2d16_dgrad_ex(const ctgan_conv_desc* d, int mma, const float* dy, const void* wp, const float* bias,
                            const float* mask, const float* resid, float* dx, int flags, const ctgan_epilogue_ext* ext, void* ws,
                            size_t ws_bytes, ctgan_stream_t stream);
size_t ctgan_conv2d16_wgrad_workspace_bytes(const ctgan_conv_desc* d, int mma);
int ctgan_conv2d16_wgrad(const ctgan_conv_desc* d, int mma, const float* x, const float* dy, float* dw, void* ws,
                         size_t ws_bytes, int flags, ctgan_stream_t stream);
/* ... with the bias gradient db[K] = sum over pixels of dy fused into the same launch (the workgroups of the first tile row sum the dy
 * tiles they stage; fixed-order reduction): replaces the separate ctgan_colsum pass.  db NULL = ctgan_conv2d16_wgrad.              */
int ctgan_conv2d16_wgrad_bias(const ctgan_conv_desc* d, int mma, const float* x, const float* dy, float* dw, float* db, void* ws,
                              size_t ws_bytes, int flags, ctgan_stream_t stream);
/* flags of ctgan_conv2d16_wgrad(_bias) besides CTGAN_IN_RELU: launch only the split-K GEMM (slabs stay in ws) / only the reduction of
   slabs a GEMM-only call left there - bench.py times the two kernels apart, as rocprofv3 lists them                               */
enum { CTGAN_WGRAD16_GEMM_ONLY = 0x100, CTGAN_WGRAD16_REDUCE_ONLY = 0x200 };
/* dw[r,s,c,k] = sum_{n,p,q} x[..] * dy[n,k,p,q]   (HWIO, contiguous; deterministic split-K);
 * db[k] = sum_{n,p,q} dy[n,k,p,q] when db != NULL (tf.nn.bias_add gradient, fused when possible) */
int ctgan_conv2d_wgrad(const ctgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                       void* ws, size_t ws_bytes, int flags /* CTGAN_IN_RELU: x -> relu(x) */,
                       ctgan_stream_t stream);
/* Patch expansion for convs with very few input channels (first critic conv, TF/CT_gan_cifar_resnet.py:
 * 144,150; TF/CT_gan_cifar.py:84; TF/CT_gan_mnist.py:92): cols[n,p,q,(r*S+s)*C+c] = x[n,c,p*st-pt+r,q*st-pl+s]
 * (zero outside the image and for columns >= R*S*C; cols is channels-last [N,P,Q,cpad], contiguous).
 * col2im is its adjoint (dx has the strides d->xs).                                              */
int ctgan_im2col(const ctgan_conv_desc* d, const float* x, int32_t cpad, float* cols, ctgan_stream_t stream);
int ctgan_col2im(const ctgan_conv_desc* d, const float* cols, int32_t cpad, float* dx, ctgan_stream_t stream);
/* out[j] = sum_i x[i*ld + j], i<rows, j<cols (bias gradient; tf.nn.bias_add grad)             */
int ctgan_colsum(const float* x, int64_t rows, int32_t cols, int64_t ld, float* out,
                 void* ws, size_t ws_bytes, ctgan_stream_t stream);
size_t ctgan_colsum_workspace_bytes(int64_t rows, int32_t cols);

/* ---- elementwise / layout (K9-K12, K19 of SURVEY 2.1) -------------------------------------- */
/* y = x>0 ? x : alpha*x   (tf.nn.relu alpha=0; LeakyReLU tf.maximum(alpha*x,x) alpha=0.2)     */
int ctgan_lrelu_fwd(const float* x, float* y, int64_t n, float alpha, ctgan_stream_t stream);
/* gx = ref>0 ? gy : alpha*gy  (ref = forward input or output; same sign)                      */
int ctgan_lrelu_bwd(const float* gy, const float* ref, float* gx, int64_t n, float alpha,
                    ctgan_stream_t stream);
/* gx = scale * (ref>0 ? gy : alpha*gy): backward of y = dropout(relu(z)) computed in a conv epilogue, where
 * ref = y (y > 0 iff z > 0 and the element was kept) and scale = 1/keep                          */
int ctgan_lrelu_bwd_scaled(const float* gy, const float* ref, float* gx, int64_t n, float alpha, float scale,
                           ctgan_stream_t stream);
/* tf.nn.dropout: y = x/keep * floor(keep + u)  (TF/CT_gan_cifar_resnet.py:173-177); the
 * backward is the same call on the gradient.                                                  */
int ctgan_dropout(const float* x, const float* u, float* y, int64_t n, float keep,
                  ctgan_stream_t stream);
int ctgan_tanh_fwd(const float* x, float* y, int64_t n, ctgan_stream_t stream);
int ctgan_tanh_bwd(const float* gy, const float* y, float* gx, int64_t n, ctgan_stream_t stream);
int ctgan_sigmoid_fwd(const float* x, float* y, int64_t n, ctgan_stream_t stream);
int ctgan_sigmoid_bwd(const float* gy, const float* y, float* gx, int64_t n, ctgan_stream_t stream);
/* out = a*x + b*y  (residual add; y may be NULL => out = a*x)                                  */
int ctgan_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b,
                ctgan_stream_t stream);
/* generic 4-D strided copy (NCHW <-> NHWC repack): y[i0,i1,i2,i3] = x[i0,i1,i2,i3]            */
int ctgan_copy4d(const float* x, const int64_t xs[4], float* y, const int64_t ys[4],
                 const int32_t dims[4], ctgan_stream_t stream);
/* y[n,c,p,q] = scale * sum of the 2x2 window of x  (mean-pool: scale=.25,
 * TF/CT_gan_cifar_resnet.py:91,96; also the adjoint of upsample2 with scale=1)                 */
int ctgan_pool2(const float* x, const int64_t xs[4], float* y, const int64_t ys[4],
                const int32_t ydims[4] /* n,c,p,q */, float scale, ctgan_stream_t stream);
/* y[n,c,h,w] = scale * x[n,c,h/2,w/2]  (nearest 2x upsample, TF/CT_gan_cifar_resnet.py:102-105;
 * also the adjoint of pool2)                                                                   */
int ctgan_upsample2(const float* x, const int64_t xs[4], float* y, const int64_t ys[4],
                    const int32_t ydims[4] /* n,c,h,w */, float scale, ctgan_stream_t stream);
/* Filters of a conv fused with its 2x resampling (exact algebra, fewer multiplies):
 *   ConvMeanPool (TF/CT_gan_cifar_resnet.py:89-92): pool2(conv_RxS(x, w))  = conv_{(R+1)x(S+1), stride 2}(x, spread(w)/4)
 *   UpsampleConv (:100-107):   conv_RxS(upsample2(x), w) = conv2d_transpose_{(R+1)x(S+1), stride 2}(x, flipped spread(w))
 * spread: out[u,v,c,k] = scale * sum_{a,b in {0,1}} w[u-a, v-b, c, k]   (w is [R,S,C,K], out [(R+1),(S+1),C,K]);
 * flip != 0 writes the rotated, I/O-swapped filter out[R-u, S-v, k, c] instead ([(R+1),(S+1),K,C]).        */
int ctgan_filter_spread(const float* w, float* out, int32_t R, int32_t S, int32_t C, int32_t K, float scale,
                        int32_t flip, ctgan_stream_t stream);
/* the adjoint map (weight gradient back to the RxS filter): out[r,s,c,k] = scale * sum_{a,b} W[r+a, s+b, c, k],
 * W = w4 (flip == 0) or its un-flipped view W[u,v,c,k] = w4[R-u, S-v, k, c] (flip != 0)                      */
int ctgan_filter_fold(const float* w4, float* out, int32_t R, int32_t S, int32_t C, int32_t K, float scale,
                      int32_t flip, ctgan_stream_t stream);
/* All derived filters of one weight update in a single launch (per CTGAN_FILTER_BATCH jobs): every job reads an
 * HWIO filter src[R,S,C,K] and writes one derived layout.  `jobs` is a HOST array; it travels in the kernel
 * arguments, so the call is hipGraph-capture safe.
 *   CTGAN_FILTER_ROTATE      dst[R,S,K,C]            = what ctgan_conv2d_repack_filter writes for a stride-1 conv
 *   CTGAN_FILTER_PHASES      dst[4][ceil(R/2)][ceil(S/2)][K][C] = ... for a stride-2 conv with pads (pad_t, pad_l)
 *   CTGAN_FILTER_SPREAD      dst[(R+1),(S+1),C,K]    = ctgan_filter_spread(scale, flip = 0)
 *   CTGAN_FILTER_SPREAD_FLIP dst[(R+1),(S+1),K,C]    = ctgan_filter_spread(scale, flip = 1)                  */
#define CTGAN_FILTER_ROTATE 0
#define CTGAN_FILTER_PHASES 1
#define CTGAN_FILTER_SPREAD 2
#define CTGAN_FILTER_SPREAD_FLIP 3
/* pre = CTGAN_FILTER_SPREAD / _SPREAD_FLIP with kind = ROTATE / PHASES: the layout is taken OF the spread filter
 * (pre_scale = its scale; its taps are (R+1) x (S+1), and for _SPREAD_FLIP its channel roles are swapped) straight from
 * src, bit-identical to running the spread job first and the layout job on its output - but independent of it.          */
#define CTGAN_FILTER_BATCH 40
typedef struct ctgan_filter_job {
    const float* src;
    float* dst;
    int32_t R, S, C, K;
    int32_t kind;
    int32_t pad_t, pad_l;
    float scale;
    int32_t pre;
    float pre_scale;
} ctgan_filter_job;
int ctgan_filter_batch(const ctgan_filter_job* jobs, int32_t n, ctgan_stream_t stream);
/* Several ctgan_filter_fold calls in one launch: dst[R,S,C,K] = scale * fold(src [(R+1),(S+1),C,K] or, flip, [(R+1),(S+1),K,C]). */
#define CTGAN_FOLD_BATCH 16
typedef struct ctgan_fold_job {
    const float* src;
    float* dst;
    int32_t R, S, C, K;
    float scale;
    int32_t flip;
} ctgan_fold_job;
int ctgan_filter_fold_batch(const ctgan_fold_job* jobs, int32_t n, ctgan_stream_t stream);
/* ---- Layernorm primitives (TF/tflib/ops/layernorm.py:6-20: tf.nn.moments over (C,H,W) per sample +
 *      tf.nn.batch_normalization with per-channel scale/offset).  The operator is composed of these linear /
 *      elementwise maps so that its first AND second derivatives (gradient penalty through a layer-normalised
 *      critic, config[4]) are compositions of the same kernels. ---------------------------------------------- */
int ctgan_mul(const float* x, const float* y, float* out, int64_t n, ctgan_stream_t stream);           /* out = x*y    */
int ctgan_rsqrt(const float* x, float* y, int64_t n, float eps, ctgan_stream_t stream);               /* 1/sqrt(x+eps) */
/* y[i] = scale * sum_j x[i][j] for a dense [n][m] tensor (per-sample reduction, fixed order)                    */
int ctgan_sample_sum(const float* x, float* y, int32_t n, int64_t m, float scale, ctgan_stream_t stream);
/* y[i][j] = scale * v[i]  (its adjoint)                                                                         */
int ctgan_sample_bcast(const float* v, float* y, int32_t n, int64_t m, float scale, ctgan_stream_t stream);
/* channels-last x [rows][c]: y = x*scale[c] + offset[c] (offset may be NULL)                                    */
int ctgan_channel_affine(const float* x, const float* scale, const float* offset, float* y, int64_t rows, int32_t c,
                         ctgan_stream_t stream);
/* y[n,c] = scale * sum_{hw} x[n,hw,c]  (tf.reduce_mean(axis=[2,3]) :179) on channels-last x    */
int ctgan_spatial_sum(const float* x, float* y, int32_t n, int32_t hw, int32_t c, float scale,
                      ctgan_stream_t stream);
/* y[n,hw,c] = scale * g[n,c]  (its adjoint)                                                    */
int ctgan_spatial_bcast(const float* g, float* y, int32_t n, int32_t hw, int32_t c, float scale,
                        ctgan_stream_t stream);
/* real = 2*((int/denom)-.5) + noise   (TF/CT_gan_cifar_resnet.py:201-202; noise may be NULL:
 * TF/CT_gan_cifar.py:103 with denom 255)                                                       */
int ctgan_real_prep(const int32_t* x_int, const float* noise, float* y, int64_t n, float denom,
                    ctgan_stream_t stream);
/* out[b,:] = real[b,:] + alpha[b]*(fake[b,:]-real[b,:])   (:282-283)                           */
int ctgan_interpolate(const float* real, const float* fake, const float* alpha, float* out,
                      int32_t b, int32_t d, ctgan_stream_t stream);

/* ---- batch norm (K13/K14): training-mode statistics over (n,h,w) per channel, biased variance,
 *      eps added inside rsqrt; `groups` independent row groups of n/groups samples each model the
 *      reference's per-tower statistics (TF/CT_gan_cifar_resnet.py:196-199,316-321).
 *      x is channels-last [n, hw, c].  scale/offset are [n_labels, c] gathered by labels[n]
 *      (cond_batchnorm.py:12-16) or [c] when labels == NULL (batchnorm.py:23-30).  ---------- */
int ctgan_bn_stats(const float* x, int32_t n, int32_t hw, int32_t c, int32_t groups, float eps,
                   float* mean /*[groups,c]*/, float* rstd /*[groups,c]*/,
                   void* ws, size_t ws_bytes, ctgan_stream_t stream);
int ctgan_bn_apply(const float* x, const float* mean, const float* rstd, const float* scale,
                   const float* offset, const int32_t* labels, float* y, int32_t n, int32_t hw,
                   int32_t c, int32_t groups, int32_t relu, ctgan_stream_t stream);
/* backward of y = [relu]( xhat*scale + offset ): gy -> gx, gscale, goffset                     */
int ctgan_bn_bwd(const float* gy, const float* x, const float* mean, const float* rstd,
                 const float* scale, const float* offset, const int32_t* labels,
                 float* gx, float* gscale, float* goffset, int32_t n, int32_t hw, int32_t c,
                 int32_t groups, int32_t n_labels, int32_t relu, void* ws, size_t ws_bytes,
                 ctgan_stream_t stream);
size_t ctgan_bn_workspace_bytes(int32_t n, int32_t hw, int32_t c, int32_t groups, int32_t n_labels);

/* ---- fused loss heads (K16-K20) ------------------------------------------------------------- */
/* slopes[b] = ||g[b,:]||_2 ; gp = lambda*mean((slopes-1)^2)   (TF/CT_gan_cifar_resnet.py:285-286); gp == NULL: slopes only */
int ctgan_gp_fwd(const float* g, int32_t b, int32_t d, float lambda, float* slopes, float* gp,
                 ctgan_stream_t stream);
/* gg[b,:] = gout * lambda*2*(s-1)/(s*B) * g[b,:]                                               */
int ctgan_gp_bwd(const float* g, const float* slopes, const float* gout, int32_t b, int32_t d,
                 float lambda, float* gg, ctgan_stream_t stream);
/* ctgan_gp_bwd that also takes the penalty's value in the same launch: gp[0] = lambda*mean((slopes-1)^2) (gp may be NULL) and, with
 * out5 != NULL, out5[0] += gp, out5[4] += gp - the two sums of ctgan_tail_critic_heads_fwd (cost; wgan + ct + gp) that contain the
 * penalty, for a step whose loss heads ran before dD/dx_hat existed (the hand-scheduled critic step: TF/CT_gan_cifar_resnet.py:284-286,
 * :295-300 evaluated after the merged backward).                                                                               */
int ctgan_gp_bwd_mean(const float* g, const float* slopes, const float* gout, int32_t b, int32_t d, float lambda, float* gg, float* gp,
                      float* out5, ctgan_stream_t stream);
/* CT_i = l2*(d-d_)^2 + l2*0.1*mean_j (f-f_)^2 ; ct = mean_i max(CT_i - M, 0)   (:288-291)      */
int ctgan_ct_fwd(const float* d, const float* d_, const float* f, const float* f_, int32_t b,
                 int32_t nf, float lambda2, float M, float* ct_i, float* ct, ctgan_stream_t stream);
int ctgan_ct_bwd(const float* d, const float* d_, const float* f, const float* f_,
                 const float* ct_i, const float* gout, int32_t b, int32_t nf, float lambda2, float M,
                 float* gd, float* gd_, float* gf, float* gf_, ctgan_stream_t stream);
/* loss = mean_i CE(logits[i,:], labels[i]); probs saved for backward; correct = #argmax==label  */
int ctgan_softmax_ce_fwd(const float* logits, const int32_t* labels, int32_t b, int32_t ncls,
                         float* probs, float* loss, float* n_correct, ctgan_stream_t stream);
int ctgan_softmax_ce_bwd(const float* probs, const int32_t* labels, const float* gout, int32_t b,
                         int32_t ncls, float* glogits, ctgan_stream_t stream);
/* out = sign_a*mean(x[0:na]) + sign_b*mean(x[na:na+nb])  (WGAN critic / generator cost :244,322) */
int ctgan_mean_diff_fwd(const float* x, int32_t na, int32_t nb, float sa, float sb, float* out,
                        ctgan_stream_t stream);
int ctgan_mean_diff_bwd(const float* gout, int32_t na, int32_t nb, float sa, float sb, float* gx,
                        ctgan_stream_t stream);

/* All loss heads of one critic step over the batched dropout passes (TF/CT_gan_cifar_resnet.py:244-248,288-291):
 * d [3B], f [3B,nf], a [3B,ncls] (a may be NULL) with rows [0,B) = real pass 1, [B,2B) = fake pass 1, [2B,3B) = real
 * pass 2.  out[5] = {wgan + ct + gp + acgan_scale*acgan, wgan, ct, acgan, wgan + ct + gp} with gp = the step's gradient
 * penalty scalar (device pointer, may be NULL = 0; :284-286, :304); ct_i [B] and probs [B,ncls] are kept for the
 * backward, which writes the full gradients gd [3B], gf [3B,nf], ga [3B,ncls] given gout[n_gout] = upstream
 * gradients of the four outputs (n_gout = 4) or of the summed cost only (n_gout = 1).  One launch each instead of ~25. */
int ctgan_critic_heads_fwd(const float* d, const float* f, const float* a, const int32_t* labels, const float* gp,
                           int32_t B, int32_t nf, int32_t ncls, float lambda2, float M, float acgan_scale, float* ct_i,
                           float* probs, float* out, ctgan_stream_t stream);
int ctgan_critic_heads_bwd(const float* d, const float* f, const float* probs, const int32_t* labels, const float* ct_i,
                           const float* gout, int32_t n_gout, int32_t B, int32_t nf, int32_t ncls, float lambda2,
                           float M, float acgan_scale, float* gd, float* gf, float* ga, ctgan_stream_t stream);
/* The critic's output head fused around those loss heads (TF/CT_gan_cifar_resnet.py:179-186): y = the last residual
 * block's relu(dropout(.)) output, dense channels-last [n][hw][nf].
 *   tail_heads_fwd: f [n,nf] = mean over hw (:180; of relu(y) when relu != 0, :179), d [n] = f.w_out + b_out (:181, w_out [nf,1]; NULL = skip),
 *                   a [n,ncls] = f.w_ac + b_ac (:183, w_ac [nf,ncls]; NULL = skip) - one launch for reduce_mean + 2 Linear.
 *   tail_heads_bwd: given the saved d, f, probs, ct_i of ctgan_critic_heads_fwd and gout as in ctgan_critic_heads_bwd, writes
 *                   gy [3B][hw][nf] = the gradient w.r.t. the last block's PRE-activation (head gradients pushed through both
 *                   Linear layers, the mean and the relu/dropout mask y > 0 with its 1/keep = mask_scale) and the head
 *                   weight gradients gw_out [nf], gb_out [1], gw_ac [nf,ncls], gb_ac [ncls] - one launch instead of nine.
 *   gp_head_grad / gp_head_wgrad: the gradient-penalty branch needs only dD/dz (:284): gz = (y > 0) * w_out / hw *
 *                   mask_scale, and in the double backward gw_out[j] = mask_scale / hw * sum_{y > 0} gg[.,.,j].            */
int ctgan_tail_heads_fwd(const float* y, int32_t n, int32_t hw, int32_t nf, int32_t relu, const float* w_out,
                         const float* b_out, const float* w_ac, const float* b_ac, int32_t ncls, float* f, float* d,
                         float* a, ctgan_stream_t stream);
/* tail_heads_fwd on the 3B rows of a critic step (relu = 0) + ctgan_critic_heads_fwd in two launches: the workgroup of
 * real sample i owns both of its dropout passes (rows i and 2B+i) and writes ct_i[i], probs[i,:] and ce_i[i] (scratch
 * [B]); a one-workgroup kernel then takes the batch means into out[5].                                                */
int ctgan_tail_critic_heads_fwd(const float* y, int32_t B, int32_t hw, int32_t nf, const float* w_out, const float* b_out,
                                const float* w_ac, const float* b_ac, int32_t ncls, const int32_t* labels,
                                const float* gp, float lambda2, float M, float acgan_scale, float* f, float* d, float* a,
                                float* ct_i, float* probs, float* ce_i, float* out, ctgan_stream_t stream);
/* ... with two more reductions folded into the same two launches:
 *   slopes != NULL (gp then non-NULL): the gradient penalty's batch mean gp[0] = gp_lambda * mean((slopes - 1)^2) (:286) is taken by the
 *     one-workgroup kernel and written to gp[0] before it enters out[] (pair with ctgan_gp_fwd(.., gp = NULL));
 *   y_clean != NULL ([2B][hw][nf], real rows then fake rows, the dropout-free pass :228): a_clean [2B,ncls] = class head of its rows
 *     (relu'd first when clean_relu; f_clean [2B,nf] scratch) by extra workgroups of the rows launch, and the accuracies acc[2]
 *     (:249-266, as ctgan_accuracy2) by the one-workgroup kernel.                                                              */
int ctgan_tail_critic_heads_fwd2(const float* y, int32_t B, int32_t hw, int32_t nf, const float* w_out, const float* b_out,
                                 const float* w_ac, const float* b_ac, int32_t ncls, const int32_t* labels, float* gp,
                                 const float* slopes, float gp_lambda, const float* y_clean, int32_t clean_relu, float* f_clean,
                                 float* a_clean, float* acc, float lambda2, float M, float acgan_scale, float* f, float* d, float* a,
                                 float* ct_i, float* probs, float* ce_i, float* out, ctgan_stream_t stream);
int ctgan_tail_heads_bwd(const float* y, const float* d, const float* f, const float* probs, const int32_t* labels,
                         const float* ct_i, const float* gout, int32_t n_gout, int32_t B, int32_t hw, int32_t nf,
                         int32_t ncls, float lambda2, float M, float acgan_scale, float mask_scale, const float* w_out,
                         const float* w_ac, float* gy, float* gw_out, float* gb_out, float* gw_ac, float* gb_ac,
                         ctgan_stream_t stream);
/* Generator-step loss on the critic's output head (TF/CT_gan_cifar_resnet.py:321-330): cost = -mean(d) + ac_scale * CE(a, labels)
 * with f, d, a as in ctgan_tail_heads_fwd (two launches), and its backward straight to the gradient w.r.t. the last critic
 * conv's result (both Linear data gradients, the mean's broadcast, the relu/dropout mask) in one launch; the critic's weights
 * get no gradient in the generator step.                                                                                    */
int ctgan_gen_heads_fwd(const float* y, int32_t n, int32_t hw, int32_t nf, const float* w_out, const float* b_out,
                        const float* w_ac, const float* b_ac, int32_t ncls, const int32_t* labels, float ac_scale, float* f,
                        float* d, float* a, float* probs, float* out, ctgan_stream_t stream);
int ctgan_gen_heads_bwd(const float* y, const float* probs, const int32_t* labels, const float* gout, int32_t n, int32_t hw,
                        int32_t nf, int32_t ncls, float ac_scale, float mask_scale, const float* w_out, const float* w_ac,
                        float* gy, ctgan_stream_t stream);
/* Round 5 (the hand-scheduled critic step, critic_schedule.py): three launches folded into their neighbours.
 *   ctgan_tail_heads_bwd_gp  = ctgan_tail_heads_bwd that also writes gz_gp = ctgan_gp_head_grad(y_gp) for the n_gp rows of the penalty pass
 *                              (the seeds of BOTH backward chains in one launch - the step runs them as one chain);
 *   ctgan_gp_head_wgrad_acc  = ctgan_gp_head_wgrad that ADDS onto gw (the head weight's gradient from the dropout passes);
 *   ctgan_gp_finish          : ga [b, c*h*w] (NCHW, in place) += scale * upsample2(gs [b,c,h/2,w/2], element strides gs_strides[4]) and
 *                              slopes[b] = ||ga[b,:]||_2 - dD/dx_hat = gradient through the first conv + gradient through the pooled 1x1
 *                              shortcut (:146-153, :284-285) in one launch instead of upsample + add + ctgan_gp_fwd.                      */
int ctgan_tail_heads_bwd_gp(const float* y, const float* d, const float* f, const float* probs, const int32_t* labels, const float* ct_i,
                            const float* gout, int32_t n_gout, int32_t B, int32_t hw, int32_t nf, int32_t ncls, float lambda2, float M,
                            float acgan_scale, float mask_scale, const float* w_out, const float* w_ac, float* gy, float* gw_out,
                            float* gb_out, float* gw_ac, float* gb_ac, const float* y_gp, int32_t n_gp, float* gz_gp, ctgan_stream_t stream);
int ctgan_gp_head_wgrad_acc(const float* gg, const float* y, int32_t n, int32_t hw, int32_t nf, float mask_scale, float* gw,
                            float* ws /* 64 * nf floats */, ctgan_stream_t stream);
int ctgan_gp_finish(float* ga, const float* gs, const int64_t* gs_strides, int32_t b, int32_t c, int32_t h, int32_t w, float scale, float* slopes,
                    ctgan_stream_t stream);
int ctgan_gp_head_grad(const float* y, const float* w_out, int32_t n, int32_t hw, int32_t nf, float mask_scale,
                       float* gz, ctgan_stream_t stream);
int ctgan_gp_head_wgrad(const float* gg, const float* y, int32_t n, int32_t hw, int32_t nf, float mask_scale, float* gw,
                        float* ws /* 64 * nf floats */, ctgan_stream_t stream);
/* ACGAN accuracies of the clean critic pass (:249-266): logits [2B,ncls] (real rows, then fake rows); acc[2]   */
int ctgan_accuracy2(const float* logits, const int32_t* labels, int32_t B, int32_t ncls, float* acc,
                    ctgan_stream_t stream);

/* ---- optimizer (K21): tf.train.AdamOptimizer on a flat buffer
 *      (TF/CT_gan_cifar_resnet.py:333-338).  `state` = device float[4]: {lr, beta1^t, beta2^t, skipped};
 *      the kernel reads lr and the running beta powers from it (graph-replay safe) and
 *      ctgan_adam_advance multiplies the powers after all buckets of a step are applied.  An element whose scaled gradient is not
 *      finite (NaN / inf: an overflow of the fp16 matrix-core mode, a degenerate input) keeps its weight and slots unchanged - one
 *      inf would otherwise poison m, v and theta for good, also at lr = 0 (build-only safeguard; TF would propagate it) - and
 *      state[3] is incremented by one per skipped element (float accumulator, exact to 2^24), so the caller can see it. ------ */
int ctgan_adam_step(float* theta, const float* g, float* m, float* v, int64_t n, float* state,
                    float beta1, float beta2, float eps, float grad_scale, ctgan_stream_t stream);
int ctgan_adam_advance(float* state, float beta1, float beta2, ctgan_stream_t stream);
/* End of a step in one launch: ctgan_adam_advance(state) and, with rng_ctr != NULL, ctgan_rng_advance(rng_ctr, rng_by).            */
int ctgan_step_advance(float* state, float beta1, float beta2, uint64_t* rng_ctr, uint64_t rng_by, ctgan_stream_t stream);
/* ctgan_pack + ctgan_adam_step in one launch (n_tensors <= 64, 16-byte aligned flat buffers; CTGAN_E_UNSUPPORTED otherwise):
 * compute_gradients' bucket and apply_gradients (TF/CT_gan_cifar_resnet.py:335-338) of a single-rank step.  flat receives the
 * packed gradients as ctgan_pack writes them; theta/m/v are updated with bit-identical arithmetic to the two-launch form.        */
int ctgan_adam_step_packed(const float* const* srcs, const int64_t* dst_offs, const int64_t* counts, int32_t n_tensors, float* flat,
                           float* theta, float* m, float* v, float* state, float beta1, float beta2, float eps, float grad_scale,
                           ctgan_stream_t stream);
/* flat[dst_offs[i] : dst_offs[i] + counts[i]] = srcs[i][0:counts[i]] (srcs[i] == NULL: zeros), i < n_tensors, in one
 * launch per 64 tensors.  The three arrays are HOST arrays (the pointers are device pointers); they are passed to
 * the kernel by value, so the call is hipGraph-capture safe.  This is the gradient bucket of compute_gradients
 * (TF/CT_gan_cifar_resnet.py:335-336) without one copy per variable.                                   */
int ctgan_pack(const float* const* srcs, const int64_t* dst_offs, const int64_t* counts, int32_t n_tensors,
               float* flat, ctgan_stream_t stream);

/* ---- RNG: Philox4x32-10 counter-based streams (tf.random_uniform / tf.random_normal /
 *      dropout masks :157,202,277,319).  counter base is read from device memory
 *      (`ctr[0]`, advanced by ctgan_rng_advance) so captured graphs draw fresh numbers. -------- */
/* tf.nn.dropout with its mask drawn inside the kernel: y[i] = x[i]/keep * floor(keep + u_i), where u_i is what
 * ctgan_rng_uniform(out, n, seed, stream_id, ctr, 0, 1) writes to out[i].  The backward (and its backward) is the
 * same call on the gradient with the same (seed, stream_id, ctr) - no mask tensor is stored.              */
int ctgan_dropout_rng(const float* x, float* y, int64_t n, float keep, uint64_t seed, uint64_t stream_id,
                      const uint64_t* ctr, ctgan_stream_t stream);
/* ctgan_dropout_rng followed by ctgan_lrelu_bwd(., ref, alpha = 0) in one pass: y (NULL = not wanted) = dropout(x),
 * y_masked = y where ref > 0, else 0.  x, ref, y, y_masked share one physical layout.                        */
int ctgan_dropout_rng_mask(const float* x, const float* ref, float* y, float* y_masked, int64_t n, float keep, uint64_t seed,
                           uint64_t stream_id, const uint64_t* ctr, ctgan_stream_t stream);
/* LeakyReLU + dropout in one pass (TF/CT_gan_cifar.py:84-98, TF/CT_gan_mnist.py:92-100: `dropout(LeakyReLU(conv))`): y = x * (ref > 0 ? 1 :
 * alpha) / keep * floor(keep + u_i), draws as ctgan_dropout_rng.  Forward: ref = x.  Backward (and its backward): x = the arriving gradient,
 * ref = the forward result (kept values carry the pre-activation's sign, dropped ones are multiplied by 0).  One physical layout.     */
int ctgan_lrelu_dropout_rng(const float* x, const float* ref, float* y, int64_t n, float alpha, float keep, uint64_t seed,
                            uint64_t stream_id, const uint64_t* ctr, ctgan_stream_t stream);
/* The same over TWO tensors laid end to end: elements [0, n1) draw stream_id, elements [n1, n) draw stream_id2 indexed from n1 (n1 % 4 == 0) -
 * each part gets the draws a launch of its own would make (TF/CT_gan_cifar.py:84-98 evaluated once on [real, fake, real | x_hat]).        */
int ctgan_lrelu_dropout_rng2(const float* x, const float* ref, float* y, int64_t n, int64_t n1, float alpha, float keep, uint64_t seed,
                             uint64_t stream_id, uint64_t stream_id2, const uint64_t* ctr, ctgan_stream_t stream);
int ctgan_rng_uniform(float* out, int64_t n, uint64_t seed, uint64_t stream_id, const uint64_t* ctr,
                      float lo, float hi, ctgan_stream_t stream);
int ctgan_rng_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, const uint64_t* ctr,
                     ctgan_stream_t stream);
/* labels = (int32)(u*nlab), u~U[0,1)   (tf.cast(tf.random_uniform*10, int32) :319)             */
int ctgan_rng_labels(int32_t* out, int64_t n, int32_t nlab, uint64_t seed, uint64_t stream_id,
                     const uint64_t* ctr, ctgan_stream_t stream);
int ctgan_rng_advance(uint64_t* ctr, uint64_t by, ctgan_stream_t stream);
/* Critic-step input preparation in one launch (TF/CT_gan_cifar_resnet.py:201-202,226,277-283): rf [2b,d] = [real ; fake]
 * with real = 2*(x_int/denom - .5) + U[lo,hi) (stream sid_deq, element i), interp [b,d] = real + alpha*(fake - real) with
 * alpha[row] = U[0,1) (stream sid_alpha, element row) - the same draws as ctgan_rng_uniform on [b,d] and [b,1] tensors
 * followed by ctgan_real_prep, ctgan_interpolate and a concat.  d % 4 == 0.                                              */
int ctgan_critic_prep(const int32_t* x_int, const float* fake, int32_t b, int32_t d, uint64_t seed, uint64_t sid_deq,
                      uint64_t sid_alpha, const uint64_t* ctr, float lo, float hi, float denom, float* rf, float* interp,
                      ctgan_stream_t stream);
/* dst [n_src + n_extra rows] = tf.nn.dropout([src ; src[0:n_extra]], keep) in one launch (input of the critic tail for the
 * two dropout passes, pass 2 on the real half only, :226-227); draws as ctgan_dropout_rng on the concatenated tensor;
 * keep = 1: plain concat.  rows_cat_bwd: the concat's adjoint gsrc[r] = g[r] + g[n_src + r] (r < n_extra).                */
int ctgan_rows_cat_dropout(const float* src, int64_t n_src, int64_t n_extra, int64_t row_elems, float keep, uint64_t seed,
                           uint64_t stream_id, const uint64_t* ctr, float* dst, ctgan_stream_t stream);
int ctgan_rows_cat_bwd(const float* g, int64_t n_src, int64_t n_extra, int64_t row_elems, float* gsrc, ctgan_stream_t stream);
/* The same with n_pass further rows behind the concat that pass straight through: g = [a (n_src) ; a' (n_extra) ; c (n_pass)] ->
 * gsrc [n_src + n_pass rows] = [a + a' ; c].  One launch for the step from the critic tail's merged backward (rows real, fake |
 * real' | x_hat: the two dropout passes of :226-227 and the penalty pass of :283-284) to the trunk's (rows real, fake, x_hat).  */
int ctgan_rows_cat_bwd2(const float* g, int64_t n_src, int64_t n_extra, int64_t n_pass, int64_t row_elems, float* gsrc,
                        ctgan_stream_t stream);
/* General form: dst = concatenation of up to CTGAN_ROW_SEGMENTS row segments of src (segment i = rows [src_row0,
 * src_row0 + rows)), each with its own tf.nn.dropout (keep >= 1: none) whose Philox element index is counted from dst row
 * index_row0 (<= the segment's first dst row): segments that share index_row0 and stream_id reproduce the dropout of their
 * own concatenated tensor.  Builds the inputs of all critic tails of a step (dropout passes, clean pass, gradient-penalty
 * pass) in one launch.                                                                                                     */
#define CTGAN_ROW_SEGMENTS 6
typedef struct ctgan_row_segment {
    int64_t src_row0, rows;
    float keep;
    uint64_t stream_id;
    int64_t index_row0;
} ctgan_row_segment;
int ctgan_rows_gather_dropout(const float* src, const ctgan_row_segment* segs, int32_t nseg, int64_t row_elems, uint64_t seed,
                              const uint64_t* ctr, float* dst, ctgan_stream_t stream);

/* ---- vertical fusion of the critic's 8x8 residual blocks (round 6) -------------------------------------------------------------------------
 * Blocks 3 and 4 of the ResNet critic (ResidualBlock, TF/CT_gan_cifar_resnet.py:109-141; Discriminator :174-178: dropout - block - dropout -
 * block) on 8 x 8 x 128 images, as the backward passes see them: a chain of up to four 3x3 SAME convs in which each conv's result is masked by a
 * ReLU pattern of the forward pass, takes a residual, is multiplied by a tf.nn.dropout mask of the forward pass (redrawn from its Philox stream,
 * :173-177) and feeds the next conv - the data-gradient chain of compute_gradients(disc_cost) through the two blocks (:335-336) and the forward-mode
 * pass of the gradient penalty's double backward (tf.gradients inside the loss, :284).  One workgroup carries one image through the whole chain (the
 * image stays in LDS between the convs); every intermediate that a weight gradient reads is written once.  Split mode (CTGAN_MMA_F32X3) only.
 *
 *   value = x[image]
 *   for step s = 0 .. n_convs:        (step 0 has no filter: it only applies its epilogue to x)
 *       if s > 0: value = conv3x3_same(value, filter of step s)
 *       if mask:       value = mask[image] > 0 ? value : 0
 *       if resid:      value += slot[resid]                    (slot 1 / 2: values saved by earlier steps)
 *       if drop:       value *= floor(keep + u) / keep         u: element (offset / 4) of stream drop[drop-1], step drop_ctr[0]; images below n_split draw
 *                                                              stream_id_lo indexed from image 0, the others stream_id_hi indexed from image n_split
 *       if save:       slot[save] = value
 *       if post_mask:  value = post_mask[image] > 0 ? value : 0
 *       if out:        out[image] = value
 * All tensors dense channels-last [n_images, 8, 8, 128] fp32, 16-byte aligned.  wp: the packed image of a 3x3x128x128 filter as
 * ctgan_conv2d16_pack_filter(.., op, CTGAN_MMA_F32X3, ..) writes it (op CTGAN_CONV_FWD: the step is conv(., w); CTGAN_CONV_DGRAD: its transpose).
 * Deterministic (fixed summation order).  CTGAN_E_UNSUPPORTED for other image sizes / channel counts.                                           */
#define CTGAN_CHAIN_MAX_CONVS 4
typedef struct {
    const void* wp;                  /* step 0: NULL */
    const float* mask;
    const float* post_mask;
    float* out;
    int32_t resid, save, drop, reserved;
} ctgan_chain_step;
typedef struct {
    float keep;
    int32_t n_split;
    uint64_t stream_id_lo, stream_id_hi;
} ctgan_chain_drop;
typedef struct {
    const float* x;
    int32_t n_images, channels, height, width;
    int32_t n_convs, reserved;
    ctgan_chain_step step[CTGAN_CHAIN_MAX_CONVS + 1];
    ctgan_chain_drop drop[2];
    uint64_t drop_seed;
    const uint64_t* drop_ctr;
} ctgan_chain8x8;
int ctgan_conv2d16_chain8x8(const ctgan_chain8x8* chain, ctgan_stream_t stream);



#ifdef __cplusplus
}
#endif
#endif /* CTGAN_HIP_H */
