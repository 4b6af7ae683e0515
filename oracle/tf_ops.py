"""TensorFlow-1.x primitive semantics restated on PyTorch-CPU (oracle; test infrastructure only).

The reference calls these TF primitives (third-party, un-vendored: TensorFlow 1.2.1, README.md:3);
their documented behaviour is restated here.  All tensors are logical NCHW like the reference.
Works in whatever dtype the inputs carry (fp64 = truth, fp32 = twin).
"""
import math

import torch
import torch.nn.functional as F


def same_pads(in_size, k, stride):
    """TF 'SAME' rule: out = ceil(in/s); total = max((out-1)*s + k - in, 0); before = total//2."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d_same(x, w_hwio, stride=1):
    """tf.nn.conv2d(x, filter=[kh,kw,Cin,Cout], strides=[1,1,s,s], 'SAME', NCHW).

    Call site: TF/tflib/ops/conv2d.py:106-112.  Cross-correlation, asymmetric SAME padding.
    """
    kh, kw, cin, cout = w_hwio.shape
    _, pt, pb = same_pads(x.shape[2], kh, stride)
    _, pl, pr = same_pads(x.shape[3], kw, stride)
    xp = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(xp, w_hwio.permute(3, 2, 0, 1), stride=stride)


def conv2d_transpose_same(x, w_hwoi, stride=2):
    """tf.nn.conv2d_transpose(x, filter=[kh,kw,Cout,Cin], output_shape=[N,sH,sW,Cout], 'SAME').

    Call site: TF/tflib/ops/deconv2d.py:97-103.  Defined by TF as the exact adjoint (gradient
    w.r.t. input) of conv2d_same on a [N,Cout,sH,sW] tensor with the same filter seen as HWIO with
    I=Cout, O=Cin: the full transposed conv (size (H-1)*s + k) cropped at the forward conv's
    leading pad.
    """
    kh, kw, cout, cin = w_hwoi.shape
    H, W = x.shape[2], x.shape[3]
    oh, ow = H * stride, W * stride
    _, pt, _ = same_pads(oh, kh, stride)
    _, pl, _ = same_pads(ow, kw, stride)
    full = F.conv_transpose2d(x, w_hwoi.permute(3, 2, 0, 1), stride=stride)
    # the full result covers padded coordinates [-pt, ...); positions past its end are zero
    need_h, need_w = pt + oh, pl + ow
    if full.shape[2] < need_h or full.shape[3] < need_w:
        full = F.pad(full, (0, max(0, need_w - full.shape[3]), 0, max(0, need_h - full.shape[2])))
    return full[:, :, pt:pt + oh, pl:pl + ow]


def bias_add_nchw(x, b):
    """tf.nn.bias_add(x, b, data_format='NCHW') (TF/tflib/ops/conv2d.py:120)."""
    return x + b.view(1, -1, *([1] * (x.dim() - 2)))


def moments(x, axes):
    """tf.nn.moments(keep_dims=True): mean and *biased* variance."""
    mean = x.mean(dim=axes, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=axes, keepdim=True)
    return mean, var


def batch_normalization(x, mean, var, offset, scale, eps):
    """tf.nn.batch_normalization: (x-mean)*rsqrt(var+eps)*scale + offset."""
    return (x - mean) * torch.rsqrt(var + eps) * scale + offset


def dropout(x, keep_prob, u):
    """tf.nn.dropout(x, keep_prob) with its uniform draw `u` in [0,1) made an explicit input.

    TF 1.x: `x / keep_prob * floor(keep_prob + uniform)`; identity when keep_prob == 1.
    Call sites: TF/CT_gan_cifar_resnet.py:173-177, TF/CT_gan_cifar.py:86,91,96.
    """
    if keep_prob == 1.0:
        return x
    return x / keep_prob * torch.floor(keep_prob + u)


def mean_pool2(x):
    """tf.add_n of the four stride-2 slices / 4 (TF/CT_gan_cifar_resnet.py:91,96)."""
    return (x[:, :, ::2, ::2] + x[:, :, 1::2, ::2] + x[:, :, ::2, 1::2] + x[:, :, 1::2, 1::2]) / 4.


def upsample2(x):
    """concat x4 on C -> NHWC -> depth_to_space(2) -> NCHW (TF/CT_gan_cifar_resnet.py:102-105).

    Restated literally (block-size-2 depth_to_space on the 4x channel concat); equal to
    nearest-neighbour 2x, which tests assert.
    """
    n, c, h, w = x.shape
    y = torch.cat([x, x, x, x], dim=1)            # [n,4c,h,w]
    y = y.permute(0, 2, 3, 1)                     # NHWC [n,h,w,4c]
    # depth_to_space(2): channel index = (dy*2 + dx)*c + cc
    y = y.reshape(n, h, w, 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * w, c)
    return y.permute(0, 3, 1, 2)


def leaky_relu(x, alpha=0.2):
    """tf.maximum(alpha*x, x) (TF/CT_gan_cifar.py:47-48)."""
    return torch.maximum(alpha * x, x)


def sparse_softmax_ce(logits, labels):
    """tf.nn.sparse_softmax_cross_entropy_with_logits -> per-row loss."""
    return F.cross_entropy(logits, labels.long(), reduction='none')


def tf_adam_step(theta, g, m, v, t, lr, beta1, beta2, eps=1e-8):
    """tf.train.AdamOptimizer update (TF form; eps OUTSIDE the bias correction).

    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    theta -= lr_t * m / (sqrt(v) + eps).   `t` is the 1-based step count.
    Used at TF/CT_gan_cifar_resnet.py:333-338, TF/CT_gan_cifar.py:153-154.
    """
    lr_t = lr * math.sqrt(1. - beta2 ** t) / (1. - beta1 ** t)
    m = beta1 * m + (1. - beta1) * g
    v = beta2 * v + (1. - beta2) * g * g
    theta = theta - lr_t * m / (torch.sqrt(v) + eps)
    return theta, m, v
