"""Second, independent restatement of the ResNet critic and its loss scalars in plain numpy - no torch, no autograd.

Oracle cross-check only (test infrastructure; the product never imports it).  oracle/nets.py + oracle/steps.py restate the
reference on torch-CPU ops and lean on torch autograd for the gradient penalty; this file restates the same path a second time
from the reference text, on the numpy tap-loop conv of oracle/np_conv.py, and derives dD/dx_hat BY HAND (the adjoint of every
layer, written out), so that a mistake in either restatement - block wiring, pooling order, dropout scaling, which half the
consistency term sees, the penalty's norm - shows up as a disagreement (tests/test_oracle_anchors.py), and the hand-derived
gradient is itself pinned by central differences of the numpy forward.  With no reference-held vectors and no runnable TF1 this is
the only further pin available for SURVEY 8(c); parity stays "unpinned by the reference".

Follows  TF/CT_gan_cifar_resnet.py:89-98 (ConvMeanPool / MeanPoolConv), :109-141 (ResidualBlock, resample 'down' / None),
:143-153 (OptimizedResBlockDisc1), :169-186 (Discriminator), :244 (wgan), :277-286 (gradient penalty), :288-291 (consistency
term); TF/tflib/ops/conv2d.py:106-120 (SAME conv + bias), linear.py:132-146.  Sizes: small widths only (tap loops).
"""
import numpy as np

from .np_conv import conv2d_same_np


def _conv(P, name, x):
    """lib.ops.conv2d.Conv2D: SAME cross-correlation with the HWIO filter + per-channel bias."""
    y = conv2d_same_np(x, P[name + '.Filters'], 1)
    if (name + '.Biases') in P:
        y = y + P[name + '.Biases'].reshape(1, -1, 1, 1)
    return y


def _conv_T(P, name, gy):
    """Adjoint of the stride-1 SAME conv w.r.t. its input: SAME cross-correlation of gy with the filter rotated by 180 degrees and
    its channel axes swapped (odd filter sizes: the SAME pads are symmetric, so the adjoint needs no crop)."""
    w = P[name + '.Filters']
    wt = np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2))
    return conv2d_same_np(gy, wt, 1)


def _pool(x):
    """(x[::2,::2] + x[1::2,::2] + x[::2,1::2] + x[1::2,1::2]) / 4   (:90-91, :95-96)"""
    return (x[:, :, ::2, ::2] + x[:, :, 1::2, ::2] + x[:, :, ::2, 1::2] + x[:, :, 1::2, 1::2]) / 4.


def _pool_T(g):
    out = np.zeros(g.shape[:2] + (2 * g.shape[2], 2 * g.shape[3]), dtype=g.dtype)
    for a in (0, 1):
        for b in (0, 1):
            out[:, :, a::2, b::2] = g / 4.
    return out


def _drop_mask(u, keep):
    """tf.nn.dropout(x, keep) = x / keep * floor(keep + U); identity for keep == 1."""
    if keep == 1.0 or u is None:
        return None
    return np.floor(keep + u) / keep


def critic(P, x_flat, kps, us, want_grad=False):
    """Discriminator(inputs, labels, kp1, kp2, kp3) -> (D [n], D_ [n,DIM_D]) and, with want_grad, d(sum_n D[n]) / d inputs [n,3072].
    P: name -> numpy array (reference layouts), us: three dropout uniforms [n,DIM_D,8,8] (or None)."""
    n = x_flat.shape[0]
    x = x_flat.reshape(n, 3, 32, 32).astype(np.float64)
    # block 1 (:143-153): shortcut = Conv1x1(pool(x)); out = pool(Conv3x3(relu(Conv3x3(x))))
    a1 = _conv(P, 'Discriminator.1.Conv1', x)
    h1 = _pool(_conv(P, 'Discriminator.1.Conv2', np.maximum(a1, 0.))) + _conv(P, 'Discriminator.1.Shortcut', _pool(x))
    # block 2, resample 'down' (:113-116,128-140): shortcut = pool(Conv1x1(h)); out = pool(Conv3x3(relu(Conv3x3(relu(h)))))
    a2 = _conv(P, 'Discriminator.2.Conv1', np.maximum(h1, 0.))
    h2 = _pool(_conv(P, 'Discriminator.2.Conv2', np.maximum(a2, 0.))) + _pool(_conv(P, 'Discriminator.2.Shortcut', h1))
    m1, m2, m3 = (_drop_mask(u, kp) for u, kp in zip(us if us is not None else (None, None, None), kps))
    d1 = h2 if m1 is None else h2 * m1
    # blocks 3 and 4, resample None, same width: identity shortcut (:128-129)
    a3 = _conv(P, 'Discriminator.3.Conv1', np.maximum(d1, 0.))
    h3 = d1 + _conv(P, 'Discriminator.3.Conv2', np.maximum(a3, 0.))
    d2 = h3 if m2 is None else h3 * m2
    a4 = _conv(P, 'Discriminator.4.Conv1', np.maximum(d2, 0.))
    h4 = d2 + _conv(P, 'Discriminator.4.Conv2', np.maximum(a4, 0.))
    d3 = h4 if m3 is None else h4 * m3
    feat = np.maximum(d3, 0.).mean(axis=(2, 3))                                   # tf.reduce_mean(output, axis=[2,3]) :180
    D = feat @ P['Discriminator.Output.W'] + P['Discriminator.Output.b']          # Linear(DIM_D, 1) :181
    D = D.reshape(-1)
    if not want_grad:
        return D, feat
    # ---- d(sum D)/dx, layer by layer, in reverse
    hw = d3.shape[2] * d3.shape[3]
    g = np.broadcast_to((P['Discriminator.Output.W'].reshape(1, -1, 1, 1) / hw), d3.shape) * (d3 > 0)
    g = g if m3 is None else g * m3                                               # -> dh4
    g = g + _conv_T(P, 'Discriminator.4.Conv1', _conv_T(P, 'Discriminator.4.Conv2', g) * (a4 > 0)) * (d2 > 0)     # -> dd2
    g = g if m2 is None else g * m2                                               # -> dh3
    g = g + _conv_T(P, 'Discriminator.3.Conv1', _conv_T(P, 'Discriminator.3.Conv2', g) * (a3 > 0)) * (d1 > 0)     # -> dd1
    g = g if m1 is None else g * m1                                               # -> dh2
    gh1 = (_conv_T(P, 'Discriminator.2.Conv1', _conv_T(P, 'Discriminator.2.Conv2', _pool_T(g)) * (a2 > 0)) * (h1 > 0)
           + _conv_T(P, 'Discriminator.2.Shortcut', _pool_T(g)))
    gx = (_conv_T(P, 'Discriminator.1.Conv1', _conv_T(P, 'Discriminator.1.Conv2', _pool_T(gh1)) * (a1 > 0))
          + _pool_T(_conv_T(P, 'Discriminator.1.Shortcut', gh1)))
    return D, feat, gx.reshape(n, -1)


def critic_scalars(P, real, fake, rnd, B, lambda2=2.0, factor_m=0.0, gp_lambda=10.0):
    """The WGAN / consistency / gradient-penalty scalars of the critic loss (:244, :277-291) for given real [B,3072] (already
    dequantised) and fake [B,3072] batches and the draws rnd = {'alpha' [B,1], 'u_pass1', 'u_pass2' (3 x [2B,DIM_D,8,8]),
    'u_gp' (3 x [B,DIM_D,8,8])}.  -> dict(wgan, ct, gp, slopes, d_real, d_fake)."""
    rf = np.concatenate([real, fake], 0)
    kps = (0.8, 0.5, 0.5)
    d1, f1 = critic(P, rf, kps, rnd['u_pass1'])
    d2, f2 = critic(P, rf, kps, rnd['u_pass2'])
    wgan = d1[B:].mean() - d1[:B].mean()                                          # :244 (fake minus real)
    ct_i = lambda2 * (d1[:B] - d2[:B]) ** 2 + lambda2 * 0.1 * ((f1[:B] - f2[:B]) ** 2).mean(axis=1)      # :288-289, real half only
    ct = np.maximum(ct_i - factor_m, 0.).mean()                                   # :290-291
    interp = real + rnd['alpha'] * (fake - real)                                  # :277-283
    _, _, gx = critic(P, interp, kps, rnd['u_gp'], want_grad=True)
    slopes = np.sqrt((gx ** 2).sum(axis=1))                                       # :285
    gp = gp_lambda * ((slopes - 1.) ** 2).mean()                                  # :286
    return {'wgan': wgan, 'ct': ct, 'gp': gp, 'slopes': slopes, 'd_real': d1[:B], 'd_fake': d1[B:], 'gp_grads': gx}
