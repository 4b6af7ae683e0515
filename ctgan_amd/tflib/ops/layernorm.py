"""tflib.ops.layernorm - same signature as TF/tflib/ops/layernorm.py:6-20, HIP kernels underneath.

Used by the layer-normalised critics (config[4], LS/wgan_LSUN_Bedrooms128.py:70-72; TF/CT_gan_64x64.py).  The
gradient penalty differentiates the critic twice, so the operator is composed of kernel-backed maps that are closed
under differentiation (functional.layer_norm).
"""
import numpy as np

from ... import functional as F
from .. import param as _param


def Layernorm(name, norm_axes, inputs, relu=False):
    """inputs [N,C,H,W] (norm_axes [1,2,3]) or [N,C] (norm_axes [1]): per-sample moments over norm_axes, then
    `name.scale` / `name.offset` of size C (the first normalised axis, :10-13), eps 1e-5.  `relu` (build-only): also apply the
    ReLU that follows the normalisation in the critics' blocks, in the same kernels."""
    norm_axes = list(norm_axes)
    if norm_axes != list(range(1, inputs.dim())):
        raise NotImplementedError('Layernorm over axes %s of a %d-D tensor (the CT scripts use all non-batch axes)'
                                  % (norm_axes, inputs.dim()))
    n_neurons = inputs.shape[norm_axes[0]]
    offset = _param(name + '.offset', lambda rng: np.zeros(n_neurons, dtype='float32'))
    scale = _param(name + '.scale', lambda rng: np.ones(n_neurons, dtype='float32'))
    return F.layer_norm(inputs, scale, offset, 1e-5, relu=relu)
