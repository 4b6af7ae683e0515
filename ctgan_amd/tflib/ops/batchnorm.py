"""tflib.ops.batchnorm - same signature as TF/tflib/ops/batchnorm.py:6-87.

The CT scripts always call it with is_training=None, i.e. training-mode batch statistics; the
inference / moving-average branches (:31-37,53-68) are unreachable from them and raise here.
Build-only kwargs: `groups` (independent statistic groups = the reference's per-tower batches),
`relu` (fuse the ReLU that always follows in the generators).
"""
import numpy as np

from ... import functional as F
from .. import param as _param


def Batchnorm(name, axes, inputs, is_training=None, stats_iter=None, update_moving_stats=True, fused=True,
              groups=1, relu=False):
    if is_training is not None:
        raise NotImplementedError('Batchnorm(is_training=...) is unreachable from the CT scripts')
    if ((axes == [0, 2, 3]) or (axes == [0, 2])) and fused:
        x = inputs.unsqueeze(3) if axes == [0, 2] else inputs
        C = x.shape[1]
        offset = _param(name + '.offset', lambda rng: np.zeros(C, dtype='float32'))
        scale = _param(name + '.scale', lambda rng: np.ones(C, dtype='float32'))
        _param(name + '.moving_mean', lambda rng: np.zeros(C, dtype='float32'), trainable=False)
        _param(name + '.moving_variance', lambda rng: np.ones(C, dtype='float32'), trainable=False)
        out = F.batch_norm(x, scale.view(1, C), offset.view(1, C), None, groups, relu)
        return out[:, :, :, 0] if axes == [0, 2] else out
    if axes == [0] and inputs.dim() == 2:
        C = inputs.shape[1]
        offset = _param(name + '.offset', lambda rng: np.zeros([1, C], dtype='float32'))   # moments' shape (:78-83)
        scale = _param(name + '.scale', lambda rng: np.ones([1, C], dtype='float32'))
        return F.batch_norm(inputs, scale, offset, None, groups, relu)
    raise NotImplementedError('Batchnorm axes %s: only [0,2,3], [0,2] and [0] (2-D input) are used' % (axes,))
