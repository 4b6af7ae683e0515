"""tflib.ops.cond_batchnorm - same signature as TF/tflib/ops/cond_batchnorm.py:6-17."""
import numpy as np

from ... import functional as F
from .. import param as _param


def Batchnorm(name, axes, inputs, is_training=None, stats_iter=None, update_moving_stats=True, fused=True,
              labels=None, n_labels=None, groups=1, relu=False):
    """conditional batchnorm (dumoulin et al 2016) for BCHW conv filtermaps"""
    if axes != [0, 2, 3]:
        raise Exception('unsupported')
    C = inputs.shape[1]
    offset_m = _param(name + '.offset', lambda rng: np.zeros([n_labels, C], dtype='float32'))
    scale_m = _param(name + '.scale', lambda rng: np.ones([n_labels, C], dtype='float32'))
    return F.batch_norm(inputs, scale_m, offset_m, labels, groups, relu)
