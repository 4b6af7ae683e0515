"""tflib.ops.deconv2d - same signature as TF/tflib/ops/deconv2d.py:20-115."""
import numpy as np

from ... import functional as F
from .. import param as _param

_default_weightnorm = False
_weights_stdev = None


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def _uniform(rng, stdev, size):
    return rng.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')


def Deconv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, weightnorm=None, biases=True,
             gain=1., mask_type=None):
    """inputs (batch, input_dim, H, W) -> (batch, output_dim, 2H, 2W); stride is hard-coded 2 (:48)."""
    if mask_type is not None:
        raise Exception('Unsupported configuration')
    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        raise NotImplementedError('weightnorm is never enabled by the CT scripts')
    stride = 2
    fan_in = input_dim * filter_size ** 2 / (stride ** 2)
    fan_out = output_dim * filter_size ** 2
    if he_init:
        filters_stdev = np.sqrt(4. / (fan_in + fan_out))
    else:
        filters_stdev = np.sqrt(2. / (fan_in + fan_out))
    stdev = _weights_stdev if _weights_stdev is not None else filters_stdev
    filters = _param(name + '.Filters', lambda rng: _uniform(
        rng, stdev, (filter_size, filter_size, output_dim, input_dim)) * gain)
    b = _param(name + '.Biases', lambda rng: np.zeros(output_dim, dtype='float32')) if biases else None
    return F.conv2d_transpose(inputs, filters, b, stride=stride)
