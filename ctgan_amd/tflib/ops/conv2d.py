"""tflib.ops.conv2d - same signature as TF/tflib/ops/conv2d.py:20-123, HIP kernels underneath."""
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


def Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, mask_type=None, stride=1,
           weightnorm=None, biases=True, gain=1., **fuse):
    """inputs: (batch, channels, height, width) logical NCHW -> (batch, output_dim, ceil(H/stride), ceil(W/stride)).

    `**fuse` (build-only): resid=tensor added in the conv epilogue, x_up=True reads the input through
    a nearest 2x upsample, out_nchw=True writes an NCHW-contiguous result, relu_in=True computes
    Conv2D(relu(inputs)) without materialising the ReLU (the pre-activation blocks of the critic), fork=True also
    returns the input (residual blocks), epi={...} fuses the neighbouring dropout / ReLU into the conv kernels
    (functional.ConvFn).
    """
    if mask_type is not None:
        raise NotImplementedError('masked convolutions are never enabled by the CT scripts')
    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        raise NotImplementedError('weightnorm is never enabled by the CT scripts')
    fan_in = input_dim * filter_size ** 2
    fan_out = output_dim * filter_size ** 2 / (stride ** 2)
    if he_init:
        filters_stdev = np.sqrt(4. / (fan_in + fan_out))
    else:
        filters_stdev = np.sqrt(2. / (fan_in + fan_out))
    stdev = _weights_stdev if _weights_stdev is not None else filters_stdev
    filters = _param(name + '.Filters', lambda rng: _uniform(
        rng, stdev, (filter_size, filter_size, input_dim, output_dim)) * gain)
    b = _param(name + '.Biases', lambda rng: np.zeros(output_dim, dtype='float32')) if biases else None
    return F.conv2d(inputs, filters, b, stride=stride, **fuse)
