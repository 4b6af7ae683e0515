"""tflib.ops.linear - same signature as TF/tflib/ops/linear.py:24-148."""
import numpy as np

from ... import functional as F
from .. import param as _param

_default_weightnorm = False
_weights_stdev = None


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


def disable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = False


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def Linear(name, input_dim, output_dim, inputs, biases=True, initialization=None, weightnorm=None, gain=1.):
    """initialization: None, 'lecun', 'glorot', 'he', 'glorot_he', 'orthogonal', ('uniform', range)"""
    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        raise NotImplementedError('weightnorm is never enabled by the CT scripts')

    def uniform(rng, stdev, size):
        if _weights_stdev is not None:
            stdev = _weights_stdev
        return rng.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')

    def make(rng):
        if initialization == 'lecun':
            return uniform(rng, np.sqrt(1. / input_dim), (input_dim, output_dim))
        if initialization == 'glorot' or initialization is None:      # None never reaches orthogonal (:55)
            return uniform(rng, np.sqrt(2. / (input_dim + output_dim)), (input_dim, output_dim))
        if initialization == 'he':
            return uniform(rng, np.sqrt(2. / input_dim), (input_dim, output_dim))
        if initialization == 'glorot_he':
            return uniform(rng, np.sqrt(4. / (input_dim + output_dim)), (input_dim, output_dim))
        if initialization == 'orthogonal':
            a = rng.normal(0.0, 1.0, (input_dim, output_dim))
            u, _, v = np.linalg.svd(a, full_matrices=False)
            q = u if u.shape == (input_dim, output_dim) else v
            return q.astype('float32')
        if isinstance(initialization, (tuple, list)) and initialization[0] == 'uniform':
            return rng.uniform(low=-initialization[1], high=initialization[1],
                               size=(input_dim, output_dim)).astype('float32')
        raise Exception('Invalid initialization!')

    weight = _param(name + '.W', lambda rng: make(rng) * gain)
    b = _param(name + '.b', lambda rng: np.zeros((output_dim,), dtype='float32')) if biases else None
    if inputs.dim() == 2:
        return F.linear(inputs, weight, b)
    lead = inputs.shape[:-1]
    return F.linear(inputs.reshape(-1, input_dim), weight, b).reshape(*lead, output_dim)
