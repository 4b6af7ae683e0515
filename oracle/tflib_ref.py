"""Restatement of the reference operator library `tflib.ops.*` + `lib.param` (oracle; tests only).

Follows TF/tflib/__init__.py:10-40 (registry) and TF/tflib/ops/{conv2d,deconv2d,linear,batchnorm,
cond_batchnorm,layernorm}.py.  Weights live in a name-keyed `Registry`; layouts are the
reference's (Filters HWIO for Conv2D, [k,k,out,in] for Deconv2D, W [in,out]).
"""
import zlib

import numpy as np
import torch

from . import tf_ops


class Registry(dict):
    """name -> leaf tensor; create-once / share-after semantics of lib.param
    (TF/tflib/__init__.py:10-34).  `trainable` mirrors tf.Variable(trainable=False) for the
    moving statistics (TF/tflib/ops/batchnorm.py:26-27)."""

    def __init__(self, dtype=torch.float64, seed=0):
        super().__init__()
        self.dtype = dtype
        self.seed = seed
        self.non_trainable = set()

    def rng_for(self, name):
        # one independent stream per parameter name: creation order does not matter
        return np.random.default_rng([self.seed, zlib.crc32(name.encode())])

    def param(self, name, make, trainable=True):
        if name not in self:
            val = torch.as_tensor(np.asarray(make(self.rng_for(name)), dtype=np.float32)).to(self.dtype)
            val.requires_grad_(trainable)
            self[name] = val
            if not trainable:
                self.non_trainable.add(name)
        return self[name]

    def params_with_name(self, sub):
        """TF/tflib/__init__.py:36-37 (substring match, includes non-trainables)."""
        return [(n, p) for n, p in self.items() if sub in n]

    def trainable_with_name(self, sub):
        return [(n, p) for n, p in self.items() if sub in n and n not in self.non_trainable]


def _uniform(rng, stdev, size):
    # TF/tflib/ops/conv2d.py:55-60: U(+-stdev*sqrt(3)) as float32
    return rng.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')


def Conv2D(reg, name, input_dim, output_dim, filter_size, inputs, he_init=True, mask_type=None,
           stride=1, weightnorm=None, biases=True, gain=1.):
    """TF/tflib/ops/conv2d.py:20-123."""
    if mask_type is not None or weightnorm:
        raise NotImplementedError('mask_type / weightnorm are never enabled by the CT scripts')
    fan_in = input_dim * filter_size ** 2
    fan_out = output_dim * filter_size ** 2 / (stride ** 2)
    stdev = np.sqrt(4. / (fan_in + fan_out)) if he_init else np.sqrt(2. / (fan_in + fan_out))
    filters = reg.param(name + '.Filters', lambda rng: _uniform(
        rng, stdev, (filter_size, filter_size, input_dim, output_dim)) * gain)
    result = tf_ops.conv2d_same(inputs, filters, stride)
    if biases:
        b = reg.param(name + '.Biases', lambda rng: np.zeros(output_dim, dtype='float32'))
        result = tf_ops.bias_add_nchw(result, b)
    return result


def Deconv2D(reg, name, input_dim, output_dim, filter_size, inputs, he_init=True, weightnorm=None,
             biases=True, gain=1., mask_type=None):
    """TF/tflib/ops/deconv2d.py:20-115 (stride hard-coded 2, :48)."""
    if mask_type is not None:
        raise Exception('Unsupported configuration')   # TF/tflib/ops/deconv2d.py:38-39
    if weightnorm:
        raise NotImplementedError
    stride = 2
    fan_in = input_dim * filter_size ** 2 / (stride ** 2)
    fan_out = output_dim * filter_size ** 2
    stdev = np.sqrt(4. / (fan_in + fan_out)) if he_init else np.sqrt(2. / (fan_in + fan_out))
    filters = reg.param(name + '.Filters', lambda rng: _uniform(
        rng, stdev, (filter_size, filter_size, output_dim, input_dim)) * gain)
    result = tf_ops.conv2d_transpose_same(inputs, filters, stride)
    if biases:
        b = reg.param(name + '.Biases', lambda rng: np.zeros(output_dim, dtype='float32'))
        result = tf_ops.bias_add_nchw(result, b)
    return result


def Linear(reg, name, input_dim, output_dim, inputs, biases=True, initialization=None,
           weightnorm=None, gain=1.):
    """TF/tflib/ops/linear.py:24-148.  `None` takes the glorot branch (:55) - the orthogonal
    test at :76-77 is unreachable for None."""
    if weightnorm:
        raise NotImplementedError
    if initialization == 'lecun':
        stdev = np.sqrt(1. / input_dim)
    elif initialization == 'glorot' or initialization is None:
        stdev = np.sqrt(2. / (input_dim + output_dim))
    elif initialization == 'he':
        stdev = np.sqrt(2. / input_dim)
    elif initialization == 'glorot_he':
        stdev = np.sqrt(4. / (input_dim + output_dim))
    else:
        raise Exception('Invalid initialization!')
    W = reg.param(name + '.W', lambda rng: _uniform(rng, stdev, (input_dim, output_dim)) * gain)
    result = inputs.reshape(-1, input_dim) @ W
    result = result.reshape(*inputs.shape[:-1], output_dim)
    if biases:
        b = reg.param(name + '.b', lambda rng: np.zeros((output_dim,), dtype='float32'))
        result = result + b
    return result


def Batchnorm(reg, name, axes, inputs, is_training=None, stats_iter=None, update_moving_stats=True,
              fused=True):
    """TF/tflib/ops/batchnorm.py:6-87.  CT scripts always pass is_training=None -> training-mode
    statistics (biased variance, eps 1e-5); moving stats are created but never updated."""
    if is_training is not None:
        raise NotImplementedError('inference / moving-stat branches are unreachable from CT scripts')
    if (axes == [0, 2, 3] or axes == [0, 2]) and fused:
        x = inputs.unsqueeze(3) if axes == [0, 2] else inputs
        C = x.shape[1]
        offset = reg.param(name + '.offset', lambda rng: np.zeros(C, dtype='float32'))
        scale = reg.param(name + '.scale', lambda rng: np.ones(C, dtype='float32'))
        reg.param(name + '.moving_mean', lambda rng: np.zeros(C, dtype='float32'), trainable=False)
        reg.param(name + '.moving_variance', lambda rng: np.ones(C, dtype='float32'), trainable=False)
        mean, var = tf_ops.moments(x, [0, 2, 3])
        out = tf_ops.batch_normalization(x, mean, var, offset.view(1, -1, 1, 1), scale.view(1, -1, 1, 1), 1e-5)
        return out[:, :, :, 0] if axes == [0, 2] else out
    mean, var = tf_ops.moments(inputs, axes)
    shape = list(mean.shape)
    if 0 not in axes:
        shape[0] = 1
    offset = reg.param(name + '.offset', lambda rng: np.zeros(shape, dtype='float32'))
    scale = reg.param(name + '.scale', lambda rng: np.ones(shape, dtype='float32'))
    return tf_ops.batch_normalization(inputs, mean, var, offset, scale, 1e-5)


def CondBatchnorm(reg, name, axes, inputs, labels=None, n_labels=None):
    """TF/tflib/ops/cond_batchnorm.py:6-17."""
    if axes != [0, 2, 3]:
        raise Exception('unsupported')
    mean, var = tf_ops.moments(inputs, axes)
    C = inputs.shape[1]
    offset_m = reg.param(name + '.offset', lambda rng: np.zeros([n_labels, C], dtype='float32'))
    scale_m = reg.param(name + '.scale', lambda rng: np.ones([n_labels, C], dtype='float32'))
    offset = offset_m[labels.long()]
    scale = scale_m[labels.long()]
    return tf_ops.batch_normalization(inputs, mean, var, offset[:, :, None, None], scale[:, :, None, None], 1e-5)


def Layernorm(reg, name, norm_axes, inputs):
    """TF/tflib/ops/layernorm.py:6-20."""
    mean, var = tf_ops.moments(inputs, norm_axes)
    n_neurons = inputs.shape[norm_axes[0]]
    offset = reg.param(name + '.offset', lambda rng: np.zeros(n_neurons, dtype='float32'))
    scale = reg.param(name + '.scale', lambda rng: np.ones(n_neurons, dtype='float32'))
    shp = [-1] + [1] * (len(norm_axes) - 1)
    return tf_ops.batch_normalization(inputs, mean, var, offset.view(*shp), scale.view(*shp), 1e-5)
