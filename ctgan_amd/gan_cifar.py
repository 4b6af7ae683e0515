"""DCGAN-style CT-WGAN for CIFAR-10 (1000-example regime): the hot path of TF/CT_gan_cifar.py.
Same `Generator(n_samples, noise=None)` / `Discriminator(inputs)` surface (MODE 'wgan-CT')."""
from . import functional as F
from . import kernels as K
from .tflib.ops import batchnorm as _bn
from .tflib.ops import conv2d as _conv2d
from .tflib.ops import deconv2d as _deconv2d
from .tflib.ops import linear as _linear


class Config:
    """UPPERCASE globals of TF/CT_gan_cifar.py:33-43 (+ the Adam learning rate of :153)."""
    LAMBDA_2 = 2.0
    Factor_M = 0.0
    MODE = 'wgan-CT'
    DIM = 128
    LAMBDA = 10
    CRITIC_ITERS = 5
    BATCH_SIZE = 64
    ITERS = 50000
    OUTPUT_DIM = 3072
    LR = 1e-4

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(Config, k):
                raise AttributeError('unknown hyper-parameter %s' % k)
            setattr(self, k, v)


cfg = Config()


def configure(**kw):
    global cfg
    cfg = Config(**kw)
    return cfg


# the critic's shape for the hand-scheduled step (dcgan_schedule.py): three 5x5 stride-2 convs DIM / 2 DIM / 4 DIM from this input, then Linear
SCHEDULED_CRITIC = {'channels': 3, 'size': 32}


def LeakyReLU(x, alpha=0.2):
    return F.leaky_relu(x, alpha)


def real_prep(real_data_int):
    """:103  2*((int/255.)-.5)"""
    return K.real_prep(real_data_int, None, 255.0)


def feat_shapes():
    D = cfg.DIM
    return [(D, 16, 16), (2 * D, 8, 8), (4 * D, 4, 4)]


def Generator(n_samples, noise=None, rng=None, groups=1):
    """:58-79.  `groups` > 1 (build-only): that many generator calls in one batch, each with its own BatchNorm statistics."""
    D = cfg.DIM
    if noise is None:
        noise = rng.normal(n_samples, 128)
    output = _linear.Linear('Generator.Input', 128, 4 * 4 * 4 * D, noise)
    output = _bn.Batchnorm('Generator.BN1', [0], output, relu=True, groups=groups)
    output = F.to_channels_last(output.reshape(-1, 4 * D, 4, 4))
    output = _deconv2d.Deconv2D('Generator.2', 4 * D, 2 * D, 5, output)
    output = _bn.Batchnorm('Generator.BN2', [0, 2, 3], output, relu=True, groups=groups)
    output = _deconv2d.Deconv2D('Generator.3', 2 * D, D, 5, output)
    output = _bn.Batchnorm('Generator.BN3', [0, 2, 3], output, relu=True, groups=groups)
    output = _deconv2d.Deconv2D('Generator.5', D, 3, 5, output)
    output = F.tanh(F.to_nchw(output))
    return output.reshape(-1, cfg.OUTPUT_DIM)


def Discriminator(inputs, u=None, rng=None):
    """:81-100 - returns (D [n], D_ [n, 4*4*4*DIM]).  `u`: the three dropout uniforms (keep 0.5)."""
    D = cfg.DIM

    def act(x, i):
        """dropout(LeakyReLU(x)), keep 0.5: one launch each way when the mask comes from the Philox stream (F.lrelu_dropout); the two ops
        apart in parity mode (injected uniforms `u`)."""
        if u is not None:
            return F.dropout(LeakyReLU(x), 0.5, u[i])
        if F.LRELU_DROP_FUSION:
            return F.lrelu_dropout(x, 0.2, 0.5, rng)
        return F.dropout(LeakyReLU(x), 0.5, rng=rng)
    output = inputs.reshape(-1, 3, 32, 32)
    output = _conv2d.Conv2D('Discriminator.1', 3, D, 5, output, stride=2)
    output = act(output, 0)
    output = _conv2d.Conv2D('Discriminator.2', D, 2 * D, 5, output, stride=2)
    output = act(output, 1)
    output = _conv2d.Conv2D('Discriminator.3', 2 * D, 4 * D, 5, output, stride=2)
    output = act(output, 2)
    output2 = F.to_nchw(output).reshape(-1, 4 * 4 * 4 * D)
    output = _linear.Linear('Discriminator.Output', 4 * 4 * 4 * D, 1, output2)
    return output.reshape(-1), output2
