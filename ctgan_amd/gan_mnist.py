"""DCGAN-style CT-WGAN for MNIST (1000-example regime): the hot path of TF/CT_gan_mnist.py,
MODE 'wgan-CT' (no batch norm).  `Generator(n_samples, noise=None)` / `Discriminator(inputs)`."""
from . import functional as F
from .tflib.ops import conv2d as _conv2d
from .tflib.ops import deconv2d as _deconv2d
from .tflib.ops import linear as _linear


class Config:
    """UPPERCASE globals of TF/CT_gan_mnist.py:26-36 (+ the Adam learning rate of :170)."""
    Factor_M = 0.0
    LAMBDA_2 = 2.0
    MODE = 'wgan-CT'
    DIM = 64
    BATCH_SIZE = 50
    CRITIC_ITERS = 5
    LAMBDA = 10
    ITERS = 50000
    OUTPUT_DIM = 784
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
SCHEDULED_CRITIC = {'channels': 1, 'size': 28}


def LeakyReLU(x, alpha=0.2):
    return F.leaky_relu(x, alpha)


def real_prep(real_data):
    """real_data is fed as float32 in [0,1] (:110)."""
    return real_data


def feat_shapes():
    D = cfg.DIM
    return [(D, 14, 14), (2 * D, 7, 7), (4 * D, 4, 4)]


def Generator(n_samples, noise=None, rng=None, groups=1):
    """:62-87  (`groups`: accepted for the batched fake draws of dcgan_step - this generator has no batch statistics.)"""
    D = cfg.DIM
    if noise is None:
        noise = rng.normal(n_samples, 128)
    output = _linear.Linear('Generator.Input', 128, 4 * 4 * 4 * D, noise)
    output = F.relu(output)
    output = F.to_channels_last(output.reshape(-1, 4 * D, 4, 4))
    output = _deconv2d.Deconv2D('Generator.2', 4 * D, 2 * D, 5, output)
    output = F.relu(output)
    output = F.crop(output, 7, 7)                                   # output[:,:,:7,:7]
    output = _deconv2d.Deconv2D('Generator.3', 2 * D, D, 5, output)
    output = F.relu(output)
    output = _deconv2d.Deconv2D('Generator.5', D, 1, 5, output)
    output = F.sigmoid(F.to_nchw(output))
    return output.reshape(-1, cfg.OUTPUT_DIM)


def Discriminator(inputs, u=None, rng=None):
    """:89-108 - returns (D [n], D_ [n, 4*4*4*DIM])."""
    D = cfg.DIM

    def act(x, i):
        """dropout(LeakyReLU(x)), keep 0.5: one launch each way when the mask comes from the Philox stream (F.lrelu_dropout); the two ops
        apart in parity mode (injected uniforms `u`)."""
        if u is not None:
            return F.dropout(LeakyReLU(x), 0.5, u[i])
        if F.LRELU_DROP_FUSION:
            return F.lrelu_dropout(x, 0.2, 0.5, rng)
        return F.dropout(LeakyReLU(x), 0.5, rng=rng)
    output = inputs.reshape(-1, 1, 28, 28)
    output = _conv2d.Conv2D('Discriminator.1', 1, D, 5, output, stride=2)
    output = act(output, 0)
    output = _conv2d.Conv2D('Discriminator.2', D, 2 * D, 5, output, stride=2)
    output = act(output, 1)
    output = _conv2d.Conv2D('Discriminator.3', 2 * D, 4 * D, 5, output, stride=2)
    output = act(output, 2)
    output2 = F.to_nchw(output).reshape(-1, 4 * 4 * 4 * D)
    output = _linear.Linear('Discriminator.Output', 4 * 4 * 4 * D, 1, output2)
    return output.reshape(-1), output2
