"""64x64 CT-WGAN: `GoodGenerator` / `GoodDiscriminator` of TF/CT_gan_64x64.py:166-221,357-373
(MODE 'wgan-ct', the only architecture pair the script selects, :48) on the shared unconditional CT-WGAN step
(dcgan_step.DCGANTrainer): generator with batch norm, critic with Layernorm (:87-92) - so the gradient penalty
differentiates the normalisation twice - Adam(1e-4, beta1 = 0, beta2 = 0.9) without decay (:561-565)."""
from . import functional as F
from . import kernels as K
from .tflib.ops import batchnorm as _bn
from .tflib.ops import conv2d as _conv2d
from .tflib.ops import layernorm as _ln
from .tflib.ops import linear as _linear


class Config:
    """UPPERCASE globals of TF/CT_gan_64x64.py:27-37."""
    LAMBDA_2 = 2.0
    Factor_M = 0.0
    MODE = 'wgan-ct'
    DIM = 64
    CRITIC_ITERS = 5
    BATCH_SIZE = 64
    ITERS = 200000
    LAMBDA = 10
    OUTPUT_DIM = 64 * 64 * 3
    LR = 1e-4

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(Config, k):
                raise AttributeError('unknown hyper-parameter %s' % k)
            setattr(self, k, v)


cfg = Config()
ADAM_BETAS = (0.0, 0.9)
GEN_TOWERS = 1
PIECEWISE_LINEAR_CRITIC = False


def configure(**kw):
    global cfg
    cfg = Config(**kw)
    return cfg


def real_prep(real_data_int):
    """:483  2*((int/255.)-.5)"""
    return K.real_prep(real_data_int, None, 255.0)


def feat_shapes():
    """Dropout sites: after Res2 [4*DIM,16,16], Res3 [8*DIM,8,8], Res4 [8*DIM,4,4] (:362-367)."""
    D = cfg.DIM
    return [(4 * D, 16, 16), (8 * D, 8, 8), (8 * D, 4, 4)]


def Normalize(name, axes, inputs, relu=False, groups=1):
    """:87-92 (+ build-only `groups`: independent BatchNorm statistic groups of a batched generator forward)"""
    if ('Discriminator' in name) and (cfg.MODE == 'wgan-ct'):
        if axes != [0, 2, 3]:
            raise Exception('Layernorm over non-standard axes is unsupported')
        return _ln.Layernorm(name, [1, 2, 3], inputs, relu=relu)      # ReLU fused into the Layernorm kernels
    return _bn.Batchnorm(name, axes, inputs, fused=True, relu=relu, groups=groups)


def ConvMeanPool(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True, resid=None):
    """:107-110 (conv + mean pool = one stride-2 conv with the spread filter when the channel counts allow)"""
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=he_init, biases=biases, pool=True, resid=resid)


def MeanPoolConv(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True):
    """:112-116"""
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, F.mean_pool2(inputs), he_init=he_init, biases=biases)


def UpsampleConv(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True):
    """:118-125"""
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=he_init, biases=biases, x_up=True)


def ResidualBlock(name, input_dim, output_dim, filter_size, inputs, resample=None, he_init=True, groups=1):
    """:127-162 (Conv1 has no bias, :157)"""
    if resample not in (None, 'down', 'up'):
        raise Exception('invalid resample value')
    if output_dim == input_dim and resample is None:
        shortcut = inputs
    elif resample == 'down':
        shortcut = MeanPoolConv(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
    elif resample == 'up':
        shortcut = UpsampleConv(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
    else:
        shortcut = _conv2d.Conv2D(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
    out = Normalize(name + '.BN1', [0, 2, 3], inputs, relu=True, groups=groups)
    if resample == 'up':
        out = UpsampleConv(name + '.Conv1', input_dim, output_dim, filter_size, out, he_init=he_init, biases=False)
        out = Normalize(name + '.BN2', [0, 2, 3], out, relu=True, groups=groups)
        return _conv2d.Conv2D(name + '.Conv2', output_dim, output_dim, filter_size, out, he_init=he_init, resid=shortcut)
    out = _conv2d.Conv2D(name + '.Conv1', input_dim, input_dim, filter_size, out, he_init=he_init, biases=False)
    out = Normalize(name + '.BN2', [0, 2, 3], out, relu=True, groups=groups)
    if resample == 'down':
        return ConvMeanPool(name + '.Conv2', input_dim, output_dim, filter_size, out, he_init=he_init, resid=shortcut)
    return _conv2d.Conv2D(name + '.Conv2', input_dim, output_dim, filter_size, out, he_init=he_init, resid=shortcut)


def Generator(n_samples, noise=None, rng=None, groups=1):
    """GoodGenerator :204-221.  `groups` > 1 (build-only): that many generator calls in one batch, each with its own BatchNorm
    statistics (dcgan_step.DCGANTrainer.generate_fakes)."""
    dim = cfg.DIM
    if noise is None:
        noise = rng.normal(n_samples, 128)
    out = _linear.Linear('Generator.Input', 128, 4 * 4 * 8 * dim, noise)
    out = F.to_channels_last(out.reshape(-1, 8 * dim, 4, 4))
    out = ResidualBlock('Generator.Res1', 8 * dim, 8 * dim, 3, out, resample='up', groups=groups)
    out = ResidualBlock('Generator.Res2', 8 * dim, 4 * dim, 3, out, resample='up', groups=groups)
    out = ResidualBlock('Generator.Res3', 4 * dim, 2 * dim, 3, out, resample='up', groups=groups)
    out = ResidualBlock('Generator.Res4', 2 * dim, 1 * dim, 3, out, resample='up', groups=groups)
    out = Normalize('Generator.OutputN', [0, 2, 3], out, relu=True, groups=groups)
    out = _conv2d.Conv2D('Generator.Output', 1 * dim, 3, 3, out, out_nchw=True)
    out = F.tanh(out)
    return out.reshape(-1, cfg.OUTPUT_DIM)


def critic_is_per_sample():
    """Layernorm (MODE 'wgan-ct') normalises each sample on its own; a batch-normalised critic couples the rows of a batch."""
    return cfg.MODE == 'wgan-ct'


def DiscriminatorTrunk(inputs):
    """Input conv + Res1 + Res2: everything before the first dropout (:358-363); deterministic and per-sample, shared by the two
    dropout passes over the real batch of a critic step (dcgan_step.DCGANTrainer.d_losses)."""
    dim = cfg.DIM
    out = inputs.reshape(-1, 3, 64, 64)
    out = _conv2d.Conv2D('Discriminator.Input', 3, dim, 3, out, he_init=False)
    out = ResidualBlock('Discriminator.Res1', dim, 2 * dim, 3, out, resample='down')
    return ResidualBlock('Discriminator.Res2', 2 * dim, 4 * dim, 3, out, resample='down')


def DiscriminatorTail(h, kp1=0.8, kp2=0.5, kp3=0.5, u=None, rng=None):
    """dropout -> Res3 -> dropout -> Res4 -> dropout -> Linear (:364-373)."""
    dim = cfg.DIM

    def drop(i, x, kp):
        if kp == 1.0:
            return x
        return F.dropout(x, kp, u[i]) if u is not None else F.dropout(x, kp, rng=rng)
    out = drop(0, h, kp1)
    out = ResidualBlock('Discriminator.Res3', 4 * dim, 8 * dim, 3, out, resample='down')
    out = drop(1, out, kp2)
    out = ResidualBlock('Discriminator.Res4', 8 * dim, 8 * dim, 3, out, resample='down')
    out = drop(2, out, kp3)
    output2 = F.to_nchw(out).reshape(-1, 4 * 4 * 8 * dim)
    out = _linear.Linear('Discriminator.Output', 4 * 4 * 8 * dim, 1, output2)
    return out.reshape(-1), output2


def Discriminator(inputs, kp1=0.8, kp2=0.5, kp3=0.5, u=None, rng=None):
    """GoodDiscriminator :357-373 -> (D [n], D_ [n, 4*4*8*DIM])."""
    return DiscriminatorTail(DiscriminatorTrunk(inputs), kp1, kp2, kp3, u=u, rng=rng)


def build_params(device=None):
    import torch
    from . import tflib as lib
    if device is not None:
        lib.set_device(device)
    dev = lib._dev()
    with torch.no_grad():
        x = Generator(2, noise=torch.zeros(2, 128, device=dev))
        Discriminator(x, 1.0, 1.0, 1.0)
