"""128x128 ResNet CT-WGAN (config[4]): the nets of
LS/wgan_LSUN_Bedrooms128.py:70-205 (LS = tensorflow_generative_model/LSUN_bedrooms) behind the same
`Generator(n_samples, noise=None)` / `Discriminator(inputs, kp1, kp2, kp3)` surface, driven by the shared
unconditional CT-WGAN step (dcgan_step.DCGANTrainer).

Generator: Linear 128 -> 4*4*DIM_G_4, four 'up' residual blocks (ScaledUpsampleConv, gain 0.5) to 64x64, BN + ReLU,
ScaledUpsampleConv 5x5 to 3x128x128, tanh.  Critic: Conv 5x5 stride 2, three 'down' blocks (conv2 = 3x3 STRIDE 2, shortcut
MeanPoolConv 1x1), dropout, two plain blocks with dropout after each, spatial mean, Linear -> 1.  Layernorm in every critic
block (so the critic is NOT piecewise linear: the gradient penalty differentiates the normalisation twice), BN in the
generator.  fp32 (the fp16-MFMA variant config[4] names is a later round).
"""
from . import functional as F
from . import kernels as K
from .tflib.ops import batchnorm as _bn
from .tflib.ops import conv2d as _conv2d
from .tflib.ops import layernorm as _ln
from .tflib.ops import linear as _linear


class Config:
    """UPPERCASE globals of LS/wgan_LSUN_Bedrooms128.py:27-58."""
    BATCH_SIZE = 64
    DIM_G_64, DIM_G_32, DIM_G_16, DIM_G_8, DIM_G_4 = 64, 128, 256, 512, 512
    DIM_D_64, DIM_D_32, DIM_D_16, DIM_D_8 = 128, 256, 512, 1024
    NORMALIZATION_G = True
    NORMALIZATION_D = True
    ITERS = 200000
    LAMBDA = 10
    LAMBDA_2 = 2.0
    Factor_M = 0.0
    LR = 1e-4
    DECAY = True
    CRITIC_ITERS = 5
    OUTPUT_DIM = 3 * 128 * 128

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(Config, k):
                raise AttributeError('unknown hyper-parameter %s' % k)
            setattr(self, k, v)


cfg = Config()
ADAM_BETAS = (0.0, 0.9)              # MOMENTUM_D = MOMENTUM_G = 0 (:54-55, :289,296)
GEN_TOWERS = 2                       # one Generator(BATCH_SIZE/len(DEVICES)) per device, own BN statistics (:215-218)
PIECEWISE_LINEAR_CRITIC = False      # Layernorm: the GP pass needs its own weight gradients


def configure(**kw):
    global cfg
    cfg = Config(**kw)
    return cfg


def lr(iteration):
    """:285-288"""
    return cfg.LR * (max(0.0, 1.0 - float(iteration) / cfg.ITERS) if cfg.DECAY else 1.0)


def real_prep(real_data_int):
    """:221  2*((int/255.)-.5)"""
    return K.real_prep(real_data_int, None, 255.0)


def feat_shapes():
    """Shapes of the three dropout sites (after blocks 16_3, 8_1, 8_2): all [DIM_D_8, 8, 8]."""
    return [(cfg.DIM_D_8, 8, 8)] * 3


def nonlinearity(x):
    return F.relu(x)


def Normalize(name, inputs, groups=1, relu=False):
    """:70-74"""
    if ('Discriminator' in name) and cfg.NORMALIZATION_D:
        return _ln.Layernorm(name, [1, 2, 3], inputs, relu=relu)      # ReLU fused into the Layernorm kernels
    if ('Generator' in name) and cfg.NORMALIZATION_G:
        return _bn.Batchnorm(name, [0, 2, 3], inputs, fused=True, groups=groups, relu=relu)
    return F.relu(inputs) if relu else inputs        # (the reference returns None here; never reached with both flags on)


def MeanPoolConv(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True):
    """:81-85"""
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, F.mean_pool2(inputs), he_init=he_init, biases=biases)


def ScaledUpsampleConv(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True, out_nchw=False):
    """:87-94  nearest-2x upsample (concat x4 + depth_to_space) then Conv2D with gain 0.5; the upsample is folded into
    the conv (stride-2 transposed conv with the spread filter, or the x_up input gather)."""
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=he_init, biases=biases, gain=0.5,
                          x_up=True, out_nchw=out_nchw)


def ResidualBlock(name, input_dim, output_dim, filter_size, inputs, resample=None, groups=1):
    """:96-135"""
    if resample not in (None, 'down', 'up'):
        raise Exception('invalid resample value')
    if output_dim == input_dim and resample is None:
        shortcut = inputs
    elif resample == 'down':
        shortcut = MeanPoolConv(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
    elif resample == 'up':
        shortcut = ScaledUpsampleConv(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
    else:
        shortcut = _conv2d.Conv2D(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
    out = Normalize(name + '.N1', inputs, groups=groups, relu=True)
    if resample == 'down':
        out = _conv2d.Conv2D(name + '.Conv1', input_dim, input_dim, filter_size, out)
        out = Normalize(name + '.N2', out, groups=groups, relu=True)
        return _conv2d.Conv2D(name + '.Conv2', input_dim, output_dim, filter_size, out, stride=2, resid=shortcut)
    if resample == 'up':
        out = ScaledUpsampleConv(name + '.Conv1', input_dim, output_dim, filter_size, out)
    else:
        out = _conv2d.Conv2D(name + '.Conv1', input_dim, output_dim, filter_size, out)
    out = Normalize(name + '.N2', out, groups=groups, relu=True)
    return _conv2d.Conv2D(name + '.Conv2', output_dim, output_dim, filter_size, out, resid=shortcut)


def Generator(n_samples, noise=None, rng=None, groups=1):
    """ResnetGenerator :137-166.  `groups` > 1 evaluates that many towers (separate BN statistics) at once."""
    if noise is None:
        noise = rng.normal(n_samples, 128)
    out = _linear.Linear('Generator.Input', 128, 4 * 4 * cfg.DIM_G_4, noise)
    out = F.to_channels_last(out.reshape(-1, cfg.DIM_G_4, 4, 4))
    out = ResidualBlock('Generator.4_3', cfg.DIM_G_4, cfg.DIM_G_8, 3, out, resample='up', groups=groups)
    out = ResidualBlock('Generator.8_3', cfg.DIM_G_8, cfg.DIM_G_16, 3, out, resample='up', groups=groups)
    out = ResidualBlock('Generator.16_3', cfg.DIM_G_16, cfg.DIM_G_32, 3, out, resample='up', groups=groups)
    out = ResidualBlock('Generator.32_3', cfg.DIM_G_32, cfg.DIM_G_64, 3, out, resample='up', groups=groups)
    out = Normalize('Generator.OutputN', out, groups=groups, relu=True)
    out = ScaledUpsampleConv('Generator.Output', cfg.DIM_G_64, 3, 5, out, he_init=False, out_nchw=True)
    out = F.tanh(out)
    return out.reshape(-1, cfg.OUTPUT_DIM)


def DiscriminatorTrunk(inputs):
    """Everything before the first dropout (:169-187): input conv + the three 'down' blocks = 5.75 of the critic's 10.55
    GFLOP / sample.  Deterministic and per-sample (Layernorm normalises each sample on its own), so the two dropout passes over the
    real batch of a critic step share ONE evaluation of it (dcgan_step.DCGANTrainer.d_losses)."""
    out = inputs.reshape(-1, 3, 128, 128)
    out = _conv2d.Conv2D('Discriminator.Input', 3, cfg.DIM_D_64, 5, out, he_init=True, stride=2)
    out = ResidualBlock('Discriminator.64_3', cfg.DIM_D_64, cfg.DIM_D_32, 3, out, resample='down')
    out = ResidualBlock('Discriminator.32_3', cfg.DIM_D_32, cfg.DIM_D_16, 3, out, resample='down')
    return ResidualBlock('Discriminator.16_3', cfg.DIM_D_16, cfg.DIM_D_8, 3, out, resample='down')


def DiscriminatorTail(h, kp1=0.8, kp2=0.5, kp3=0.5, u=None, rng=None):
    """dropout -> block -> dropout -> block -> dropout -> mean -> Linear (:188-205) -> (D [n], D_ [n, DIM_D_8])."""
    def drop(i, x, kp):
        if kp == 1.0:
            return x
        return F.dropout(x, kp, u[i]) if u is not None else F.dropout(x, kp, rng=rng)
    out = drop(0, h, kp1)
    out = ResidualBlock('Discriminator.8_1', cfg.DIM_D_8, cfg.DIM_D_8, 3, out, resample=None)
    out = drop(1, out, kp2)
    out = ResidualBlock('Discriminator.8_2', cfg.DIM_D_8, cfg.DIM_D_8, 3, out, resample=None)
    out = drop(2, out, kp3)
    output2 = F.spatial_mean(out)
    out = _linear.Linear('Discriminator.Output', cfg.DIM_D_8, 1, output2)
    return out.reshape(-1), output2


def Discriminator(inputs, kp1=0.8, kp2=0.5, kp3=0.5, u=None, rng=None):
    """ResnetDiscriminator :168-205 -> (D [n], D_ [n, DIM_D_8]).  `u`: the three dropout uniforms [n, DIM_D_8, 8, 8]."""
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
