"""Generator / Discriminator of the three north-star scripts, restated on the oracle ops.

Test infrastructure only.  Every random draw the reference makes inside these functions
(tf.random_normal noise, tf.nn.dropout masks) is an explicit argument.
  resnet : TF/CT_gan_cifar_resnet.py:67-186
  cifar  : TF/CT_gan_cifar.py:47-100     (DCGAN, MODE 'wgan-CT')
  mnist  : TF/CT_gan_mnist.py:39-108     (DCGAN, MODE 'wgan-CT')
"""
import functools

import torch

from . import tf_ops
from . import tflib_ref as ops


# ----------------------------------------------------------------------------- ResNet (CIFAR)
class ResnetCfg:
    """UPPERCASE globals of TF/CT_gan_cifar_resnet.py:33-56 that shape the nets."""
    def __init__(self, DIM_G=128, DIM_D=128, OUTPUT_DIM=3072, CONDITIONAL=True, ACGAN=True,
                 NORMALIZATION_G=True, NORMALIZATION_D=False):
        self.DIM_G, self.DIM_D, self.OUTPUT_DIM = DIM_G, DIM_D, OUTPUT_DIM
        self.CONDITIONAL, self.ACGAN = CONDITIONAL, ACGAN
        self.NORMALIZATION_G, self.NORMALIZATION_D = NORMALIZATION_G, NORMALIZATION_D


def Normalize(reg, cfg, name, inputs, labels=None):
    """TF/CT_gan_cifar_resnet.py:70-87."""
    if not cfg.CONDITIONAL:
        labels = None
    if cfg.CONDITIONAL and cfg.ACGAN and ('Discriminator' in name):
        labels = None
    if ('Discriminator' in name) and cfg.NORMALIZATION_D:
        return ops.Layernorm(reg, name, [1, 2, 3], inputs)
    elif ('Generator' in name) and cfg.NORMALIZATION_G:
        if labels is not None:
            return ops.CondBatchnorm(reg, name, [0, 2, 3], inputs, labels=labels, n_labels=10)
        return ops.Batchnorm(reg, name, [0, 2, 3], inputs, fused=True)
    return inputs


def ConvMeanPool(reg, name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True):
    """:89-92"""
    out = ops.Conv2D(reg, name, input_dim, output_dim, filter_size, inputs, he_init=he_init, biases=biases)
    return tf_ops.mean_pool2(out)


def MeanPoolConv(reg, name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True):
    """:94-98"""
    out = tf_ops.mean_pool2(inputs)
    return ops.Conv2D(reg, name, input_dim, output_dim, filter_size, out, he_init=he_init, biases=biases)


def UpsampleConv(reg, name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True):
    """:100-107"""
    out = tf_ops.upsample2(inputs)
    return ops.Conv2D(reg, name, input_dim, output_dim, filter_size, out, he_init=he_init, biases=biases)


def ResidualBlock(reg, cfg, name, input_dim, output_dim, filter_size, inputs, resample=None, labels=None):
    """:109-141"""
    if resample == 'down':
        conv_1 = functools.partial(ops.Conv2D, reg, input_dim=input_dim, output_dim=input_dim)
        conv_2 = functools.partial(ConvMeanPool, reg, input_dim=input_dim, output_dim=output_dim)
        conv_shortcut = functools.partial(ConvMeanPool, reg)
    elif resample == 'up':
        conv_1 = functools.partial(UpsampleConv, reg, input_dim=input_dim, output_dim=output_dim)
        conv_shortcut = functools.partial(UpsampleConv, reg)
        conv_2 = functools.partial(ops.Conv2D, reg, input_dim=output_dim, output_dim=output_dim)
    elif resample is None:
        conv_shortcut = functools.partial(ops.Conv2D, reg)
        conv_1 = functools.partial(ops.Conv2D, reg, input_dim=input_dim, output_dim=output_dim)
        conv_2 = functools.partial(ops.Conv2D, reg, input_dim=output_dim, output_dim=output_dim)
    else:
        raise Exception('invalid resample value')

    if output_dim == input_dim and resample is None:
        shortcut = inputs
    else:
        shortcut = conv_shortcut(name + '.Shortcut', input_dim=input_dim, output_dim=output_dim,
                                 filter_size=1, he_init=False, biases=True, inputs=inputs)
    out = inputs
    out = Normalize(reg, cfg, name + '.N1', out, labels=labels)
    out = torch.relu(out)
    out = conv_1(name + '.Conv1', filter_size=filter_size, inputs=out)
    out = Normalize(reg, cfg, name + '.N2', out, labels=labels)
    out = torch.relu(out)
    out = conv_2(name + '.Conv2', filter_size=filter_size, inputs=out)
    return shortcut + out


def OptimizedResBlockDisc1(reg, cfg, inputs):
    """:143-153"""
    D = cfg.DIM_D
    shortcut = MeanPoolConv(reg, 'Discriminator.1.Shortcut', input_dim=3, output_dim=D, filter_size=1,
                            he_init=False, biases=True, inputs=inputs)
    out = ops.Conv2D(reg, 'Discriminator.1.Conv1', 3, D, 3, inputs)
    out = torch.relu(out)
    out = ConvMeanPool(reg, 'Discriminator.1.Conv2', D, D, 3, out)
    return shortcut + out


def resnet_generator(reg, cfg, n_samples, labels, noise):
    """Generator(n_samples, labels, noise) :155-167.  `noise` [n,128] is required (explicit draw)."""
    G = cfg.DIM_G
    out = ops.Linear(reg, 'Generator.Input', 128, 4 * 4 * G, noise)
    out = out.reshape(-1, G, 4, 4)
    out = ResidualBlock(reg, cfg, 'Generator.1', G, G, 3, out, resample='up', labels=labels)
    out = ResidualBlock(reg, cfg, 'Generator.2', G, G, 3, out, resample='up', labels=labels)
    out = ResidualBlock(reg, cfg, 'Generator.3', G, G, 3, out, resample='up', labels=labels)
    out = Normalize(reg, cfg, 'Generator.OutputN', out)
    out = torch.relu(out)
    out = ops.Conv2D(reg, 'Generator.Output', G, 3, 3, out, he_init=False)
    out = torch.tanh(out)
    return out.reshape(-1, cfg.OUTPUT_DIM)


def resnet_discriminator(reg, cfg, inputs, labels, kp1, kp2, kp3, u=None):
    """Discriminator(inputs, labels, kp1, kp2, kp3) :169-186.
    `u` = (u1,u2,u3): the three dropout uniforms, each [n,DIM_D,8,8]; may be None iff all kp == 1."""
    D = cfg.DIM_D
    u1, u2, u3 = u if u is not None else (None, None, None)
    out = inputs.reshape(-1, 3, 32, 32)
    out = OptimizedResBlockDisc1(reg, cfg, out)
    out = ResidualBlock(reg, cfg, 'Discriminator.2', D, D, 3, out, resample='down', labels=labels)
    out = tf_ops.dropout(out, kp1, u1)
    out = ResidualBlock(reg, cfg, 'Discriminator.3', D, D, 3, out, resample=None, labels=labels)
    out = tf_ops.dropout(out, kp2, u2)
    out = ResidualBlock(reg, cfg, 'Discriminator.4', D, D, 3, out, resample=None, labels=labels)
    out = tf_ops.dropout(out, kp3, u3)
    out = torch.relu(out)
    output2 = out.mean(dim=[2, 3])
    output_wgan = ops.Linear(reg, 'Discriminator.Output', D, 1, output2).reshape(-1)
    if cfg.CONDITIONAL and cfg.ACGAN:
        output_acgan = ops.Linear(reg, 'Discriminator.ACGANOutput', D, 10, output2)
        return output_wgan, output2, output_acgan
    return output_wgan, output2, None


# ----------------------------------------------------------------------------- DCGAN (CIFAR)
def cifar_generator(reg, n_samples, noise, DIM=128, OUTPUT_DIM=3072):
    """TF/CT_gan_cifar.py:58-79."""
    out = ops.Linear(reg, 'Generator.Input', 128, 4 * 4 * 4 * DIM, noise)
    out = ops.Batchnorm(reg, 'Generator.BN1', [0], out)
    out = torch.relu(out)
    out = out.reshape(-1, 4 * DIM, 4, 4)
    out = ops.Deconv2D(reg, 'Generator.2', 4 * DIM, 2 * DIM, 5, out)
    out = ops.Batchnorm(reg, 'Generator.BN2', [0, 2, 3], out)
    out = torch.relu(out)
    out = ops.Deconv2D(reg, 'Generator.3', 2 * DIM, DIM, 5, out)
    out = ops.Batchnorm(reg, 'Generator.BN3', [0, 2, 3], out)
    out = torch.relu(out)
    out = ops.Deconv2D(reg, 'Generator.5', DIM, 3, 5, out)
    out = torch.tanh(out)
    return out.reshape(-1, OUTPUT_DIM)


def cifar_discriminator(reg, inputs, u, DIM=128):
    """TF/CT_gan_cifar.py:81-100 (MODE 'wgan-CT': no BN).  u = 3 dropout uniforms (keep 0.5)."""
    out = inputs.reshape(-1, 3, 32, 32)
    out = ops.Conv2D(reg, 'Discriminator.1', 3, DIM, 5, out, stride=2)
    out = tf_ops.dropout(tf_ops.leaky_relu(out), 0.5, u[0])
    out = ops.Conv2D(reg, 'Discriminator.2', DIM, 2 * DIM, 5, out, stride=2)
    out = tf_ops.dropout(tf_ops.leaky_relu(out), 0.5, u[1])
    out = ops.Conv2D(reg, 'Discriminator.3', 2 * DIM, 4 * DIM, 5, out, stride=2)
    out = tf_ops.dropout(tf_ops.leaky_relu(out), 0.5, u[2])
    output2 = out.reshape(-1, 4 * 4 * 4 * DIM)
    out = ops.Linear(reg, 'Discriminator.Output', 4 * 4 * 4 * DIM, 1, output2)
    return out.reshape(-1), output2


# ----------------------------------------------------------------------------- DCGAN (MNIST)
def mnist_generator(reg, n_samples, noise, DIM=64, OUTPUT_DIM=784):
    """TF/CT_gan_mnist.py:62-87 (MODE 'wgan-CT': no BN)."""
    out = ops.Linear(reg, 'Generator.Input', 128, 4 * 4 * 4 * DIM, noise)
    out = torch.relu(out)
    out = out.reshape(-1, 4 * DIM, 4, 4)
    out = ops.Deconv2D(reg, 'Generator.2', 4 * DIM, 2 * DIM, 5, out)
    out = torch.relu(out)
    out = out[:, :, :7, :7]
    out = ops.Deconv2D(reg, 'Generator.3', 2 * DIM, DIM, 5, out)
    out = torch.relu(out)
    out = ops.Deconv2D(reg, 'Generator.5', DIM, 1, 5, out)
    out = torch.sigmoid(out)
    return out.reshape(-1, OUTPUT_DIM)


def mnist_discriminator(reg, inputs, u, DIM=64):
    """TF/CT_gan_mnist.py:89-108."""
    out = inputs.reshape(-1, 1, 28, 28)
    out = ops.Conv2D(reg, 'Discriminator.1', 1, DIM, 5, out, stride=2)
    out = tf_ops.dropout(tf_ops.leaky_relu(out), 0.5, u[0])
    out = ops.Conv2D(reg, 'Discriminator.2', DIM, 2 * DIM, 5, out, stride=2)
    out = tf_ops.dropout(tf_ops.leaky_relu(out), 0.5, u[1])
    out = ops.Conv2D(reg, 'Discriminator.3', 2 * DIM, 4 * DIM, 5, out, stride=2)
    out = tf_ops.dropout(tf_ops.leaky_relu(out), 0.5, u[2])
    output2 = out.reshape(-1, 4 * 4 * 4 * DIM)
    out = ops.Linear(reg, 'Discriminator.Output', 4 * 4 * 4 * DIM, 1, output2)
    return out.reshape(-1), output2


# ----------------------------------------------------------------------------- 128x128 ResNet (config[4])
class Lsun128Cfg:
    """LS/wgan_LSUN_Bedrooms128.py:27-47 (LS = tensorflow_generative_model/LSUN_bedrooms)."""

    def __init__(self, DIM_G_64=64, DIM_G_32=128, DIM_G_16=256, DIM_G_8=512, DIM_G_4=512,
                 DIM_D_64=128, DIM_D_32=256, DIM_D_16=512, DIM_D_8=1024):
        self.DIM_G_64, self.DIM_G_32, self.DIM_G_16, self.DIM_G_8, self.DIM_G_4 = DIM_G_64, DIM_G_32, DIM_G_16, DIM_G_8, DIM_G_4
        self.DIM_D_64, self.DIM_D_32, self.DIM_D_16, self.DIM_D_8 = DIM_D_64, DIM_D_32, DIM_D_16, DIM_D_8
        self.OUTPUT_DIM = 3 * 128 * 128


def _lsun_normalize(reg, name, x):
    """:70-74"""
    if 'Discriminator' in name:
        return ops.Layernorm(reg, name, [1, 2, 3], x)
    return ops.Batchnorm(reg, name, [0, 2, 3], x, fused=True)


def _lsun_scaled_upsample_conv(reg, name, input_dim, output_dim, filter_size, x, he_init=True, biases=True):
    """:87-94 (gain 0.5)"""
    return ops.Conv2D(reg, name, input_dim, output_dim, filter_size, tf_ops.upsample2(x), he_init=he_init, biases=biases, gain=0.5)


def _lsun_block(reg, name, input_dim, output_dim, filter_size, x, resample=None):
    """:96-135"""
    if output_dim == input_dim and resample is None:
        shortcut = x
    elif resample == 'down':
        shortcut = ops.Conv2D(reg, name + '.Shortcut', input_dim, output_dim, 1, tf_ops.mean_pool2(x), he_init=False, biases=True)
    elif resample == 'up':
        shortcut = _lsun_scaled_upsample_conv(reg, name + '.Shortcut', input_dim, output_dim, 1, x, he_init=False, biases=True)
    else:
        shortcut = ops.Conv2D(reg, name + '.Shortcut', input_dim, output_dim, 1, x, he_init=False, biases=True)
    out = torch.relu(_lsun_normalize(reg, name + '.N1', x))
    if resample == 'down':
        out = ops.Conv2D(reg, name + '.Conv1', input_dim, input_dim, filter_size, out)
        out = torch.relu(_lsun_normalize(reg, name + '.N2', out))
        out = ops.Conv2D(reg, name + '.Conv2', input_dim, output_dim, filter_size, out, stride=2)
    elif resample == 'up':
        out = _lsun_scaled_upsample_conv(reg, name + '.Conv1', input_dim, output_dim, filter_size, out)
        out = torch.relu(_lsun_normalize(reg, name + '.N2', out))
        out = ops.Conv2D(reg, name + '.Conv2', output_dim, output_dim, filter_size, out)
    else:
        out = ops.Conv2D(reg, name + '.Conv1', input_dim, output_dim, filter_size, out)
        out = torch.relu(_lsun_normalize(reg, name + '.N2', out))
        out = ops.Conv2D(reg, name + '.Conv2', output_dim, output_dim, filter_size, out)
    return shortcut + out


def lsun128_generator(reg, cfg, n_samples, noise):
    """ResnetGenerator :137-166 (one tower)."""
    out = ops.Linear(reg, 'Generator.Input', 128, 4 * 4 * cfg.DIM_G_4, noise)
    out = out.reshape(-1, cfg.DIM_G_4, 4, 4)
    out = _lsun_block(reg, 'Generator.4_3', cfg.DIM_G_4, cfg.DIM_G_8, 3, out, 'up')
    out = _lsun_block(reg, 'Generator.8_3', cfg.DIM_G_8, cfg.DIM_G_16, 3, out, 'up')
    out = _lsun_block(reg, 'Generator.16_3', cfg.DIM_G_16, cfg.DIM_G_32, 3, out, 'up')
    out = _lsun_block(reg, 'Generator.32_3', cfg.DIM_G_32, cfg.DIM_G_64, 3, out, 'up')
    out = torch.relu(_lsun_normalize(reg, 'Generator.OutputN', out))
    out = _lsun_scaled_upsample_conv(reg, 'Generator.Output', cfg.DIM_G_64, 3, 5, out, he_init=False)
    return torch.tanh(out).reshape(-1, cfg.OUTPUT_DIM)


def lsun128_discriminator(reg, cfg, inputs, kp1, kp2, kp3, u=None):
    """ResnetDiscriminator :168-205 -> (D [n], D_ [n, DIM_D_8])."""
    out = inputs.reshape(-1, 3, 128, 128)
    out = ops.Conv2D(reg, 'Discriminator.Input', 3, cfg.DIM_D_64, 5, out, he_init=True, stride=2)
    out = _lsun_block(reg, 'Discriminator.64_3', cfg.DIM_D_64, cfg.DIM_D_32, 3, out, 'down')
    out = _lsun_block(reg, 'Discriminator.32_3', cfg.DIM_D_32, cfg.DIM_D_16, 3, out, 'down')
    out = _lsun_block(reg, 'Discriminator.16_3', cfg.DIM_D_16, cfg.DIM_D_8, 3, out, 'down')
    if kp1 != 1.0:
        out = tf_ops.dropout(out, kp1, u[0])
    out = _lsun_block(reg, 'Discriminator.8_1', cfg.DIM_D_8, cfg.DIM_D_8, 3, out, None)
    if kp2 != 1.0:
        out = tf_ops.dropout(out, kp2, u[1])
    out = _lsun_block(reg, 'Discriminator.8_2', cfg.DIM_D_8, cfg.DIM_D_8, 3, out, None)
    if kp3 != 1.0:
        out = tf_ops.dropout(out, kp3, u[2])
    output2 = out.mean(dim=(2, 3))
    out = ops.Linear(reg, 'Discriminator.Output', cfg.DIM_D_8, 1, output2)
    return out.reshape(-1), output2


# ----------------------------------------------------------------------------- 64x64 "Good" nets
def _g64_normalize(reg, name, x):
    """TF/CT_gan_64x64.py:87-92 (MODE 'wgan-ct')."""
    if 'Discriminator' in name:
        return ops.Layernorm(reg, name, [1, 2, 3], x)
    return ops.Batchnorm(reg, name, [0, 2, 3], x, fused=True)


def _g64_block(reg, name, input_dim, output_dim, filter_size, x, resample=None):
    """:127-162"""
    if output_dim == input_dim and resample is None:
        shortcut = x
    elif resample == 'down':
        shortcut = ops.Conv2D(reg, name + '.Shortcut', input_dim, output_dim, 1, tf_ops.mean_pool2(x), he_init=False, biases=True)
    elif resample == 'up':
        shortcut = ops.Conv2D(reg, name + '.Shortcut', input_dim, output_dim, 1, tf_ops.upsample2(x), he_init=False, biases=True)
    else:
        shortcut = ops.Conv2D(reg, name + '.Shortcut', input_dim, output_dim, 1, x, he_init=False, biases=True)
    out = torch.relu(_g64_normalize(reg, name + '.BN1', x))
    if resample == 'up':
        out = ops.Conv2D(reg, name + '.Conv1', input_dim, output_dim, filter_size, tf_ops.upsample2(out), biases=False)
        out = torch.relu(_g64_normalize(reg, name + '.BN2', out))
        out = ops.Conv2D(reg, name + '.Conv2', output_dim, output_dim, filter_size, out)
    else:
        out = ops.Conv2D(reg, name + '.Conv1', input_dim, input_dim, filter_size, out, biases=False)
        out = torch.relu(_g64_normalize(reg, name + '.BN2', out))
        out = ops.Conv2D(reg, name + '.Conv2', input_dim, output_dim, filter_size, out)
        if resample == 'down':
            out = tf_ops.mean_pool2(out)
    return shortcut + out


def good_generator(reg, n_samples, noise, dim=64):
    """GoodGenerator TF/CT_gan_64x64.py:204-221."""
    out = ops.Linear(reg, 'Generator.Input', 128, 4 * 4 * 8 * dim, noise).reshape(-1, 8 * dim, 4, 4)
    out = _g64_block(reg, 'Generator.Res1', 8 * dim, 8 * dim, 3, out, 'up')
    out = _g64_block(reg, 'Generator.Res2', 8 * dim, 4 * dim, 3, out, 'up')
    out = _g64_block(reg, 'Generator.Res3', 4 * dim, 2 * dim, 3, out, 'up')
    out = _g64_block(reg, 'Generator.Res4', 2 * dim, 1 * dim, 3, out, 'up')
    out = torch.relu(_g64_normalize(reg, 'Generator.OutputN', out))
    out = ops.Conv2D(reg, 'Generator.Output', 1 * dim, 3, 3, out)
    return torch.tanh(out).reshape(-1, 64 * 64 * 3)


def good_discriminator(reg, inputs, kp1, kp2, kp3, u=None, dim=64):
    """GoodDiscriminator :357-373."""
    out = inputs.reshape(-1, 3, 64, 64)
    out = ops.Conv2D(reg, 'Discriminator.Input', 3, dim, 3, out, he_init=False)
    out = _g64_block(reg, 'Discriminator.Res1', dim, 2 * dim, 3, out, 'down')
    out = _g64_block(reg, 'Discriminator.Res2', 2 * dim, 4 * dim, 3, out, 'down')
    if kp1 != 1.0:
        out = tf_ops.dropout(out, kp1, u[0])
    out = _g64_block(reg, 'Discriminator.Res3', 4 * dim, 8 * dim, 3, out, 'down')
    if kp2 != 1.0:
        out = tf_ops.dropout(out, kp2, u[1])
    out = _g64_block(reg, 'Discriminator.Res4', 8 * dim, 8 * dim, 3, out, 'down')
    if kp3 != 1.0:
        out = tf_ops.dropout(out, kp3, u[2])
    output2 = out.reshape(-1, 4 * 4 * 8 * dim)
    out = ops.Linear(reg, 'Discriminator.Output', 4 * 4 * 8 * dim, 1, output2)
    return out.reshape(-1), output2
