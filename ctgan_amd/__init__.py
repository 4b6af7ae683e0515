"""ctgan_amd - MI355X-native CT-WGAN adversarial-step hot path (drop-in for biuyq/CT-GAN's
tflib operator API and the Generator/Discriminator surfaces of CT_gan_{mnist,cifar,cifar_resnet}.py).

Layout
  csrc/                  hand-written gfx950 HIP kernels + the C-ABI (include/ctgan_hip.h)
  _lib.py, kernels.py    ctypes binding and tensor-level wrappers (no CPU fallback)
  functional.py          autograd wiring (double-backward capable) over the kernels
  tflib/                 the reference's operator library API: lib.param registry + tflib.ops.*
  gan_cifar_resnet.py    ResNet CT-WGAN (Generator/Discriminator, D/G step, train loop)
  gan_cifar.py, gan_mnist.py   the DCGAN scripts
  ddp.py                 batch-sharded step over RCCL (flat gradient buckets)
"""
__version__ = '0.1.0'
