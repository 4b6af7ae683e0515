"""Checkpoint / resume of a training run.

The reference has no resume for its CIFAR/MNIST scripts (`np.save("param.pyn", ...)` of the critic only,
TF/CT_gan_cifar.py:216-222; a `tf.train.Saver` in the LSUN script).  Here the registry's names make it
trivial: one file holds every parameter by its reference name and layout, both Adam slot sets (m, v,
beta-power state, step count), the Philox step counter and the loop iteration - enough for a bit-exact
continuation (tests/test_gpu_checkpoint.py)."""
import torch

from . import tflib as lib


def save(path, trainer, iteration, extra=None):
    torch.save({
        'format': 1,
        'iteration': int(iteration),
        'params': lib.state_dict(),
        'd_opt': trainer.d_opt.state_dict(),
        'g_opt': trainer.g_opt.state_dict(),
        'rng': {'seed': trainer.rng.seed, 'rank': trainer.rng.rank, 'ctr': int(trainer.rng.ctr.item())},
        'extra': extra or {},
    }, path)


def load(path, trainer):
    """Restores weights, optimizer slots and random-stream position into `trainer`; returns the iteration
    to continue from."""
    ck = torch.load(path, map_location='cpu', weights_only=False)
    if ck.get('format') != 1:
        raise ValueError('unknown checkpoint format')
    lib.load_state_dict(ck['params'], strict=True)
    trainer.d_opt.load_state_dict(ck['d_opt'])
    trainer.g_opt.load_state_dict(ck['g_opt'])
    trainer.rng.seed = ck['rng']['seed']
    trainer.rng.ctr.fill_(ck['rng']['ctr'])
    return ck['iteration']
