"""Counter-based device random streams (Philox4x32-10 kernels in csrc/optim_rng.hip).

Replaces the reference's tf.random_normal / tf.random_uniform / tf.nn.dropout draws
(TF/CT_gan_cifar_resnet.py:157,202,277,319,173-177).  A stream is addressed by
(seed, rank, call-site index, step counter); the step counter lives in device memory and is
advanced by a kernel, so a captured hipGraph draws fresh numbers on every replay.
"""
import torch

from . import kernels as K


class DeviceRNG:
    def __init__(self, seed=2024, rank=0, device=None):
        self.seed = int(seed)
        self.rank = int(rank)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.ctr = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._site = 0

    def _sid(self):
        s = self._site
        self._site += 1
        if s >= (1 << 16):
            raise RuntimeError('too many random call sites in one step')
        return (self.rank << 16) | s

    def begin_step(self):
        """Call-site numbering restarts with every step (keeps captured and eager runs aligned)."""
        self._site = 0

    def end_step(self):
        K.rng_advance(self.ctr, 1)

    def uniform(self, *shape, lo=0.0, hi=1.0, channels_last=False):
        if channels_last and len(shape) == 4:
            out = K.empty_cl(*shape, device=self.device)
        else:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        return K.rng_uniform(out, self.seed, self._sid(), self.ctr, lo, hi)

    def normal(self, *shape):
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        return K.rng_normal(out, self.seed, self._sid(), self.ctr)

    def labels(self, n, nlab=10):
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        return K.rng_labels(out, nlab, self.seed, self._sid(), self.ctr)
