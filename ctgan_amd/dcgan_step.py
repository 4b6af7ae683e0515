"""Shared step of the two DCGAN scripts (MODE 'wgan-CT'): TF/CT_gan_cifar.py:102-154,190-204 and
TF/CT_gan_mnist.py:110-179,232-249.  Loss = WGAN + consistency term (two dropout passes over the
real batch) + LAMBDA * gradient penalty; Adam(1e-4, beta1=.5, beta2=.9), no LR decay.

Exact restructurings: the three live critic calls of the reference (real with masks A, real with masks
B, fake with masks C) are evaluated as ONE batch of 3B rows (dropout is elementwise, the critic has no
batch-coupled op); where the module exposes `DiscriminatorTrunk` / `DiscriminatorTail` (the layer-normalised
ResNet critics, whose first dropout sits after the 16x16 blocks) the deterministic trunk runs once on [real ; fake]
and the two dropout passes over the real batch share it; the WGAN difference, the consistency term and the sum with the
gradient penalty are one fused loss-heads launch each way (rows ordered real, fake, real); the dead 4th call `disc_fake_2` and the extra
generators are not executed.  Without injected draws (`rnd=None`) the dropout masks are regenerated from the
Philox streams inside the kernels.
"""
import torch

from . import functional as F
from . import kernels as K
from . import tflib as lib
from .optim import FlatAdam
from .rng import DeviceRNG


import os as _os
# A/B switch: the two dropout passes over the real batch share the critic's layers before the first dropout (modules that expose
# DiscriminatorTrunk / DiscriminatorTail: the layer-normalised ResNet critics, whose dropouts sit after the 16x16 blocks)
TRUNK_SHARE = _os.environ.get('CTGAN_UNCOND_TRUNK_SHARE', '1') != '0'


# The fake batches of an iteration's critic steps from one generator forward (A/B switch; the ResNet trainer's BATCH_FAKES)
BATCH_FAKES = _os.environ.get('CTGAN_DCGAN_BATCH_FAKES', '1') != '0'


class DCGANTrainer:
    def __init__(self, module, seed=2024, rank=0, world_size=1, allreduce=None):
        """`module` = ctgan_amd.gan_cifar or ctgan_amd.gan_mnist (provides cfg, Generator, Discriminator,
        real_prep, feat_shapes)."""
        self.mod = module
        self.dev = lib._dev()
        self.rank, self.world, self.allreduce = rank, world_size, allreduce
        self.rng = DeviceRNG(seed, rank, self.dev)
        self.d_named = lib.named_params_with_name('Discriminator', trainable_only=True)
        self.g_named = lib.named_params_with_name('Generator', trainable_only=True)
        b1, b2 = getattr(module, 'ADAM_BETAS', (0.5, 0.9))
        self.d_opt = FlatAdam(self.d_named, b1, b2)
        self.g_opt = FlatAdam(self.g_named, b1, b2)
        self.towers = getattr(module, 'GEN_TOWERS', 1)                    # generator calls per batch, each with its own BN statistics
        self.piecewise = getattr(module, 'PIECEWISE_LINEAR_CRITIC', True)
        self.iteration = 0
        self.d_params = [p for _, p in self.d_named]
        self.g_params = [p for _, p in self.g_named]
        # Power-of-two loss scale for the fp16 matrix-core mode (kernels.set_mma_dtype('f16')): the backward passes are linear in the
        # seed, so the cost gradient is seeded with S instead of 1 - every first-order gradient TENSOR the fp16 kernels round is S
        # times larger, away from fp16's subnormals (per-pixel gradients shrink with 1 / (B H W)) - and Adam divides it out again
        # (grad_scale); exact in fp32 (a power of two), a no-op for S = 1.
        self.loss_scale = float(getattr(module, 'LOSS_SCALE', 1.0))
        self._seed = None

    def cost_seed(self):
        """grad_outputs of the final backward: a cached 0-dim tensor holding the loss scale (no fill kernel per step)."""
        if self._seed is None or float(self._seed_val) != self.loss_scale:
            self._seed = torch.full((), self.loss_scale, dtype=torch.float32, device=self.dev)
            self._seed_val = self.loss_scale
        return self._seed

    def d_losses(self, real_in, rnd=None, fake=None):
        m, cfg = self.mod, self.mod.cfg
        B = cfg.BATCH_SIZE
        with torch.no_grad():
            if fake is None:        # `fake`: a batch drawn earlier from the same generator weights (generate_fakes)
                fake = self._gen(B, rnd['z'] if rnd is not None else None)
            real = m.real_prep(real_in)
            alpha = rnd['alpha'] if rnd is not None else self.rng.uniform(B, 1)
            interp = K.interpolate(real, fake, alpha)
        # rows of the batched passes: real (masks A), fake (masks C), real (masks B) - the order the fused loss heads read
        u = [torch.cat([a, c, b], 0) for a, b, c in zip(rnd['u_real'], rnd['u_real_'], rnd['u_fake'])] if rnd is not None else None
        if TRUNK_SHARE and hasattr(m, 'DiscriminatorTrunk') and getattr(m, 'critic_is_per_sample', lambda: True)():
            # the critic's layers before its first dropout are deterministic and per-sample: the two dropout passes over the real
            # batch share ONE evaluation of them - rows [real ; fake] through the trunk, rows [real, fake, real] through the tail
            h = m.DiscriminatorTrunk(torch.cat([real, fake], 0))
            h3 = F.rows_select(h, [(0, B), (B, 2 * B), (0, B)])
            d, f = m.DiscriminatorTail(h3, u=u, rng=None if rnd is not None else self.rng)
        else:
            x3 = torch.cat([real, fake, real], 0)
            # masks regenerated from the Philox stream inside the dropout kernels when none are injected: no uniform tensors
            d, f = m.Discriminator(x3, u=u) if rnd is not None else m.Discriminator(x3, rng=self.rng)
        interp.requires_grad_(True)
        with F.weight_grads(not self.piecewise):     # a LeakyReLU + dropout critic is piecewise linear; a layer-normalised one is not
            d_gp = (m.Discriminator(interp, u=rnd['u_gp']) if rnd is not None else m.Discriminator(interp, rng=self.rng))[0]
        (grads,) = torch.autograd.grad(d_gp, interp, grad_outputs=torch.ones_like(d_gp), create_graph=True)
        gp, slopes = F.gradient_penalty(grads, cfg.LAMBDA)
        # mean(fake) - mean(real), the consistency term over the two real passes and the sum with gp: one launch each way
        cost, wgan, ct, _, _ = F.critic_heads(d, f, None, None, B, cfg.LAMBDA_2, cfg.Factor_M, 0.0, gp)
        return {'cost': cost, 'wgan_only': wgan, 'ct': ct, 'gp': gp, 'fake': fake, 'slopes': slopes, 'gp_grads': grads}

    def d_grads(self, real_in, rnd=None, fake=None):
        """Losses and parameter gradients of one critic step -> (out, grads aligned with self.d_params; scaled by the loss scale).  The
        DCGAN scripts' piecewise-linear critic runs the hand-scheduled step (dcgan_schedule.py: one forward and one backward chain over
        [real, fake, real | x_hat], the penalty's double backward on its own rows); everything else - injected draws, the layer-normalised
        ResNet critics - the autograd form."""
        from . import dcgan_schedule as DS
        if fake is None and rnd is None:
            with torch.no_grad():
                fake = self._gen(self.mod.cfg.BATCH_SIZE, None)
        if DS.usable(self, rnd, fake, real_in):
            with torch.no_grad(), F.deferred_wgrads():
                return DS.critic_step(self, real_in, fake)
        out = self.d_losses(real_in, rnd, fake=fake)
        with F.deferred_wgrads():       # the queued weight gradients of the step: one grouped launch (functional._flush_groups)
            grads = torch.autograd.grad(out['cost'], self.d_params, grad_outputs=self.cost_seed().reshape(out['cost'].shape), allow_unused=True)
        return out, grads

    def _gen(self, n, z, groups=1):
        g = self.towers * groups
        if g > 1:
            return self.mod.Generator(n, noise=z, rng=self.rng, groups=g)
        return self.mod.Generator(n, noise=z, rng=self.rng)

    def generate_fakes(self, n_steps):
        """The fake batches of the next `n_steps` critic steps in ONE generator forward: the generator does not change between the critic
        updates of an iteration (TF/CT_gan_cifar.py:190-204), each step's batch keeps its own BatchNorm statistic group(s).  A step of its
        own in the Philox numbering (as gan_cifar_resnet.Trainer.generate_fakes)."""
        B = self.mod.cfg.BATCH_SIZE
        self.rng.begin_step()
        with torch.no_grad():
            fake = self._gen(n_steps * B, None, groups=n_steps)
        self.rng.end_step()
        return fake.reshape(n_steps, B, -1)

    def g_losses(self, rnd=None):
        m, B = self.mod, self.mod.cfg.BATCH_SIZE
        x = self._gen(B, rnd['z'] if rnd is not None else None)
        with F.weight_grads(False):
            d, _ = m.Discriminator(x, u=rnd['u_fake']) if rnd is not None else m.Discriminator(x, rng=self.rng)
        return {'cost': F.mean_diff(d, B, 0, -1.0, 0.0), 'samples': x}

    def _apply(self, opt, grads):
        opt.set_lr(self.mod.lr(self.iteration) if hasattr(self.mod, 'lr') else self.mod.cfg.LR)
        scale = 1.0 / (self.world * self.loss_scale)
        if self.allreduce is None or self.world <= 1:
            opt.update(grads, scale, rng=self.rng)          # bucket + Adam: one launch; end of the step (beta powers, Philox counter): one launch
            return
        flat = opt.gather_grads(grads)
        self.allreduce(flat)
        if hasattr(self.allreduce, 'wait'):
            self.allreduce.wait()
        opt.step(grad_scale=scale, rng=self.rng)

    def _unscaled(self, grads):
        if self.loss_scale == 1.0:
            return grads
        return [None if g is None else g / self.loss_scale for g in grads]

    def d_step(self, real_in, rnd=None, fake=None):
        self.rng.begin_step()
        out, grads = self.d_grads(real_in, rnd, fake=fake)
        self._apply(self.d_opt, grads)
        out['grads'] = dict(zip([n for n, _ in self.d_named], self._unscaled(grads)))
        return out

    def g_step(self, rnd=None):
        self.rng.begin_step()
        out = self.g_losses(rnd)
        with F.deferred_wgrads():
            grads = torch.autograd.grad(out['cost'], self.g_params, grad_outputs=self.cost_seed().reshape(out['cost'].shape), allow_unused=True)
        self._apply(self.g_opt, grads)
        out['grads'] = dict(zip([n for n, _ in self.g_named], self._unscaled(grads)))
        return out

    def train_iteration(self, iteration, next_batch):
        self.iteration = iteration
        if iteration > 0:
            self.g_step()
        out = None
        n = self.mod.cfg.CRITIC_ITERS
        fakes = self.generate_fakes(n) if BATCH_FAKES else None
        for i in range(n):
            out = self.d_step(next_batch(), fake=None if fakes is None else fakes[i])
        return out
