"""Loss graphs, optimizer and step ordering of the CT-WGAN scripts, restated AS WRITTEN (oracle).

Test infrastructure only.  "As written" = the live nodes of the reference graph: 2 G towers, three
critic passes over the concatenated real+fake batch (two dropout passes + the clean accuracy
pass), the gradient-penalty critic call with its own masks, the consistency term on the real
half.  Every tf.random_* draw is an explicit entry of the `rnd` dict.
  resnet: TF/CT_gan_cifar_resnet.py:190-338 (losses/optimizers), :393-404 (ordering)
  dcgan : TF/CT_gan_cifar.py:102-154, TF/CT_gan_mnist.py:110-179
"""
import torch

from . import nets, tf_ops


# --------------------------------------------------------------------------- random inputs
def make_rnd_resnet_d(B, DIM_D, gen, dtype=torch.float64):
    """All draws of one ResNet D step (shapes of TF/CT_gan_cifar_resnet.py:157,202,226-227,277,284)."""
    def U(*s):
        return torch.rand(*s, generator=gen, dtype=torch.float32).to(dtype)
    h = B // 2
    return {
        'z': [torch.randn(h, 128, generator=gen, dtype=torch.float32).to(dtype) for _ in range(2)],
        'dequant': U(B, 3072) / 128.,
        'alpha': U(B, 1),
        'u_pass1': [U(2 * B, DIM_D, 8, 8) for _ in range(3)],
        'u_pass2': [U(2 * B, DIM_D, 8, 8) for _ in range(3)],
        'u_gp': [U(B, DIM_D, 8, 8) for _ in range(3)],
    }


def make_rnd_resnet_g(B, DIM_D, gen, dtype=torch.float64, mult=2):
    """Draws of one ResNet G step: per tower z, label uniforms, dropout uniforms (:316-321)."""
    n = mult * B // 2
    def U(*s):
        return torch.rand(*s, generator=gen, dtype=torch.float32).to(dtype)
    return {
        'z': [torch.randn(n, 128, generator=gen, dtype=torch.float32).to(dtype) for _ in range(2)],
        'label_u': [U(n) for _ in range(2)],
        'u': [[U(n, DIM_D, 8, 8) for _ in range(3)] for _ in range(2)],
    }


# --------------------------------------------------------------------------- ResNet losses
def resnet_real_prep(real_int, dequant):
    """:201-202  2*((int/256.)-.5) + U[0,1/128)"""
    return 2. * ((real_int.to(dequant.dtype) / 256.) - .5) + dequant


def ct_term(d, d_, f, f_, LAMBDA_2=2.0, Factor_M=0.0):
    """:288-291 consistency term on the real half."""
    CT = LAMBDA_2 * (d - d_) ** 2
    CT = CT + LAMBDA_2 * 0.1 * ((f - f_) ** 2).mean(dim=1)
    CT_ = torch.maximum(CT - Factor_M, 0.0 * (CT - Factor_M))
    return CT_.mean()


def resnet_d_losses(reg, cfg, real_int, labels, rnd, B=64, LAMBDA_2=2.0, Factor_M=0.0,
                    ACGAN_SCALE=1.0, create_graph=True):
    """The D-step loss graph, single-GPU placement (DEVICES=[d,d]) :194-305.  Returns dict."""
    h = B // 2
    lab = [labels[:h], labels[h:]]
    with torch.no_grad():    # generator output is a constant w.r.t. the critic parameters
        fake = torch.cat([nets.resnet_generator(reg, cfg, h, lab[i], rnd['z'][i]) for i in range(2)], 0)
    real = resnet_real_prep(real_int, rnd['dequant'])
    rf = torch.cat([real, fake], 0)
    rf_labels = torch.cat([labels, labels], 0)
    d1, f1, a1 = nets.resnet_discriminator(reg, cfg, rf, rf_labels, 0.8, 0.5, 0.5, rnd['u_pass1'])
    d2, f2, a2 = nets.resnet_discriminator(reg, cfg, rf, rf_labels, 0.8, 0.5, 0.5, rnd['u_pass2'])
    with torch.no_grad():
        dc, fc, ac = nets.resnet_discriminator(reg, cfg, rf, rf_labels, 1.0, 1.0, 1.0, None)
    wgan = d1[B:].mean() - d1[:B].mean()
    out = {}
    if cfg.CONDITIONAL and cfg.ACGAN:
        acgan = tf_ops.sparse_softmax_ce(a1[:B], labels).mean()
        out['acc_real'] = (ac[:B].argmax(1) == labels).to(real.dtype).mean()
        out['acc_fake'] = (ac[B:].argmax(1) == labels).to(real.dtype).mean()
    else:
        acgan = torch.zeros((), dtype=real.dtype)
        out['acc_real'] = out['acc_fake'] = torch.zeros((), dtype=real.dtype)
    # gradient penalty :277-286
    interp = (real + rnd['alpha'] * (fake - real)).detach().requires_grad_(True)
    dgp = nets.resnet_discriminator(reg, cfg, interp, labels, 0.8, 0.5, 0.5, rnd['u_gp'])[0]
    grads = torch.autograd.grad(dgp.sum(), interp, create_graph=create_graph)[0]
    slopes = torch.sqrt((grads ** 2).sum(dim=1))
    gp = 10.0 * ((slopes - 1.) ** 2).mean()
    ct = ct_term(d1[:B], d2[:B], f1[:B], f2[:B], LAMBDA_2, Factor_M)
    disc_wgan = wgan + ct + gp
    out.update(cost=disc_wgan + ACGAN_SCALE * acgan, wgan=disc_wgan, acgan=acgan,
               wgan_only=wgan, ct=ct, gp=gp, slopes=slopes, fake=fake, real=real,
               d_real=d1[:B], d_fake=d1[B:], gp_grads=grads)
    return out


def resnet_g_losses(reg, cfg, rnd, B=64, GEN_BS_MULTIPLE=2, ACGAN_SCALE_G=0.1):
    """The G-step loss graph :314-330 (two towers of GEN_BS_MULTIPLE*B/2 samples each)."""
    n = GEN_BS_MULTIPLE * B // 2
    costs, acg, samples = [], [], []
    for t in range(2):
        fake_labels = (rnd['label_u'][t] * 10).to(torch.int32)        # tf.cast truncates
        x = nets.resnet_generator(reg, cfg, n, fake_labels, rnd['z'][t])
        d, _, a = nets.resnet_discriminator(reg, cfg, x, fake_labels, 0.8, 0.5, 0.5, rnd['u'][t])
        costs.append(-d.mean())
        if cfg.CONDITIONAL and cfg.ACGAN:
            acg.append(tf_ops.sparse_softmax_ce(a, fake_labels).mean())
        samples.append(x)
    gen_cost = sum(costs) / 2
    if acg:
        gen_cost = gen_cost + ACGAN_SCALE_G * (sum(acg) / 2)
    return {'cost': gen_cost, 'samples': samples}


# --------------------------------------------------------------------------- optimizer
class TFAdam:
    """tf.train.AdamOptimizer slots for a named parameter list (one instance per optimizer, as at
    :333-334).  `names` fixes the variable list (params_with_name minus non-trainables)."""

    def __init__(self, reg, names, beta1, beta2, eps=1e-8):
        self.reg, self.names, self.b1, self.b2, self.eps = reg, list(names), beta1, beta2, eps
        self.t = 0
        self.m = {n: torch.zeros_like(reg[n]) for n in self.names}
        self.v = {n: torch.zeros_like(reg[n]) for n in self.names}

    def apply(self, grads, lr):
        self.t += 1
        with torch.no_grad():
            for n in self.names:
                g = grads.get(n)
                if g is None:
                    continue
                th, self.m[n], self.v[n] = tf_ops.tf_adam_step(
                    self.reg[n], g, self.m[n], self.v[n], self.t, lr, self.b1, self.b2, self.eps)
                self.reg[n].copy_(th)


def lr_decay(iteration, ITERS):
    """:309-310  max(0, 1 - it/ITERS)"""
    return max(0., 1. - float(iteration) / ITERS)


def grads_of(cost, reg, sub):
    named = reg.trainable_with_name(sub)
    gs = torch.autograd.grad(cost, [p for _, p in named], allow_unused=True)
    return {n: g for (n, _), g in zip(named, gs) if g is not None}


def resnet_d_step(reg, cfg, opt, real_int, labels, rnd, iteration, ITERS=100000, LR=2e-4, B=64, **kw):
    """One session.run([... disc_train_op]) :402.  Returns the loss dict (+ 'grads')."""
    out = resnet_d_losses(reg, cfg, real_int, labels, rnd, B=B, **kw)
    g = grads_of(out['cost'], reg, 'Discriminator.')
    opt.apply(g, LR * lr_decay(iteration, ITERS))
    out['grads'] = g
    return out


def resnet_g_step(reg, cfg, opt, rnd, iteration, ITERS=100000, LR=2e-4, B=64, **kw):
    """One session.run([gen_train_op]) :397."""
    out = resnet_g_losses(reg, cfg, rnd, B=B, **kw)
    g = grads_of(out['cost'], reg, 'Generator')
    opt.apply(g, LR * lr_decay(iteration, ITERS))
    out['grads'] = g
    return out


# --------------------------------------------------------------------------- DCGAN scripts
def make_rnd_dcgan_d(B, feat_shapes, gen, dtype=torch.float64):
    """feat_shapes: the three post-LeakyReLU activation shapes [(C,H,W)]*3 of the critic."""
    def U(*s):
        return torch.rand(*s, generator=gen, dtype=torch.float32).to(dtype)
    def masks():
        return [U(B, *s) for s in feat_shapes]
    return {'z': torch.randn(B, 128, generator=gen, dtype=torch.float32).to(dtype),
            'alpha': U(B, 1), 'u_real': masks(), 'u_real_': masks(), 'u_fake': masks(), 'u_gp': masks()}


def make_rnd_dcgan_g(B, feat_shapes, gen, dtype=torch.float64):
    def U(*s):
        return torch.rand(*s, generator=gen, dtype=torch.float32).to(dtype)
    return {'z': torch.randn(B, 128, generator=gen, dtype=torch.float32).to(dtype),
            'u_fake': [U(B, *s) for s in feat_shapes]}


def dcgan_d_losses(reg, G, D, real, rnd, LAMBDA=10., LAMBDA_2=2.0, Factor_M=0.0):
    """MODE 'wgan-CT' critic loss: TF/CT_gan_cifar.py:107-151 == TF/CT_gan_mnist.py:114-167.
    `real` is already float (cifar: 2*(int/255-.5), TF/CT_gan_cifar.py:103; mnist: fed in [0,1])."""
    B = real.shape[0]
    with torch.no_grad():
        fake = G(reg, B, rnd['z'])
    d_real, f_real = D(reg, real, rnd['u_real'])
    d_real_, f_real_ = D(reg, real, rnd['u_real_'])
    d_fake, _ = D(reg, fake, rnd['u_fake'])
    wgan = d_fake.mean() - d_real.mean()
    ct = ct_term(d_real, d_real_, f_real, f_real_, LAMBDA_2, Factor_M)
    interp = (real + rnd['alpha'] * (fake - real)).detach().requires_grad_(True)
    dgp = D(reg, interp, rnd['u_gp'])[0]
    grads = torch.autograd.grad(dgp.sum(), interp, create_graph=True)[0]
    slopes = torch.sqrt((grads ** 2).sum(dim=1))
    gp = ((slopes - 1.) ** 2).mean()
    return {'cost': wgan + ct + LAMBDA * gp, 'wgan_only': wgan, 'ct': ct, 'gp': gp, 'fake': fake,
            'slopes': slopes, 'gp_grads': grads}


def dcgan_g_losses(reg, G, D, B, rnd):
    """gen_cost = -mean(D(G(z)))  (TF/CT_gan_cifar.py:124, TF/CT_gan_mnist.py:147)."""
    x = G(reg, B, rnd['z'])
    d, _ = D(reg, x, rnd['u_fake'])
    return {'cost': -d.mean(), 'samples': x}
