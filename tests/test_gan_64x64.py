"""64x64 GoodGenerator / GoodDiscriminator (TF/CT_gan_64x64.py:166-221,357-373) on the shared
CT-WGAN step against their oracle restatement at reduced width: forward, critic step (GP through Layernorm), generator step."""
import pytest
import torch

from oracle import nets as onets, steps as osteps, tflib_ref as oref
from tests.test_lsun128 import _cmp, _oracle_from_product, _to


def _run(lib, dev, dim, B, tol):
    import ctgan_amd.gan_64x64 as M
    from ctgan_amd.dcgan_step import DCGANTrainer
    M.configure(BATCH_SIZE=B, DIM=dim)
    try:
        lib.set_seed(4)
        M.build_params(dev)
        assert 'Discriminator.Res1.Conv1.Biases' not in lib._params and 'Discriminator.Res1.Conv2.Biases' in lib._params
        reg = _oracle_from_product(lib)
        g = torch.Generator().manual_seed(8)
        G = lambda r, n, zz: onets.good_generator(r, n, zz, dim=dim)                              # noqa: E731
        D = lambda r, xx, uu: onets.good_discriminator(r, xx, 0.8, 0.5, 0.5, uu, dim=dim)         # noqa: E731
        z = torch.randn(B, 128, generator=g)
        x = M.Generator(B, noise=z.to(dev)); xo = G(reg, B, z.double())
        _cmp(x, xo, tol, 'generator')
        tr = DCGANTrainer(M, seed=1)
        real_in = torch.randint(0, 256, (B, M.cfg.OUTPUT_DIM), generator=g, dtype=torch.int32)
        real_o = 2 * ((real_in.double() / 255.) - .5)
        rnd = osteps.make_rnd_dcgan_d(B, M.feat_shapes(), g)
        out = tr.d_step(real_in.to(dev), {k: _to(v, dev) for k, v in rnd.items()})
        ref = osteps.dcgan_d_losses(reg, G, D, real_o, rnd)
        gref = osteps.grads_of(ref['cost'], reg, 'Discriminator')
        for k in ('cost', 'wgan_only', 'ct'):
            _cmp(out[k], ref[k], 10 * tol, 'd.' + k, atol=1e-6)
        _cmp(out['gp'], M.cfg.LAMBDA * ref['gp'], 10 * tol, 'd.gp', atol=1e-6)
        assert set(n for n, v in out['grads'].items() if v is not None) == set(gref)
        for n in gref:
            a = out['grads'][n].detach().cpu().double().reshape(-1); b = gref[n].detach().double().reshape(-1)
            assert (a - b).norm().item() <= 100 * tol * b.norm().item() + 3e-6, 'dgrad ' + n
        lib.load_state_dict({n: t.detach().float() for n, t in reg.items()})
        rg = osteps.make_rnd_dcgan_g(B, M.feat_shapes(), g)
        out = tr.g_step({k: _to(v, dev) for k, v in rg.items()})
        ref = osteps.dcgan_g_losses(reg, G, D, B, rg)
        _cmp(out['cost'], ref['cost'], 10 * tol, 'g cost', atol=1e-6)
        gref = osteps.grads_of(ref['cost'], reg, 'Generator')
        for n in gref:
            a = out['grads'][n].detach().cpu().double().reshape(-1); b = gref[n].detach().double().reshape(-1)
            assert (a - b).norm().item() <= 400 * tol * b.norm().item() + 3e-6, 'ggrad ' + n
    finally:
        M.configure()


def test_64x64_nets_and_steps_match_oracle(cpu_kernels):
    import ctgan_amd.tflib as lib
    _run(lib, 'cpu', 4, 3, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [32, 64])
def test_64x64_on_gpu(dim):
    """GoodGenerator / GoodDiscriminator (TF/CT_gan_64x64.py:166-221,357-373) against the fp64 oracle at half width and at the
    reference width DIM = 64 (critic 64..512 channels), B = 4."""
    import ctgan_amd.tflib as lib
    lib.delete_all_params(); lib.set_device(None)
    try:
        _run(lib, 'cuda', dim, 4, 5e-5)
    finally:
        lib.delete_all_params()
