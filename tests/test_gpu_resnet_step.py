"""End-to-end parity of the ResNet CT-WGAN step on the MI355X against the CPU oracle: identical
weights and identical injected randomness (z, dequant noise, alpha, every dropout uniform).
Tolerances are fp32 round-off bounds vs the fp64 oracle, stated per quantity."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets as onets, steps as osteps, tflib_ref as oref  # noqa: E402


def _oracle_from_product(lib, dtype=torch.float64):
    reg = oref.Registry(dtype=dtype)
    for n, p in lib._params.items():
        t = p.detach().cpu().clone().to(dtype)
        trainable = n not in lib._non_trainable
        t.requires_grad_(trainable)
        reg[n] = t
        if not trainable:
            reg.non_trainable.add(n)
    return reg


def _cmp(a, b, tol, what, atol=1e-7):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    assert err <= tol * scale + atol, '%s: max err %.3e vs scale %.3e' % (what, err, scale)


def _to_dev(o):
    if isinstance(o, list):
        return [_to_dev(t) for t in o]
    return o.float().cuda()


@pytest.fixture
def setup():
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    lib.delete_all_params(); lib.set_device(None)

    def make(dim, B, seed=5):
        lib.delete_all_params(); lib.set_seed(seed)
        R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
        R.build_params()
        return R, lib
    yield make
    lib.delete_all_params(); R.configure()


@pytest.mark.parametrize('dim,B,steps', [(16, 8, 3), (128, 64, 1)])
def test_d_and_g_step_parity(setup, dim, B, steps):
    R, lib = setup(dim, B)
    reg = _oracle_from_product(lib)
    cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    g = torch.Generator().manual_seed(11)
    tr = R.Trainer(seed=1)
    optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
    for it in range(steps):
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd = osteps.make_rnd_resnet_d(B, dim, g)
        out = tr.d_step(real.cuda(), labels.cuda(), {k: _to_dev(v) for k, v in rnd.items()}, iteration=it)
        ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=it, B=B)
        # north-star tolerance: losses within 1e-3 relative; measured fp32-vs-fp64 error is ~1e-5
        for k in ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only'):
            _cmp(out[k], ref[k], 2e-4, 'd_step[%d].%s' % (it, k), atol=1e-6)
        _cmp(out['acc_real'], ref['acc_real'], 0, 'acc_real', atol=1.01 / B)     # argmax ties only
        _cmp(out['fake'], ref['fake'], 1e-4, 'generator samples')
        _cmp(out['gp_grads'], ref['gp_grads'], 5e-4, 'dD/dx_hat')
        for n in ref['grads']:
            _cmp(out['grads'][n], ref['grads'][n], 2e-3, 'dgrad ' + n, atol=1e-6)
        for n, _ in reg.trainable_with_name('Discriminator.'):
            _cmp(lib._params[n], reg[n], 1e-3, 'theta ' + n, atol=3e-5)
        rg = osteps.make_rnd_resnet_g(B, dim, g)
        out = tr.g_step({'z': _to_dev(rg['z']), 'label_u': _to_dev(rg['label_u']), 'u': _to_dev(rg['u'])},
                        iteration=it + 1)
        ref = osteps.resnet_g_step(reg, cfg, optG, rg, iteration=it + 1, B=B)
        _cmp(out['cost'], ref['cost'], 2e-4, 'g cost', atol=1e-6)
        for n in ref['grads']:
            _cmp(out['grads'][n], ref['grads'][n], 5e-3, 'ggrad ' + n, atol=1e-6)


def test_step_is_deterministic_and_rng_advances(setup):
    R, lib = setup(16, 8)
    g = torch.Generator().manual_seed(3)
    real = torch.randint(0, 256, (8, 3072), generator=g, dtype=torch.int32).cuda()
    labels = torch.randint(0, 10, (8,), generator=g, dtype=torch.int32).cuda()
    sd = lib.state_dict()
    costs = []
    for _ in range(2):
        lib.load_state_dict(sd)
        tr = R.Trainer(seed=42)
        o1 = tr.d_step(real, labels, iteration=0)
        o2 = tr.d_step(real, labels, iteration=0)
        costs.append((o1['cost'].item(), o2['cost'].item(), tr.d_opt.theta.clone()))
    assert costs[0][0] == costs[1][0] and costs[0][1] == costs[1][1]        # same seed -> same bits
    assert torch.equal(costs[0][2], costs[1][2])
    assert costs[0][0] != costs[0][1]                                       # fresh draws every step


def test_train_iteration_and_samples(setup):
    R, lib = setup(16, 8)
    tr = R.Trainer(seed=7)
    g = torch.Generator().manual_seed(4)
    batches = [(torch.randint(0, 256, (8, 3072), generator=g, dtype=torch.int32).cuda(),
                torch.randint(0, 10, (8,), generator=g, dtype=torch.int32).cuda()) for _ in range(3)]
    i = [0]

    def nxt():
        i[0] += 1
        return batches[i[0] % 3]
    for it in range(3):
        out = tr.train_iteration(it, nxt)
        assert torch.isfinite(out['cost'])
    assert tr.d_opt.t == 15 and tr.g_opt.t == 2
    s, px = tr.generate_samples(torch.randn(100, 128, device='cuda'),
                                torch.arange(10, dtype=torch.int32, device='cuda').repeat(10))
    assert s.shape == (100, 3072) and px.min() >= 0 and px.max() <= 255
