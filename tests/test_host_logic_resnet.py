"""Host logic of the ResNet CT-WGAN step (autograd wiring incl. the GP double backward, registry,
restructured loss graph, flat TF-Adam) checked against the CPU oracle - with the HIP kernel
wrappers swapped for torch-CPU stand-ins (fixture cpu_kernels).  The kernels themselves are
checked on the GPU box by tests/test_gpu_*.py."""
import numpy as np
import pytest
import torch

from oracle import nets as onets, steps as osteps, tflib_ref as oref


def _oracle_from_product(lib, dtype=torch.float64):
    reg = oref.Registry(dtype=dtype)
    for n, p in lib._params.items():
        t = p.detach().clone().to(dtype)
        trainable = n not in lib._non_trainable
        t.requires_grad_(trainable)
        reg[n] = t
        if not trainable:
            reg.non_trainable.add(n)
    return reg


def _cmp(a, b, tol, what, atol=1e-7):
    """max-norm relative check.  `atol` absorbs quantities that are analytically zero (e.g. the
    gradient of a bias that feeds a batch norm), where fp32 leaves O(1e-9) noise; for parameters
    Adam turns that noise into O(0.1*lr) steps on weights the network is invariant to."""
    a = a.detach().double().reshape(-1)
    b = b.detach().double().reshape(-1)
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    assert err <= tol * scale + atol, '%s: max err %.3e vs scale %.3e' % (what, err, scale)


@pytest.fixture
def small(cpu_kernels):
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    lib.set_seed(5)
    R.configure(DIM_G=8, DIM_D=8, BATCH_SIZE=4)
    R.build_params('cpu')
    yield R, lib
    R.configure()


def test_param_names_shapes_and_counts_full_width(cpu_kernels):
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    R.configure()
    R.build_params('cpu')
    nD = sum(p.numel() for _, p in lib.named_params_with_name('Discriminator.', True))
    nG = sum(p.numel() for _, p in lib.named_params_with_name('Generator', True))
    assert nD == 1055115 and nG == 1218307
    oreg = oref.Registry(dtype=torch.float32)
    cfg = onets.ResnetCfg()
    lab = torch.zeros(2, dtype=torch.int32)
    x = onets.resnet_generator(oreg, cfg, 2, lab, torch.zeros(2, 128))
    onets.resnet_discriminator(oreg, cfg, x, lab, 1., 1., 1.)
    assert {n: tuple(p.shape) for n, p in lib._params.items()} == {n: tuple(p.shape) for n, p in oreg.items()}
    assert lib._non_trainable == oreg.non_trainable
    # same init scheme (per-name streams): identical values for the same seed
    lib.delete_all_params(); lib.set_seed(0)
    R.build_params('cpu')
    for n in ('Discriminator.2.Conv1.Filters', 'Generator.Input.W', 'Generator.1.Shortcut.Filters'):
        assert torch.equal(lib._params[n].detach(), oreg[n].detach()), n


def test_forward_matches_oracle(small):
    R, lib = small
    reg = _oracle_from_product(lib)
    cfg = onets.ResnetCfg(DIM_G=8, DIM_D=8)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(6, 128, generator=g)
    lab = torch.randint(0, 10, (6,), generator=g, dtype=torch.int32)
    x = R.Generator(6, lab, noise=z)
    xo = onets.resnet_generator(reg, cfg, 6, lab, z.double())
    _cmp(x, xo, 2e-5, 'generator')
    u = [torch.rand(6, 8, 8, 8, generator=g) for _ in range(3)]
    d, f, a = R.Discriminator(x, lab, 0.8, 0.5, 0.5, u=u)
    do, fo, ao = onets.resnet_discriminator(reg, cfg, xo, lab, 0.8, 0.5, 0.5, [t.double() for t in u])
    _cmp(d, do, 5e-5, 'D'); _cmp(f, fo, 5e-5, 'D_'); _cmp(a, ao, 5e-5, 'acgan')
    # groups=2 == two separate towers
    x2 = R.Generator(6, lab, noise=z, groups=2)
    xa = onets.resnet_generator(reg, cfg, 3, lab[:3], z[:3].double())
    xb = onets.resnet_generator(reg, cfg, 3, lab[3:], z[3:].double())
    _cmp(x2, torch.cat([xa, xb]), 2e-5, 'generator groups=2')


def test_d_step_and_g_step_match_oracle(small):
    R, lib = small
    B = 4
    reg = _oracle_from_product(lib)
    cfg = onets.ResnetCfg(DIM_G=8, DIM_D=8)
    g = torch.Generator().manual_seed(1)
    real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
    rnd64 = osteps.make_rnd_resnet_d(B, 8, g)
    rnd32 = {k: ([t.float() for t in v] if isinstance(v, list) else v.float()) for k, v in rnd64.items()}
    tr = R.Trainer(seed=1)
    optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)

    for it in range(2):                                   # two D steps: checks Adam slot/beta-power carry-over
        out = tr.d_step(real, labels, rnd32, iteration=it)
        ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd64, iteration=it, B=B)
        for k in ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only', 'acc_real', 'acc_fake'):
            _cmp(out[k], ref[k], 2e-4, 'd_step[%d].%s' % (it, k))
        _cmp(out['gp_grads'], ref['gp_grads'], 2e-4, 'gp grads')
        assert set(out['grads']) == set(ref['grads'])
        for n in ref['grads']:
            _cmp(out['grads'][n], ref['grads'][n], 5e-4, 'dgrad ' + n)
        for n, _ in reg.trainable_with_name('Discriminator.'):
            _cmp(lib._params[n], reg[n], 5e-4, 'theta ' + n, atol=2e-5)

    rg64 = osteps.make_rnd_resnet_g(B, 8, g)
    rg32 = {'z': [t.float() for t in rg64['z']], 'label_u': [t.float() for t in rg64['label_u']],
            'u': [[t.float() for t in tw] for tw in rg64['u']]}
    out = tr.g_step(rg32, iteration=1)
    ref = osteps.resnet_g_step(reg, cfg, optG, rg64, iteration=1, B=B)
    _cmp(out['cost'], ref['cost'], 2e-4, 'g cost')
    assert set(out['grads']) == set(ref['grads'])
    for n in ref['grads']:
        _cmp(out['grads'][n], ref['grads'][n], 1e-3, 'ggrad ' + n)
    for n, _ in reg.trainable_with_name('Generator'):
        if ref['grads'][n].abs().max() < 1e-12:
            continue     # bias feeding a batch norm: analytically zero gradient, Adam amplifies fp32 noise
        _cmp(lib._params[n], reg[n], 1e-3, 'theta ' + n, atol=2e-5)
    # the critic received no update from the G step
    assert tr.d_opt.t == 2 and tr.g_opt.t == 1


def test_eager_rng_path_runs(small):
    """No injected randomness: draws come from DeviceRNG (mocked here) - finite losses, params move."""
    R, lib = small
    tr = R.Trainer(seed=3)
    g = torch.Generator().manual_seed(2)
    real = torch.randint(0, 256, (4, 3072), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 10, (4,), generator=g, dtype=torch.int32)
    before = tr.d_opt.theta.clone()
    it = iter(lambda: (real, labels), None)
    out = tr.train_iteration(1, lambda: next(it))
    assert torch.isfinite(out['cost']) and not torch.equal(before, tr.d_opt.theta)
    s, px = tr.generate_samples(torch.zeros(10, 128), torch.arange(10, dtype=torch.int32))
    assert s.shape == (10, 3072) and px.dtype == torch.int32 and px.min() >= 0 and px.max() <= 255


def test_steps_match_oracle_with_fused_resampling_convs(cpu_kernels, monkeypatch):
    """Width 32 switches ConvMeanPool / UpsampleConv to the single stride-2 (transposed) conv with the spread
    4x4 filter (functional.conv2d_mean_pool / upsample_conv2d): the oracle keeps the reference formulation
    (conv -> pool, upsample -> conv; TF/CT_gan_cifar_resnet.py:89-107), so this is the parity check of the
    restructuring through a whole critic step (incl. the gradient-penalty double backward) and generator step."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    calls = {'pool': 0, 'up': 0}
    orig_pool, orig_up = F.conv2d_mean_pool, F.upsample_conv2d
    monkeypatch.setattr(F, 'conv2d_mean_pool', lambda *a, **k: (calls.__setitem__('pool', calls['pool'] + 1), orig_pool(*a, **k))[1])
    monkeypatch.setattr(F, 'upsample_conv2d', lambda *a, **k: (calls.__setitem__('up', calls['up'] + 1), orig_up(*a, **k))[1])
    B, dim = 2, 32
    lib.set_seed(11)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    try:
        R.build_params('cpu')
        reg = _oracle_from_product(lib)
        cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
        g = torch.Generator().manual_seed(3)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd64 = osteps.make_rnd_resnet_d(B, dim, g)
        rnd32 = {k: ([t.float() for t in v] if isinstance(v, list) else v.float()) for k, v in rnd64.items()}
        tr = R.Trainer(seed=1)
        optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
        optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
        out = tr.d_step(real, labels, rnd32, iteration=0)
        ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd64, iteration=0, B=B)
        assert calls['pool'] > 0 and calls['up'] > 0
        for k in ('cost', 'wgan', 'acgan', 'ct', 'gp'):
            _cmp(out[k], ref[k], 2e-4, 'd_step.%s' % k)
        _cmp(out['gp_grads'], ref['gp_grads'], 2e-4, 'gp grads')
        for n in ref['grads']:
            _cmp(out['grads'][n], ref['grads'][n], 5e-4, 'dgrad ' + n)
        rg64 = osteps.make_rnd_resnet_g(B, dim, g)
        rg32 = {'z': [t.float() for t in rg64['z']], 'label_u': [t.float() for t in rg64['label_u']],
                'u': [[t.float() for t in tw] for tw in rg64['u']]}
        out = tr.g_step(rg32, iteration=1)
        ref = osteps.resnet_g_step(reg, cfg, optG, rg64, iteration=1, B=B)
        _cmp(out['cost'], ref['cost'], 2e-4, 'g cost')
        for n in ref['grads']:
            _cmp(out['grads'][n], ref['grads'][n], 1e-3, 'ggrad ' + n)
    finally:
        R.configure()


def test_batched_fake_generation_equals_per_step_generation(small):
    """Trainer.generate_fakes draws the fake batches of all N_CRITIC critic steps in one generator forward with one
    pair of BN statistic groups per step: row block i equals the generator run on step i's noise / labels alone."""
    R, lib = small
    g = torch.Generator().manual_seed(4)
    B, NC = 4, 3
    z = torch.randn(NC * B, 128, generator=g)
    lab = torch.randint(0, 10, (NC * B,), generator=g, dtype=torch.int32)
    allx = R.Generator(NC * B, lab, noise=z, groups=2 * NC)
    for i in range(NC):
        xi = R.Generator(B, lab[i * B:(i + 1) * B], noise=z[i * B:(i + 1) * B], groups=2)
        _cmp(allx[i * B:(i + 1) * B], xi, 1e-6, 'step %d' % i)
    tr = R.Trainer(seed=3)
    fakes = tr.generate_fakes(lab[:2 * B])
    assert tuple(fakes.shape) == (2, B, 3072) and torch.isfinite(fakes).all()
    real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
    out = tr.d_step(real, lab[:B], fake=fakes[0])
    assert torch.isfinite(out['cost']) and torch.equal(out['fake'], fakes[0])


def test_deferred_multi_segment_wgrads_equal_immediate_wgrads(cpu_kernels):
    """functional.deferred_wgrads queues the weight gradients of a critic step per filter and launches one
    multi-segment wgrad per filter on exit (dropout passes + GP double backward summed in the kernel): the parameter
    gradients must equal those of the immediate path, for plain, pooled (spread-filter) and shortcut convs."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    B, dim = 4, 32
    lib.set_seed(7)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    try:
        R.build_params('cpu')
        tr = R.Trainer(seed=1)
        g = torch.Generator().manual_seed(100)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd = osteps.make_rnd_resnet_d(B, dim, g, dtype=torch.float32)
        res, ngroups = {}, 0
        for mode in (False, True):
            F.DEFER_WGRADS = mode
            tr.rng.begin_step()
            out = tr.d_losses(real, labels, rnd)
            with F.deferred_wgrads():
                grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
                if mode:
                    ngroups = len(F._DEFER['groups'])
                    assert all(len(grp.segs) == 2 for grp in F._DEFER['groups'].values())     # main pass + GP double backward
            res[mode] = [None if x is None else x.clone() for x in grads]
        assert ngroups >= 8
        for (n, _), a, b in zip(tr.d_named, res[False], res[True]):
            assert (a is None) == (b is None), n
            if a is not None:
                _cmp(b, a, 1e-5, 'deferred ' + n, atol=1e-7)
    finally:
        F.DEFER_WGRADS = True
        R.configure()


@pytest.mark.parametrize('min_px', [1, 2048, 20000])
def test_split_mode_wgrads_written_in_place_equal_immediate_wgrads(cpu_kernels, monkeypatch, min_px):
    """functional._wgrad with uses the split-mode kernel takes at request time (kernels.wgrad_prefers_x3): the FIRST such use of a
    filter writes straight into the filter's result buffers (bias gradient included), the queued fp32 segments are accumulated onto
    it inside the grouped reduction (ctgan_wgrad_group.add_dw / add_db aliasing dw / db), further uses are added at the flush.  For
    every routing threshold - everything in place (1 pixel), main pass in place + GP segment queued (2048), nothing (20000: more pixels than any use has) - the
    parameter gradients of a full critic step must equal the immediate path's."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    B, dim = 4, 32
    lib.set_seed(7)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    try:
        R.build_params('cpu')
        tr = R.Trainer(seed=1)
        g = torch.Generator().manual_seed(100)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd = osteps.make_rnd_resnet_d(B, dim, g, dtype=torch.float32)
        res, seen = {}, {}
        for mode in (False, True):
            F.DEFER_WGRADS = mode
            monkeypatch.setattr(cpu_kernels, 'PREFERS_X3_MIN_PIXELS', min_px if mode else None)
            tr.rng.begin_step()
            out = tr.d_losses(real, labels, rnd)
            with F.deferred_wgrads():
                grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
                if mode:
                    grps = list(F._DEFER['groups'].values())
                    seen = {'inplace': sum(gr.inplace for gr in grps), 'with_segs': sum(bool(gr.segs) and gr.inplace for gr in grps),
                            'extra': sum(len(gr.pre) for gr in grps), 'bias_inplace': sum(gr.inplace_b for gr in grps)}
            res[mode] = [None if x is None else x.clone() for x in grads]
        if min_px == 1:
            assert seen['inplace'] >= 8 and seen['extra'] >= 8 and seen['bias_inplace'] >= 6, seen      # both uses taken at once
        elif min_px == 2048:
            assert seen['with_segs'] >= 2, seen          # main pass in place, GP segment queued behind it
        else:
            assert seen['inplace'] == 0, seen
        for (n, _), a, b in zip(tr.d_named, res[False], res[True]):
            assert (a is None) == (b is None), n
            if a is not None:
                _cmp(b, a, 1e-5, 'in-place split-mode wgrad ' + n, atol=1e-7)
    finally:
        F.DEFER_WGRADS = True
        R.configure()


def test_fused_critic_heads_equal_separate_heads(small):
    """Host wiring of HEAD_FUSION (CriticTailHeadsFn, GpHeadGradFn, ConvFn 'mask_done') against the op-by-op head, on the
    CPU stand-ins and identical random streams."""
    import ctgan_amd.functional as F
    R, lib = small
    B = R.cfg.BATCH_SIZE
    g = torch.Generator().manual_seed(3)
    real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
    lab = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
    tr = R.Trainer(seed=4)
    fake = tr.generate_fakes(lab)[0]
    res = {}
    for mode in (False, True):
        R.HEAD_FUSION = R.PREP_FUSION = R.TRUNK_SHARE = R.TAIL_SHARE = mode            # also: input preparation / concat+dropout in single launches
        try:
            tr.rng.begin_step()
            out = tr.d_losses(real, lab, fake=fake)
            grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
            res[mode] = ({k: out[k].detach().clone() for k in ('cost', 'wgan', 'ct', 'acgan', 'gp', 'd_real', 'd_fake', 'gp_grads')},
                         [None if t is None else t.detach().clone() for t in grads])
        finally:
            R.HEAD_FUSION = R.PREP_FUSION = R.TRUNK_SHARE = R.TAIL_SHARE = True
    for k, v in res[True][0].items():
        _cmp(v, res[False][0][k], 1e-5, k)
    for (n, _), a_, b_ in zip(tr.d_named, res[True][1], res[False][1]):
        assert (a_ is None) == (b_ is None), n
        if a_ is not None:
            _cmp(a_, b_, 2e-5, n)


def test_fused_generator_heads_equal_separate_heads(small):
    """Host wiring of GenTailHeadsFn against the op-by-op generator loss on the CPU stand-ins and identical random streams."""
    R, lib = small
    tr = R.Trainer(seed=6)
    res = {}
    for mode in (False, True):
        R.HEAD_FUSION = mode
        try:
            tr.rng.begin_step()
            out = tr.g_losses()
            grads = torch.autograd.grad(out['cost'], tr.g_params, allow_unused=True)
            res[mode] = (out['cost'].detach().clone(), [None if t is None else t.detach().clone() for t in grads])
        finally:
            R.HEAD_FUSION = True
    _cmp(res[True][0], res[False][0], 1e-5, 'cost')
    for (n, _), a_, b_ in zip(tr.g_named, res[True][1], res[False][1]):
        assert (a_ is None) == (b_ is None), n
        if a_ is not None:
            _cmp(a_, b_, 5e-5, n, atol=1e-6)


def test_layernorm_critic_step_matches_oracle(cpu_kernels):
    """NORMALIZATION_D=True (TF/CT_gan_cifar_resnet.py:76-77: Layernorm after every critic conv input): the critic is no
    longer piecewise linear, so the gradient penalty differentiates Layernorm twice - forward, losses and every critic
    gradient of a D step against the oracle."""
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    B, dim = 4, 8
    lib.set_seed(13)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B, NORMALIZATION_D=True)
    try:
        R.build_params('cpu')
        assert any('.N1.scale' in n for n in lib._params if n.startswith('Discriminator.'))
        reg = _oracle_from_product(lib)
        cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim, NORMALIZATION_D=True)
        g = torch.Generator().manual_seed(5)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd64 = osteps.make_rnd_resnet_d(B, dim, g)
        rnd32 = {k: ([t.float() for t in v] if isinstance(v, list) else v.float()) for k, v in rnd64.items()}
        tr = R.Trainer(seed=1)
        optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
        out = tr.d_step(real, labels, rnd32, iteration=0)
        ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd64, iteration=0, B=B)
        for k in ('cost', 'wgan', 'acgan', 'ct', 'gp'):
            _cmp(out[k], ref[k], 5e-4, 'd_step.%s' % k)
        _cmp(out['gp_grads'], ref['gp_grads'], 5e-4, 'gp grads')
        assert set(out['grads']) == set(ref['grads'])
        for n in ref['grads']:
            _cmp(out['grads'][n], ref['grads'][n], 2e-3, 'dgrad ' + n, atol=1e-6)
    finally:
        R.configure()


def test_deferred_wgrads_mixed_operand_layouts_stay_one_queue(cpu_kernels, monkeypatch):
    """Two uses of ONE filter whose operands have different memory layouts (channels-last vs plain NCHW) under
    deferred_wgrads: the second use joins the first one's queue (repacked), so autograd receives exactly one gradient
    buffer for the parameter and it is filled by the flush.  Every torch.empty* is NaN-poisoned: a gradient that was summed
    before the flush wrote it would be NaN."""
    import ctgan_amd.functional as F
    from ctgan_amd.kernels import ConvGeom
    for name in ('empty', 'empty_like', 'empty_strided'):
        real = getattr(torch, name)

        def poisoned(*a, _real=real, **k):
            t = _real(*a, **k)
            return t.fill_(float('nan')) if t.is_floating_point() else t
        monkeypatch.setattr(torch, name, poisoned)
    g = torch.Generator().manual_seed(3)
    C, K_, H = 32, 32, 6
    w = torch.nn.Parameter(torch.randn(3, 3, C, K_, generator=g) * 0.1)
    x_cl = torch.randn(2, H, H, C, generator=g).permute(0, 3, 1, 2)                 # channels-last
    x_nchw = torch.randn(3, C, H, H, generator=g)                                    # plain NCHW, another batch size
    geom = ConvGeom(C, H, H, K_, 3, 3, 1, False)

    def loss():
        ya = F.ConvFn.apply(x_cl, w, None, None, geom, None)
        yb = F.ConvFn.apply(x_nchw, w, None, None, geom, (K_ * H * H, H * H, H, 1))    # NCHW result: its gradient is NCHW too
        return (ya * ya).sum() + (yb * yb).sum()
    (ref,) = torch.autograd.grad(loss(), [w])
    with F.deferred_wgrads():
        (got,) = torch.autograd.grad(loss(), [w])
        assert len(F._DEFER['groups']) == 1 and len(next(iter(F._DEFER['groups'].values())).segs) == 2
    assert torch.isfinite(got).all()
    _cmp(got, ref, 1e-5, 'mixed-layout deferred wgrad', atol=1e-6)


def test_default_fused_step_matches_oracle_on_identical_philox_streams(cpu_kernels):
    """The DEFAULT path (rnd=None: shared trunk / tail forward on the tape, per-row-range dropout, fused heads and input
    preparation, grouped deferred weight gradients) against the oracle fed the same Philox streams (oracle/philox.py): the
    host logic of the benchmarked step, not of the op-by-op parity mode.  (The stand-ins draw the device's streams.)"""
    from oracle import philox
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    B, dim = 4, 32
    lib.set_seed(5)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    try:
        R.build_params('cpu')
        reg = _oracle_from_product(lib)
        cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
        g = torch.Generator().manual_seed(1)
        tr = R.Trainer(seed=77)
        assert R._heads_fusable(None, tr.rng) and R.TRUNK_SHARE and R.TAIL_SHARE
        optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
        optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
        pos = 0
        for it in range(2):
            real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
            labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
            out = tr.d_step(real, labels, None, iteration=it)
            ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, philox.rnd_resnet_d(77, 0, pos, B, dim), iteration=it, B=B)
            pos += 1
            for k in ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only', 'acc_real', 'acc_fake'):
                _cmp(out[k], ref[k], 2e-4, 'd_step[%d].%s' % (it, k), atol=1e-6)
            _cmp(out['gp_grads'], ref['gp_grads'], 5e-4, 'gp grads')
            assert set(out['grads']) == set(ref['grads'])
            for n in ref['grads']:
                _cmp(out['grads'][n], ref['grads'][n], 1e-3, 'dgrad ' + n, atol=1e-6)
        out = tr.g_step(None, iteration=1)
        ref = osteps.resnet_g_step(reg, cfg, optG, philox.rnd_resnet_g(77, 0, pos, B, dim), iteration=1, B=B)
        _cmp(out['cost'], ref['cost'], 2e-4, 'g cost')
        _cmp(out['samples'], torch.cat(ref['samples']), 1e-4, 'g samples')
        for n in ref['grads']:
            _cmp(out['grads'][n], ref['grads'][n], 2e-3, 'ggrad ' + n, atol=1e-6)
    finally:
        R.configure()


def test_gp_double_backward_mask_rides_the_consumers_conv_epilogue(cpu_kernels, monkeypatch):
    """functional.PREMASK_FUSION: in the double backward of the gradient penalty, the ReLU mask a block's second data-gradient node
    applies to what arrives for it is taken by the conv epilogue of the node that produces it (conv_fwd(mask=...)) - one mask pass per
    residual block less; a node that drops and masks does both in one launch (dropout_rng_mask) - and the parameter gradients of the
    critic step do not change."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.kernels as K
    import ctgan_amd.tflib as lib
    B, dim = 4, 32
    lib.set_seed(9)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    try:
        R.build_params('cpu')
        tr = R.Trainer(seed=3)
        g = torch.Generator().manual_seed(5)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        calls = {'mask_pass': 0, 'masked_conv': 0, 'drop_mask': 0}
        real_bwd, real_fwd, real_dm = K.lrelu_bwd, K.conv_fwd, K.dropout_rng_mask

        def count_dm(*a, **k):
            calls['drop_mask'] += 1
            return real_dm(*a, **k)

        def count_bwd(*a, **k):
            calls['mask_pass'] += 1
            return real_bwd(*a, **k)

        def count_fwd(*a, **k):
            if k.get('mask') is not None:
                calls['masked_conv'] += 1
            return real_fwd(*a, **k)
        monkeypatch.setattr(K, 'lrelu_bwd', count_bwd)
        monkeypatch.setattr(K, 'conv_fwd', count_fwd)
        monkeypatch.setattr(K, 'dropout_rng_mask', count_dm)
        res, seen = {}, {}
        for mode in (False, True):
            monkeypatch.setattr(F, 'PREMASK_FUSION', mode)
            calls.update(mask_pass=0, masked_conv=0, drop_mask=0)
            F.prepare_filters()
            tr.rng.begin_step()
            out = tr.d_losses(real, labels)
            with F.deferred_wgrads():
                grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
            res[mode], seen[mode] = [None if x is None else x.clone() for x in grads], dict(calls)
        assert seen[False]['masked_conv'] == 0
        assert seen[True]['masked_conv'] == 4, seen     # one per residual block of the critic
        assert seen[True]['drop_mask'] == 2 and seen[False]['drop_mask'] == 0, seen     # blocks 3 and 4: dropout + mask in one launch
        assert seen[True]['mask_pass'] == seen[False]['mask_pass'] - 6, seen
        for (n, _), a, b in zip(tr.d_named, res[False], res[True]):
            assert (a is None) == (b is None), n
            if a is not None:
                assert torch.equal(a, b), n
    finally:
        R.configure()


def test_penalty_mean_and_clean_pass_accuracies_folded_into_the_heads_launches(cpu_kernels, monkeypatch):
    """gan_cifar_resnet.HEADS_FOLD: gradient_penalty(defer_mean=True) leaves gp as a slot that critic_tail_heads fills from the slopes,
    and the clean pass's class head + accuracies are computed by the same call (y_clean) - cost, wgan / ct / gp terms, accuracies and every
    parameter gradient must equal the unfolded path's."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    B, dim = 4, 32
    lib.set_seed(11)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    try:
        R.build_params('cpu')
        tr = R.Trainer(seed=4)
        g = torch.Generator().manual_seed(6)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        res = {}
        for mode in (False, True):
            monkeypatch.setattr(R, 'HEADS_FOLD', mode)
            F.prepare_filters()
            tr.rng.begin_step()
            out = tr.d_losses(real, labels)
            with F.deferred_wgrads():
                grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
            res[mode] = ({k: out[k].detach().clone() for k in ('cost', 'wgan', 'ct', 'gp', 'acgan', 'acc_real', 'acc_fake')},
                         [None if x is None else x.clone() for x in grads])
        for k, v in res[False][0].items():
            assert torch.isfinite(v).all() and torch.allclose(res[True][0][k], v, rtol=1e-6, atol=1e-7), k
        assert float(res[True][0]['gp']) > 0
        for (n, _), a, b in zip(tr.d_named, res[False][1], res[True][1]):
            assert (a is None) == (b is None), n
            if a is not None:
                _cmp(b, a, 1e-6, 'folded heads ' + n, atol=1e-8)
    finally:
        R.configure()


def test_premask_links_do_not_keep_the_backward_graph_alive(cpu_kernels, monkeypatch):
    """The producer/consumer links of functional.PREMASK_FUSION must die with the step by reference counting alone: a link that held the
    data gradient ITSELF (whose grad_fn owns the link) was a reference cycle, freed only by the cycle collector - at an arbitrary later
    point, once inside hipStreamEndCapture (segfault of `bench.py --gp-unit-only`)."""
    import gc
    import weakref
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    B, dim = 4, 32
    lib.set_seed(13)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    made = []

    class Tracked(F._PreMask):
        def __init__(self):
            made.append(weakref.ref(self))
    monkeypatch.setattr(F, '_PreMask', Tracked)
    try:
        R.build_params('cpu')
        tr = R.Trainer(seed=5)
        g = torch.Generator().manual_seed(7)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        gc.collect()
        gc.disable()
        try:
            F.prepare_filters()
            tr.rng.begin_step()
            out = tr.d_losses(real, labels)
            with F.deferred_wgrads():
                grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
            assert len(made) >= 4
            del out, grads
            assert all(r() is None for r in made), 'a premask link survived the step without the cycle collector'
        finally:
            gc.enable()
    finally:
        R.configure()


def test_both_optimizers_share_one_state_allocation_and_one_learning_rate_fill(small):
    """Trainer.set_lr: one fill writes the learning rate of the critic's and the generator's Adam state (slices of one allocation);
    their beta powers stay independent; a repeated value is not written again."""
    R, lib = small
    tr = R.Trainer(seed=1)
    assert tr.d_opt.state.data_ptr() == tr._opt_state.data_ptr() and tr.g_opt.state.data_ptr() == tr._opt_state[4:].data_ptr()
    tr.set_lr(3e-4)
    assert float(tr.d_opt.state[0]) == pytest.approx(3e-4) and float(tr.g_opt.state[0]) == pytest.approx(3e-4)
    assert float(tr.d_opt.state[1]) == 0.0 and float(tr.d_opt.state[2]) == pytest.approx(0.9)       # beta1 = 0, beta2 = 0.9 at t = 1
    tr._opt_state[0::4].fill_(7.0)
    tr.set_lr(3e-4)                                       # same value as the last call: skipped
    assert float(tr.d_opt.state[0]) == 7.0
    tr.set_lr(2e-4)
    assert float(tr.g_opt.state[0]) == pytest.approx(2e-4)
    tr.d_opt.grad.zero_()
    tr.d_opt.step(rng=tr.rng)                             # the critic's step advances ITS powers and the Philox counter only
    assert float(tr.d_opt.state[2]) == pytest.approx(0.81) and float(tr.g_opt.state[2]) == pytest.approx(0.9) and int(tr.rng.ctr) == 1
    sd = tr.d_opt.state_dict()
    tr.d_opt.load_state_dict(sd)
    assert float(tr.g_opt.state[0]) == pytest.approx(2e-4)


def _scheduled_vs_autograd(R, lib, F, B, D, ac, dev, seed=1):
    """One critic step's losses and parameter gradients through critic_schedule.critic_step (merged=True) and through d_losses +
    autograd (merged=False), from identical weights, inputs and Philox streams."""
    import ctgan_amd.critic_schedule as CS
    res = {}
    for merged in (False, True):
        lib.delete_all_params(); lib.set_device(dev); lib.set_seed(seed)
        R.configure(DIM_G=32, DIM_D=D, BATCH_SIZE=B, CONDITIONAL=ac, ACGAN=ac)
        R.build_params()
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for n, p in lib._params.items():
                if n.endswith(('.Biases', '.b')):           # zero-initialised in the reference: make them count
                    p.add_((0.1 * torch.randn(p.shape, generator=g)).to(p.device))
        tr = R.Trainer(seed=3)
        real = torch.randint(0, 256, (B, 3072), dtype=torch.int32, generator=g).to(tr.dev)
        labels = torch.randint(0, 10, (B,), dtype=torch.int32, generator=g).to(tr.dev)
        old, CS.MERGED_BWD = CS.MERGED_BWD, merged
        try:
            F.prepare_filters()
            fake = tr.generate_fakes(labels)[0]
            assert CS.usable(R, None, tr.rng, real, fake) == merged
            tr.rng.begin_step()
            out, grads = tr.d_grads(real, labels, fake=fake)
        finally:
            CS.MERGED_BWD = old
        res[merged] = (out, grads, [n for n, _ in tr.d_named])
    return res[False], res[True]


@pytest.mark.parametrize('ac,mode', [(True, None), (False, None), (False, 'bf16'), (False, 'f32x3')])
def test_hand_scheduled_critic_step_equals_the_autograd_path(cpu_kernels, ac, mode):
    """critic_schedule.critic_step - ONE backward chain over the rows of the two dropout passes and the rows of the gradient-penalty
    pass, weight gradients from the dropout-pass rows only, the penalty's double backward on the x_hat rows only -
    against the path it replaces (Trainer.d_losses + two torch.autograd.grad calls): every loss term, the slopes, dD/dx_hat and every
    parameter gradient, with and without the class head."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    import ctgan_amd.kernels as K
    try:
        # modes 'bf16' / 'f32x3' (the stand-ins still multiply in fp32): no filter is queued there - every use returns a finished weight
        # gradient, and the schedule has to fold each spread-filter gradient itself before summing the uses
        with K.mma_dtype(mode):
            a, b = _scheduled_vs_autograd(R, lib, F, 2, 64, ac, 'cpu')
        for k in ('cost', 'wgan', 'acgan', 'wgan_only', 'ct', 'gp', 'acc_real', 'acc_fake', 'slopes', 'gp_grads', 'd_real', 'd_fake', 'real'):
            assert (a[0].get(k) is None) == (b[0].get(k) is None), k
            if a[0].get(k) is not None:
                _cmp(b[0][k], a[0][k], 2e-6, 'scheduled.' + k, atol=1e-7)
        assert a[2] == b[2]
        for n, x, y in zip(a[2], a[1], b[1]):
            assert (x is None) == (y is None), n
            if x is not None:
                assert tuple(x.shape) == tuple(dict(lib.named_params_with_name('Discriminator.', True))[n].shape), n
                _cmp(y, x, 5e-6, 'scheduled grad ' + n, atol=1e-8)
    finally:
        R.configure()


def test_hand_scheduled_critic_step_with_the_early_asynchronous_weight_gradient_flush(cpu_kernels, monkeypatch):
    """functional.flush_async (CTGAN_WGRAD_OVERLAP=1; off by default - measured slower): the weight gradients of the dropout-pass rows are
    launched when the backward chain has produced them, the penalty's double-backward segments are accumulated onto those results by the final flush
    (the in-place form of the grouped launch).  Same gradients as the autograd path, incl. the folded spread filters and the bias gradients that
    only the early launch computes."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    monkeypatch.setattr(F, 'WGRAD_OVERLAP', True)
    fired = []
    orig = F.flush_async
    monkeypatch.setattr(F, 'flush_async', lambda: (fired.append(orig()), fired[-1])[1])
    try:
        a, b = _scheduled_vs_autograd(R, lib, F, 2, 64, True, 'cpu')
        assert fired == [True]
        for n, x, y in zip(a[2], a[1], b[1]):
            assert (x is None) == (y is None), n
            if x is not None:
                _cmp(y, x, 5e-6, 'scheduled grad (early flush) ' + n, atol=1e-8)
    finally:
        R.configure()


def test_hand_scheduled_critic_step_matches_oracle_on_identical_philox_streams(cpu_kernels):
    """The scheduled step (DIM_D = 64: the width from which the first critic convs run on the direct few-channel kernels, which the
    schedule requires) through Trainer.d_step against the fp64 oracle of the reference graph AS WRITTEN, fed the same Philox streams."""
    from oracle import philox
    import ctgan_amd.critic_schedule as CS
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    B, dim = 2, 64
    lib.set_seed(5)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    try:
        R.build_params('cpu')
        reg = _oracle_from_product(lib)
        cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
        g = torch.Generator().manual_seed(1)
        tr = R.Trainer(seed=77)
        optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
        calls = []
        orig = CS.critic_step
        CS.critic_step = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            for it in range(2):
                real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
                labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
                out = tr.d_step(real, labels, None, iteration=it)
                ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, philox.rnd_resnet_d(77, 0, it, B, dim), iteration=it, B=B)
                for k in ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only', 'acc_real', 'acc_fake'):
                    _cmp(out[k], ref[k], 2e-4, 'd_step[%d].%s' % (it, k), atol=1e-6)
                _cmp(out['gp_grads'], ref['gp_grads'], 5e-4, 'gp grads')
                assert set(n for n, v in out['grads'].items() if v is not None) == set(ref['grads'])
                for n in ref['grads']:
                    _cmp(out['grads'][n], ref['grads'][n], 1e-3, 'dgrad ' + n, atol=1e-6)
        finally:
            CS.critic_step = orig
        assert len(calls) == 2, 'the default critic step did not take the hand-scheduled path'
    finally:
        R.configure()
