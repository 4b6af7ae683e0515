"""Host logic of the two DCGAN scripts (Deconv2D as conv-dgrad, LeakyReLU, BN axes [0], MNIST crop,
3B-row batched critic, GP double backward) against the oracle, HIP wrappers swapped for CPU stand-ins."""
import pytest
import torch

from oracle import nets as onets, steps as osteps, tflib_ref as oref


def _oracle_from_product(lib, dtype=torch.float64):
    reg = oref.Registry(dtype=dtype)
    for n, p in lib._params.items():
        t = p.detach().clone().to(dtype)
        tr = n not in lib._non_trainable
        reg[n] = t.requires_grad_(tr)
        if not tr:
            reg.non_trainable.add(n)
    return reg


def _cmp(a, b, tol, what, atol=1e-7):
    a = a.detach().double().reshape(-1); b = b.detach().double().reshape(-1)
    err = (a - b).abs().max().item(); scale = b.abs().max().item()
    assert err <= tol * scale + atol, '%s: max err %.3e vs scale %.3e' % (what, err, scale)


def _f32(o):
    if isinstance(o, list):
        return [_f32(t) for t in o]
    return o.float()


@pytest.mark.parametrize('which', ['cifar', 'mnist'])
def test_dcgan_steps_match_oracle(cpu_kernels, which):
    import ctgan_amd.tflib as lib
    from ctgan_amd.dcgan_step import DCGANTrainer
    if which == 'cifar':
        import ctgan_amd.gan_cifar as M
        M.configure(DIM=8, BATCH_SIZE=4)
        G = lambda reg, n, z: onets.cifar_generator(reg, n, z, DIM=8)          # noqa: E731
        D = lambda reg, x, u: onets.cifar_discriminator(reg, x, u, DIM=8)      # noqa: E731
        g = torch.Generator().manual_seed(3)
        real_in = torch.randint(0, 256, (4, 3072), generator=g, dtype=torch.int32)
        real_o = 2 * ((real_in.double() / 255.) - .5)
    else:
        import ctgan_amd.gan_mnist as M
        M.configure(DIM=8, BATCH_SIZE=4)
        G = lambda reg, n, z: onets.mnist_generator(reg, n, z, DIM=8)          # noqa: E731
        D = lambda reg, x, u: onets.mnist_discriminator(reg, x, u, DIM=8)      # noqa: E731
        g = torch.Generator().manual_seed(4)
        real_in = torch.rand(4, 784, generator=g)
        real_o = real_in.double()
    try:
        lib.set_seed(9)
        with torch.no_grad():
            x = M.Generator(2, noise=torch.zeros(2, 128))
            M.Discriminator(x, u=[torch.full((2,) + s, 0.9) for s in M.feat_shapes()])
        tr = DCGANTrainer(M, seed=1)
        reg = _oracle_from_product(lib)
        optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator')], 0.5, 0.9)
        optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.5, 0.9)
        rnd = osteps.make_rnd_dcgan_d(4, M.feat_shapes(), g)
        out = tr.d_step(real_in, {k: _f32(v) for k, v in rnd.items()})
        ref = osteps.dcgan_d_losses(reg, G, D, real_o, rnd)
        gref = osteps.grads_of(ref['cost'], reg, 'Discriminator')
        for k in ('cost', 'wgan_only', 'ct'):
            _cmp(out[k], ref[k], 2e-4, which + ' d.' + k)
        _cmp(out['gp'], M.cfg.LAMBDA * ref['gp'], 2e-4, which + ' d.gp')      # product reports LAMBDA*gp
        _cmp(out['fake'], ref['fake'], 5e-5, which + ' fake')
        for n in gref:
            _cmp(out['grads'][n], gref[n], 1e-3, which + ' dgrad ' + n, atol=2e-6)
        optD.apply(gref, 1e-4)
        for n in optD.names:
            if gref[n].abs().max() > 1e-12:
                _cmp(lib._params[n], reg[n], 1e-3, 'theta ' + n, atol=2e-5)
        lib.load_state_dict({n: t.detach().float() for n, t in reg.items()})
        rg = osteps.make_rnd_dcgan_g(4, M.feat_shapes(), g)
        out = tr.g_step({k: _f32(v) for k, v in rg.items()})
        ref = osteps.dcgan_g_losses(reg, G, D, 4, rg)
        _cmp(out['cost'], ref['cost'], 2e-4, which + ' g cost')
        gref = osteps.grads_of(ref['cost'], reg, 'Generator')
        assert set(n for n, v in out['grads'].items() if v is not None) == set(gref)
        for n in gref:
            _cmp(out['grads'][n], gref[n], 2e-3, which + ' ggrad ' + n, atol=2e-6)
    finally:
        M.configure()


def test_loss_scale_is_divided_out_exactly(cpu_kernels):
    """DCGANTrainer.loss_scale (the fp16 mode's power-of-two loss scale): seeds the final backward with S and hands Adam 1/S - in fp32
    a power of two changes no bit: reported gradients, updated weights and Adam slots equal the unscaled step's."""
    import ctgan_amd.gan_mnist as M
    import ctgan_amd.tflib as lib
    from ctgan_amd.dcgan_step import DCGANTrainer
    from oracle import steps as osteps
    res = {}
    for S in (1.0, 1024.0):
        lib.delete_all_params(); lib.set_device('cpu'); lib.set_seed(3)
        M.configure(BATCH_SIZE=4, DIM=8)
        try:
            with torch.no_grad():
                M.Discriminator(M.Generator(2, noise=torch.zeros(2, 128)), u=[torch.ones(2, *s) for s in M.feat_shapes()])
            tr = DCGANTrainer(M, seed=1)
            tr.loss_scale = S
            g = torch.Generator().manual_seed(5)
            real = torch.rand(4, M.cfg.OUTPUT_DIM, generator=g)
            rnd = osteps.make_rnd_dcgan_d(4, M.feat_shapes(), g, dtype=torch.float32)
            out = tr.d_step(real, rnd)
            rg = osteps.make_rnd_dcgan_g(4, M.feat_shapes(), g, dtype=torch.float32)
            og = tr.g_step(rg)
            res[S] = ([v.clone() for v in out['grads'].values() if v is not None], tr.d_opt.theta.clone(), tr.d_opt.m.clone(), tr.d_opt.v.clone(),
                      tr.g_opt.theta.clone(), og['cost'].item())
        finally:
            M.configure()
    a, b = res[1.0], res[1024.0]
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0]))
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]) and a[5] == b[5]


@pytest.mark.parametrize('net', ['cifar', 'lsun128'])
def test_queued_weight_gradients_equal_immediate_ones_in_the_dcgan_family_steps(cpu_kernels, net):
    """The DCGAN-family steps queue their weight gradients (`with F.deferred_wgrads():` around the backward, as the ResNet step; DESIGN
    4.7): every parameter gradient of a critic step and of a generator step must equal the immediate path's -
    several uses per filter (dropout passes, GP double backward through Layernorm for the 128x128 ResNet), few-channel layers, spread
    filters."""
    import ctgan_amd.functional as F
    import ctgan_amd.tflib as lib
    from ctgan_amd.dcgan_step import DCGANTrainer
    if net == 'cifar':
        import ctgan_amd.gan_cifar as M
        M.configure(DIM=32, BATCH_SIZE=4)
    else:
        import ctgan_amd.gan_lsun128 as M
        M.configure(BATCH_SIZE=2, DIM_G_64=4, DIM_G_32=8, DIM_G_16=8, DIM_G_8=16, DIM_G_4=16, DIM_D_64=8, DIM_D_32=8, DIM_D_16=16, DIM_D_8=16)
    try:
        B = M.cfg.BATCH_SIZE
        lib.set_seed(21)
        if hasattr(M, 'build_params'):
            M.build_params('cpu')
        else:                                   # the DCGAN scripts create their parameters lazily: one forward of each net
            with torch.no_grad():
                x0 = M.Generator(2, noise=torch.zeros(2, 128))
                M.Discriminator(x0, u=[torch.full((2,) + s, 0.9) for s in M.feat_shapes()])
        tr = DCGANTrainer(M, seed=2)
        g = torch.Generator().manual_seed(8)
        real = torch.randint(0, 256, (B, M.cfg.OUTPUT_DIM), generator=g, dtype=torch.int32)
        res = {}
        for queued in (False, True):
            outs = []
            for which in ('d', 'g'):
                tr.rng.begin_step()
                out = tr.d_losses(real, None) if which == 'd' else tr.g_losses(None)
                params = tr.d_params if which == 'd' else tr.g_params
                if queued:
                    with F.deferred_wgrads():
                        grads = torch.autograd.grad(out['cost'], params, allow_unused=True)
                else:
                    grads = torch.autograd.grad(out['cost'], params, allow_unused=True)
                outs.append([None if x is None else x.clone() for x in grads])
            res[queued] = outs
        for a_list, b_list, named in zip(res[False], res[True], (tr.d_named, tr.g_named)):
            for (n, _), a, b in zip(named, a_list, b_list):
                assert (a is None) == (b is None), n
                if a is not None:
                    _cmp(b, a, 2e-5, 'queued wgrad ' + n, atol=1e-7)
    finally:
        M.configure()


def test_fused_lrelu_dropout_equals_the_two_ops_through_the_double_backward(cpu_kernels):
    """F.lrelu_dropout (one launch each way) against dropout(leaky_relu(x)) on the same Philox stream: value, gradient, and the gradient
    of a function of that gradient (the gradient penalty differentiates the critic twice; the pair is piecewise linear, so the second
    derivative w.r.t. x is zero and the double backward reaches only the seed)."""
    import ctgan_amd.functional as F
    from ctgan_amd.rng import DeviceRNG
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(4, 8, 6, 6, generator=g)
    res = []
    for fused in (True, False):
        rng = DeviceRNG(seed=11, device='cpu')
        rng.begin_step()
        x = x0.clone().requires_grad_(True)
        w = torch.randn(x0.shape, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
        y = F.lrelu_dropout(x * w, 0.2, 0.5, rng) if fused else F.dropout(F.leaky_relu(x * w, 0.2), 0.5, rng=rng)
        (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
        pen = (gx ** 2).sum()                       # a function of the first gradient, differentiated w.r.t. w
        (gw,) = torch.autograd.grad(pen, w)
        res.append((y.detach(), gx.detach(), gw))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    assert 0.3 < (res[0][0] == 0).float().mean().item() < 0.7          # about half the values dropped


def _module_for_loop(net):
    if net == 'cifar':
        import ctgan_amd.gan_cifar as M
        M.configure(DIM=8, BATCH_SIZE=2)
    elif net == 'mnist':
        import ctgan_amd.gan_mnist as M
        M.configure(DIM=8, BATCH_SIZE=2)
    elif net == 'lsun128':
        import ctgan_amd.gan_lsun128 as M
        M.configure(BATCH_SIZE=2, DIM_G_64=4, DIM_G_32=4, DIM_G_16=4, DIM_G_8=8, DIM_G_4=8, DIM_D_64=4, DIM_D_32=4, DIM_D_16=8, DIM_D_8=8)
    else:
        import ctgan_amd.gan_64x64 as M
        M.configure(DIM=4, BATCH_SIZE=2)
    return M


def _build_lazy(M, dev):
    if hasattr(M, 'build_params'):
        M.build_params(dev)
    else:
        with torch.no_grad():
            M.Discriminator(M.Generator(2, noise=torch.zeros(2, 128, device=dev)), u=[torch.full((2,) + s, 0.9, device=dev) for s in M.feat_shapes()])


@pytest.mark.parametrize('net', ['cifar', 'mnist', 'lsun128', '64x64'])
def test_train_iteration_runs_for_every_module_of_the_shared_step(cpu_kernels, net):
    """DCGANTrainer.train_iteration - the loop body with the batched fake draw (BATCH_FAKES: `Generator(..., groups=CRITIC_ITERS)`) -
    for EVERY module the shared step serves (gan_64x64.Generator lacked the `groups` argument and the loop raised; only the
    single steps were tested).  Two iterations (the second one includes the generator step); the batched draw must equal per-step draws
    of the same Philox streams group by group (each group = one generator call with its own BatchNorm statistics)."""
    import ctgan_amd.tflib as lib
    from ctgan_amd import dcgan_step
    from ctgan_amd.dcgan_step import DCGANTrainer
    M = _module_for_loop(net)
    try:
        lib.set_seed(5)
        _build_lazy(M, 'cpu')
        assert dcgan_step.BATCH_FAKES
        tr = DCGANTrainer(M, seed=3)
        B, n = M.cfg.BATCH_SIZE, M.cfg.CRITIC_ITERS
        g = torch.Generator().manual_seed(1)
        if net == 'mnist':
            batch = torch.rand(B, M.cfg.OUTPUT_DIM, generator=g)
        else:
            batch = torch.randint(0, 256, (B, M.cfg.OUTPUT_DIM), generator=g, dtype=torch.int32)
        for it in range(2):
            out = tr.train_iteration(it, lambda: batch)
            assert torch.isfinite(out['cost']).item()
        # statistic groups: a batched forward over n groups equals n separate generator calls on the same noise rows
        z = torch.randn(n * B, 128, generator=g)
        with torch.no_grad():
            both = M.Generator(n * B, noise=z, groups=n)
            for k in range(n):
                one = M.Generator(B, noise=z[k * B:(k + 1) * B])
                _cmp(both[k * B:(k + 1) * B], one, 2e-5, '%s group %d' % (net, k), atol=1e-6)
    finally:
        M.configure()


def _dcgan_scheduled_vs_autograd(lib, which, dim, B, S, dev):
    """One critic step of a DCGAN script through dcgan_schedule.critic_step and through d_losses + autograd: same weights, inputs, Philox streams."""
    import ctgan_amd.dcgan_schedule as DS
    from ctgan_amd.dcgan_step import DCGANTrainer
    if which == 'cifar':
        import ctgan_amd.gan_cifar as M
    else:
        import ctgan_amd.gan_mnist as M
    res = {}
    for merged in (False, True):
        lib.delete_all_params(); lib.set_device(dev); lib.set_seed(1)
        M.configure(DIM=dim, BATCH_SIZE=B)
        try:
            d = lib._dev()
            with torch.no_grad():
                M.Discriminator(M.Generator(2, noise=torch.zeros(2, 128, device=d)), u=[torch.full((2,) + s, 0.9, device=d) for s in M.feat_shapes()])
            g = torch.Generator().manual_seed(5)
            with torch.no_grad():
                for n, p in lib._params.items():
                    if n.endswith(('.Biases', '.b')):
                        p.add_((0.1 * torch.randn(p.shape, generator=g)).to(p.device))
            tr = DCGANTrainer(M, seed=3)
            tr.loss_scale = S
            real = (torch.randint(0, 256, (B, 3072), dtype=torch.int32, generator=g) if which == 'cifar' else torch.rand(B, 784, generator=g)).to(d)
            old, DS.MERGED_BWD = DS.MERGED_BWD, merged
            try:
                fake = tr.generate_fakes(1)[0]
                assert DS.usable(tr, None, fake, real) == merged
                tr.rng.begin_step()
                out, grads = tr.d_grads(real, fake=fake)
            finally:
                DS.MERGED_BWD = old
            res[merged] = (out, grads, [n for n, _ in tr.d_named])
        finally:
            M.configure()
    return res[False], res[True]


@pytest.mark.parametrize('which,dim,S,mode', [('cifar', 32, 1.0, None), ('mnist', 32, 1024.0, None), ('mnist', 64, 1.0, None), ('cifar', 32, 1.0, 'bf16')])
def test_hand_scheduled_dcgan_critic_step_equals_the_autograd_form(cpu_kernels, which, dim, S, mode):
    """dcgan_schedule.critic_step (round 5: ONE forward and ONE backward chain over [real, fake, real | x_hat], weight gradients from the
    first 3B rows, the penalty's double backward on the x_hat rows only) against DCGANTrainer.d_losses + autograd: loss terms, slopes,
    dD/dx_hat and every parameter gradient - first conv through im2col + GEMM (CIFAR; MNIST at DIM 32) and on the direct few-channel
    kernels (MNIST at DIM 64), with and without a loss scale."""
    import ctgan_amd.kernels as K
    import ctgan_amd.tflib as lib
    # mode 'bf16' (the stand-ins still multiply in fp32): the 16-bit modes' per-filter policy applies - on a CPU tensor no filter is queued, every
    # use's weight gradient arrives at once and the schedule has to sum a filter's two uses itself (on the GPU: the large layers of config[1])
    with K.mma_dtype(mode):
        a, b = _dcgan_scheduled_vs_autograd(lib, which, dim, 2, S, 'cpu')
    for k in ('cost', 'wgan_only', 'ct', 'gp', 'slopes', 'gp_grads'):
        _cmp(b[0][k], a[0][k], 2e-6, 'scheduled.' + k, atol=1e-7)
    assert a[2] == b[2]
    for n, x, y in zip(a[2], a[1], b[1]):
        assert (x is None) == (y is None) and (x is None or x.shape == y.shape), n
        if x is not None:
            _cmp(y, x, 5e-6, 'scheduled grad ' + n, atol=1e-8 * S)
