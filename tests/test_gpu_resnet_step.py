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


def _cmp_l2(a, b, tol, what, atol=1e-9):
    """relative L2 error.  Used for gradient tensors of the full-width nets: with ~1e8 ReLU inputs per
    step a handful sit within fp32 round-off of zero, and a flipped mask changes individual gradient
    entries by O(1) while leaving the tensor as a whole accurate to ~1e-4."""
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    err = (a - b).norm().item()
    scale = b.norm().item()
    assert err <= tol * scale + atol, '%s: L2 err %.3e vs norm %.3e' % (what, err, scale)


def _rel_l2(a, b):
    a = a.detach().cpu().double().reshape(-1); b = b.detach().cpu().double().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _twin_d_grads(reg, cfg, real, labels, rnd, B):
    reg32 = oref.Registry(dtype=torch.float32)
    for n, t in reg.items():
        reg32[n] = t.detach().float().requires_grad_(t.requires_grad)
    reg32.non_trainable = set(reg.non_trainable)
    rnd32 = {k: ([t.float() for t in v] if isinstance(v, list) else v.float()) for k, v in rnd.items()}
    out = osteps.resnet_d_losses(reg32, cfg, real, labels, rnd32, B=B)
    g = osteps.grads_of(out['cost'], reg32, 'Discriminator.')
    g['gp_grads'] = out['gp_grads']
    return g


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


def _teacher_force(lib, reg, opt, oopt):
    """Copy the oracle's weights and Adam slots into the product so the next comparison starts
    from identical state: isolates per-step error from chaotic drift."""
    lib.load_state_dict({n: t.detach().float() for n, t in reg.items()})
    opt.load_named_slots(oopt.m, oopt.v, oopt.t)


@pytest.mark.parametrize('dim,B,steps,streams', [(16, 8, 3, False), pytest.param(128, 64, 1, False, marks=pytest.mark.slow),
                                                 (16, 8, 3, True), (32, 8, 2, True), (128, 64, 2, True)])
def test_d_and_g_step_parity_teacher_forced(setup, dim, B, steps, streams):
    """streams=False: every random tensor is injected into both sides (`rnd=`), which switches the product to its op-by-op
    path.  streams=True: the product runs its DEFAULT path (rnd=None: in-kernel Philox dropout, shared trunk / tail forward,
    fused heads and input preparation, grouped weight gradients) and the oracle is fed the same Philox streams regenerated in
    numpy (oracle/philox.py) - the benchmarked step against the reference graph as written."""
    from oracle import philox
    R, lib = setup(dim, B)
    pos = [0]                    # stream position = number of steps executed (DeviceRNG.ctr)
    reg = _oracle_from_product(lib)
    cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    g = torch.Generator().manual_seed(11)
    tr = R.Trainer(seed=1)
    optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
    for it in range(steps):
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd = philox.rnd_resnet_d(tr.rng.seed, 0, pos[0], B, dim) if streams else osteps.make_rnd_resnet_d(B, dim, g)
        # fp32 twin of the oracle on the same inputs: how far ANY fp32 evaluation of this graph sits
        # from the fp64 truth (ReLU masks of pre-activations within round-off of zero flip)
        twin = _twin_d_grads(reg, cfg, real, labels, rnd, B)
        out = tr.d_step(real.cuda(), labels.cuda(), None if streams else {k: _to_dev(v) for k, v in rnd.items()}, iteration=it)
        pos[0] += 1
        assert tr.rng.ctr.item() == pos[0]
        ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=it, B=B)
        # north-star tolerance: losses within 1e-3 relative; measured fp32-vs-fp64 error is ~1e-5
        for k in ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only'):
            _cmp(out[k], ref[k], 2e-4, 'd_step[%d].%s' % (it, k), atol=1e-6)
        _cmp(out['acc_real'], ref['acc_real'], 0, 'acc_real', atol=1.01 / B)     # argmax ties only
        _cmp(out['fake'], ref['fake'], 1e-4, 'generator samples')
        tol = lambda key, floor: max(floor, 3.0 * _rel_l2(twin[key], ref['grads'][key] if key != 'gp_grads' else ref['gp_grads']))
        _cmp_l2(out['gp_grads'], ref['gp_grads'], tol('gp_grads', 1e-3), 'dD/dx_hat')
        for n in ref['grads']:
            _cmp_l2(out['grads'][n], ref['grads'][n], tol(n, 2e-3), 'dgrad ' + n, atol=1e-7)
        for n, _ in reg.trainable_with_name('Discriminator.'):
            if ref['grads'][n].abs().max() < 1e-12:
                # analytically zero gradient (the critic's output bias cancels in every loss term):
                # fp32 leaves O(1e-9) noise that Adam normalises into O(lr) steps; TF's fp32 does too
                continue
            # Adam's first steps are sign-like (|delta| = lr_t/sqrt(1-b2) = 2e-4 whatever |g| is): where g
            # is within fp32 noise of zero its SIGN is noise, so single entries may differ by two full
            # steps; the tensor as a whole must still agree
            _cmp(lib._params[n], reg[n], 1e-3, 'theta ' + n, atol=4.2e-4)
            # entries that moved differently (beyond rounding) must be ones whose gradient is at the fp32 noise
            # floor of its tensor (there the SIGN of g, hence the whole first Adam step, is noise), and few
            dth = (lib._params[n].detach().cpu().double() - reg[n].detach().double()).abs().reshape(-1)
            gref = ref['grads'][n].detach().double().abs().reshape(-1)
            bad = dth > 2e-5
            assert bad.sum().item() <= max(1, 0.02 * dth.numel()), 'theta %s: %d entries differ' % (n, bad.sum().item())
            if bad.any():
                assert gref[bad].max().item() <= 2e-3 * gref.max().item(), \
                    'theta %s: differing entry has |g| %.3e vs max %.3e' % (n, gref[bad].max().item(), gref.max().item())
            # L2 over the entries with a meaningful gradient sign - the Adam kernel itself is checked to 1e-6 in
            # test_gpu_kernels.py::test_tf_adam_kernel
            keep = ~bad
            _cmp_l2(lib._params[n].detach().cpu().double().reshape(-1)[keep], reg[n].detach().double().reshape(-1)[keep], 1e-3,
                    'theta(L2) ' + n, atol=0.02 * 2e-4 * reg[n].numel() ** 0.5)
        _teacher_force(lib, reg, tr.d_opt, optD)
        rg = philox.rnd_resnet_g(tr.rng.seed, 0, pos[0], B, dim) if streams else osteps.make_rnd_resnet_g(B, dim, g)
        out = tr.g_step(None if streams else {'z': _to_dev(rg['z']), 'label_u': _to_dev(rg['label_u']), 'u': _to_dev(rg['u'])},
                        iteration=it + 1)
        pos[0] += 1
        ref = osteps.resnet_g_step(reg, cfg, optG, rg, iteration=it + 1, B=B)
        _cmp(out['cost'], ref['cost'], 2e-4, 'g cost', atol=1e-6)
        _cmp(out['samples'], torch.cat(ref['samples']), 1e-4, 'g samples')
        for n in ref['grads']:
            _cmp_l2(out['grads'][n], ref['grads'][n], 5e-3, 'ggrad ' + n, atol=1e-7)
        _teacher_force(lib, reg, tr.g_opt, optG)


def test_free_running_losses_stay_within_north_star_tolerance(setup):
    """No teacher forcing: 3 iterations of (D step, G step) from identical initial weights and
    identical injected randomness; critic losses must stay within the north-star 1e-3 relative."""
    dim, B = 16, 8
    R, lib = setup(dim, B)
    reg = _oracle_from_product(lib)
    cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    g = torch.Generator().manual_seed(12)
    tr = R.Trainer(seed=1)
    optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
    for it in range(3):
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd = osteps.make_rnd_resnet_d(B, dim, g)
        out = tr.d_step(real.cuda(), labels.cuda(), {k: _to_dev(v) for k, v in rnd.items()}, iteration=it)
        ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=it, B=B)
        _cmp(out['cost'], ref['cost'], 1e-3, 'free-running d cost[%d]' % it)
        rg = osteps.make_rnd_resnet_g(B, dim, g)
        out = tr.g_step({'z': _to_dev(rg['z']), 'label_u': _to_dev(rg['label_u']), 'u': _to_dev(rg['u'])}, iteration=it + 1)
        ref = osteps.resnet_g_step(reg, cfg, optG, rg, iteration=it + 1, B=B)
        # the generator cost sees the critic's output bias, a null direction of the critic loss that
        # random-walks by O(lr) per step in any fp32 implementation: absolute allowance 3 steps * 2e-4
        _cmp(out['cost'], ref['cost'], 1e-3, 'free-running g cost[%d]' % it, atol=6e-4)


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


@pytest.mark.parametrize('dim', [16, 128])
def test_fixed_noise_sample_tensors_match_oracle(setup, dim):
    """north_star "sample tensors": generate_image's path (TF/CT_gan_cifar_resnet.py:341-348) - Generator(100, fixed_labels =
    [0..9] x 10, noise = fixed_noise[100,128]), conditional BN on the statistics of those 100 samples, pixels =
    ((s + 1) * 255 / 2).astype(int32) - against the fp64 oracle at reduced and FULL width.  Float samples: 1e-4 of the tanh range.
    Integer pixels: EQUAL, except where the oracle's real-valued pixel lies within 2e-2 of an integer boundary (1e-4 * 127.5 of
    float error + fp32 rounding of the affine map can move such a value across the truncation); at most 1 level apart there."""
    import numpy as np
    R, lib = setup(dim, 8)
    reg = _oracle_from_product(lib)
    cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(100, 128, generator=g, dtype=torch.float64)
    labels = torch.arange(10, dtype=torch.int32).repeat(10)
    tr = R.Trainer(seed=7)
    s, px = tr.generate_samples(noise.float().cuda(), labels.cuda())
    with torch.no_grad():
        ref = onets.resnet_generator(reg, cfg, 100, labels, noise)
    assert s.shape == (100, 3072) and px.dtype == torch.int32 and px.shape == (100, 3072)
    _cmp(s, ref, 1e-4, 'fixed-noise samples', atol=1e-6)
    real_px = ((ref + 1.) * (255. / 2)).numpy()
    ref_px = real_px.astype(np.int32)                            # numpy's float -> int32 cast truncates, like tf.cast / astype('int32')
    got = px.cpu().numpy()
    assert got.min() >= 0 and got.max() <= 255
    near_boundary = np.abs(real_px - np.round(real_px)) < 2e-2
    diff = got != ref_px
    assert not (diff & ~near_boundary).any(), 'integer pixels differ away from a truncation boundary: %d' % int((diff & ~near_boundary).sum())
    assert np.abs(got - ref_px).max() <= 1
    assert diff.mean() < 2e-3, 'fraction of boundary flips %.2e' % diff.mean()


@pytest.mark.parametrize('dim,B', [(32, 6), (128, 16)])
def test_dropout_fused_into_conv_epilogues_equals_separate_dropout_kernels(setup, dim, B):
    """DiscriminatorTail with the dropout masks (and the final ReLU) inside the conv kernels - forward in the producing
    conv's epilogue, backward in the consuming conv's dgrad epilogue - against the same tail with stand-alone dropout
    kernels, on identical Philox streams: outputs, parameter gradients and the gradient-penalty double backward."""
    import ctgan_amd.functional as F
    R, lib = setup(dim, B)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, 3072, generator=g).mul_(0.5).cuda()
    lab = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32).cuda()
    tr = R.Trainer(seed=9)
    params = tr.d_params
    res = {}
    for mode in (False, True):
        R.DROP_FUSION = mode
        try:
            tr.rng.begin_step()
            xi = x.clone().requires_grad_(True)
            d, f, a = R.Discriminator(xi, lab, 0.8, 0.5, 0.5, rng=tr.rng)
            (gx,) = torch.autograd.grad(d, xi, torch.ones_like(d), create_graph=True)
            gp, _ = F.gradient_penalty(gx, 10.0)
            cost = d.mean() + (f * f).mean() + a.mean() + gp
            grads = torch.autograd.grad(cost, params, allow_unused=True)
            res[mode] = (d.detach(), f.detach(), a.detach(), gx.detach(), [None if t is None else t.detach() for t in grads])
        finally:
            R.DROP_FUSION = True
    for i in range(4):
        assert _rel_l2(res[True][i], res[False][i]) < 1e-6, i
    for (n, _), a_, b_ in zip(tr.d_named, res[True][4], res[False][4]):
        assert (a_ is None) == (b_ is None), n
        if a_ is not None and b_.abs().max() > 0:
            assert _rel_l2(a_, b_) < 2e-5, n


@pytest.mark.parametrize('dim,B', [(32, 6), (128, 64)])
def test_fused_critic_heads_equal_separate_head_kernels(setup, dim, B):
    """HEAD_FUSION: mean + both Linear heads + loss heads (F.critic_tail_heads) and the gradient-penalty branch started at
    dD/dz of the last block (F.gp_head_grad) against the op-by-op head on identical Philox streams: every loss term, the
    critic outputs, dD/dx_hat and all parameter gradients of a critic step."""
    import ctgan_amd.functional as F
    import ctgan_amd.kernels as K
    R, lib = setup(dim, B)
    g = torch.Generator().manual_seed(33)
    real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32).cuda()
    lab = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32).cuda()
    tr = R.Trainer(seed=11)
    res = {}
    # The two forms batch the critic's rows differently (384-row shared tail vs separate passes), so with the split-mode routing on
    # they would not run the same conv kernels (kernels.X3_HYBRID routes by launch size); this equivalence is checked at 2e-5 / 5e-5 on
    # ONE conv family and on the fake batch that family generates (the tolerance leaves no room for a ReLU mask that flips between
    # the forms; whether one does depends on the inputs).  The routed path itself is checked against the oracle by the step / loop
    # parity tests.
    hybrid, K.X3_HYBRID = K.X3_HYBRID, False
    try:
        fake = tr.generate_fakes(lab)[0]
        for mode in (False, True):
            R.HEAD_FUSION = R.PREP_FUSION = R.TRUNK_SHARE = R.TAIL_SHARE = mode            # also: input preparation / concat+dropout in single launches
            try:
                tr.rng.begin_step()
                out = tr.d_losses(real, lab, fake=fake)
                with F.deferred_wgrads():
                    grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
                res[mode] = ({k: out[k].detach().clone() for k in ('cost', 'wgan', 'ct', 'acgan', 'gp', 'd_real', 'd_fake', 'gp_grads',
                                                                   'acc_real', 'acc_fake')},
                             [None if t is None else t.detach().clone() for t in grads])
            finally:
                R.HEAD_FUSION = R.PREP_FUSION = R.TRUNK_SHARE = R.TAIL_SHARE = True
    finally:
        K.X3_HYBRID = hybrid
    for k, v in res[True][0].items():
        assert _rel_l2(v, res[False][0][k]) < 2e-5, k
    gmax = max(float(b_.abs().max()) for b_ in res[False][1] if b_ is not None)
    for (n, _), a_, b_ in zip(tr.d_named, res[True][1], res[False][1]):
        assert (a_ is None) == (b_ is None), n
        if a_ is not None:       # analytically-zero gradients (the wgan head's bias: the loss sees differences of D only) are fp32 noise
            assert float((a_ - b_).norm()) <= 5e-5 * float(b_.norm()) + 1e-6 * gmax, n


@pytest.mark.parametrize('dim,B', [(32, 6), (128, 64)])
def test_fused_generator_heads_equal_separate_head_kernels(setup, dim, B):
    """HEAD_FUSION in the generator step (F.gen_tail_heads) against the op-by-op head on identical Philox streams: the cost and
    every generator parameter gradient."""
    import ctgan_amd.functional as F
    R, lib = setup(dim, B)
    tr = R.Trainer(seed=13)
    res = {}
    for mode in (False, True):
        R.HEAD_FUSION = mode
        try:
            tr.rng.begin_step()
            out = tr.g_losses()
            with F.deferred_wgrads():
                grads = torch.autograd.grad(out['cost'], tr.g_params, allow_unused=True)
            res[mode] = (out['cost'].detach().clone(), out['samples'].detach().clone(), [None if t is None else t.detach().clone() for t in grads])
        finally:
            R.HEAD_FUSION = True
    assert _rel_l2(res[True][0], res[False][0]) < 1e-5 and torch.equal(res[True][1], res[False][1])
    gmax = max(float(b_.abs().max()) for b_ in res[False][2] if b_ is not None)
    for (n, _), a_, b_ in zip(tr.g_named, res[True][2], res[False][2]):
        assert (a_ is None) == (b_ is None), n
        if a_ is not None:
            assert float((a_ - b_).norm()) <= 5e-5 * float(b_.norm()) + 1e-6 * gmax, n


def test_layernorm_critic_d_step_on_gpu(setup):
    """NORMALIZATION_D=True: the gradient penalty differentiates Layernorm twice (functional.layer_norm primitives)."""
    import ctgan_amd.gan_cifar_resnet as R0
    import ctgan_amd.tflib as lib0
    dim, B = 32, 4
    lib0.delete_all_params(); lib0.set_seed(5)
    R0.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B, NORMALIZATION_D=True)
    try:
        R0.build_params()
        reg = _oracle_from_product(lib0)
        cfg = onets.ResnetCfg(DIM_G=dim, DIM_D=dim, NORMALIZATION_D=True)
        g = torch.Generator().manual_seed(2)
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd = osteps.make_rnd_resnet_d(B, dim, g)
        tr = R0.Trainer(seed=1)
        optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
        out = tr.d_step(real.cuda(), labels.cuda(), {k: _to_dev(v) for k, v in rnd.items()}, iteration=0)
        ref = osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=0, B=B)
        for k in ('cost', 'wgan', 'acgan', 'ct', 'gp'):
            _cmp(out[k], ref[k], 5e-4, 'd_step.%s' % k, atol=1e-6)
        _cmp_l2(out['gp_grads'], ref['gp_grads'], 2e-3, 'dD/dx_hat')
        for n in ref['grads']:
            _cmp_l2(out['grads'][n], ref['grads'][n], 5e-3, 'dgrad ' + n, atol=1e-7)
    finally:
        lib0.delete_all_params(); R0.configure()


@pytest.mark.parametrize('dim,B,ac,mode', [(64, 8, True, None), (128, 64, True, None), (128, 64, False, None), (64, 8, False, 'f32x3'),
                                           (64, 8, False, 'bf16'), (128, 16, False, 'bf16')])
def test_hand_scheduled_critic_step_equals_the_autograd_path_on_gpu(setup, dim, B, ac, mode):
    """critic_schedule.critic_step (one backward chain over the rows of the dropout passes and of the gradient-penalty
    pass; weight gradients restricted to the dropout-pass rows; the penalty's double backward on the x_hat rows only) against the path it
    replaces (Trainer.d_losses + two autograd calls), at the benchmarked size: same weights, inputs and Philox streams.  The merged
    launches run other tile shapes over other row counts than the separate chains (another summation order inside fp32), and ReLU
    masks are taken from the same forward tensors, so the gradients agree to fp32 rounding - not bit for bit."""
    import ctgan_amd.functional as F
    from tests.test_host_logic_resnet import _scheduled_vs_autograd
    import ctgan_amd.kernels as K
    R, lib = setup(dim, B)
    # modes 'f32x3' / 'bf16' (ADVICE r5, high): filters whose uses are launched at once instead of queued - the schedule (and the autograd
    # path's FilterSpreadFn) must fold each finished spread-filter gradient itself.  bf16: the two paths round different partial sums to
    # 16 bits (other row counts per launch), so they agree to bf16 rounding of the operands, not to fp32 rounding
    with K.mma_dtype(mode):
        a, b = _scheduled_vs_autograd(R, lib, F, B, dim, ac, None)
    t = 2e-2 if mode == 'bf16' else 0.0
    for k in ('cost', 'wgan', 'acgan', 'wgan_only', 'ct', 'gp', 'acc_real', 'acc_fake', 'd_real', 'd_fake', 'real'):
        assert (a[0].get(k) is None) == (b[0].get(k) is None), k
        if a[0].get(k) is not None:
            _cmp(b[0][k], a[0][k], max(1e-5, t * 1e-2), 'scheduled.' + k, atol=1e-6)
    assert _rel_l2(b[0]['slopes'], a[0]['slopes']) < max(1e-5, t * 1e-2) and _rel_l2(b[0]['gp_grads'], a[0]['gp_grads']) < max(1e-5, t)
    assert a[2] == b[2]
    for n, x, y in zip(a[2], a[1], b[1]):
        assert (x is None) == (y is None), n
        if x is not None and x.abs().max() > 0:
            assert _rel_l2(y, x) < max(2e-5, t), (n, _rel_l2(y, x))


def test_hand_scheduled_critic_step_with_the_switched_off_overlap_forms_on_gpu(setup, monkeypatch):
    """The structural attempts of round 6 that stay behind switches (all measured level or slower): the dropout-pass rows' weight
    gradients on a side stream under the penalty's double backward (functional.flush_async; another split of the same sums: fp32 rounding),
    and - through the graphed engine - the filter images rebuilt on a side stream (functional.prepare_filters_async; the same launches in
    another order: bit-identical step outputs and weights)."""
    import ctgan_amd.functional as F
    from tests.test_host_logic_resnet import _scheduled_vs_autograd
    R, lib = setup(128, 64)
    monkeypatch.setattr(F, 'WGRAD_OVERLAP', True)
    a, b = _scheduled_vs_autograd(R, lib, F, 64, 128, True, None)
    for n, x, y in zip(a[2], a[1], b[1]):
        assert (x is None) == (y is None), n
        if x is not None and x.abs().max() > 0:
            assert _rel_l2(y, x) < 2e-5, (n, _rel_l2(y, x))
    monkeypatch.setattr(F, 'WGRAD_OVERLAP', False)
    # ... and blocks 3-4 of the backward chain / of the penalty's double backward as ONE launch each (kernels.conv_chain8x8, csrc/chain8x8.hip;
    # CTGAN_CHAIN8X8=1): the same sums in another order
    import ctgan_amd.kernels as K
    monkeypatch.setattr(K, 'CHAIN8X8', True)
    seen = []
    orig = K.conv_chain8x8
    monkeypatch.setattr(K, 'conv_chain8x8', lambda *a, **k: (seen.append(len(a[1])), orig(*a, **k))[1])
    a, b = _scheduled_vs_autograd(R, lib, F, 64, 128, True, None)
    assert seen == [5, 5], seen
    for k in ('cost', 'gp', 'ct'):
        _cmp(b[0][k], a[0][k], 1e-5, 'scheduled (chains).' + k, atol=1e-6)
    for n, x, y in zip(a[2], a[1], b[1]):
        assert (x is None) == (y is None), n
        if x is not None and x.abs().max() > 0:
            assert _rel_l2(y, x) < 2e-5, (n, _rel_l2(y, x))
    monkeypatch.setattr(K, 'CHAIN8X8', False)
    monkeypatch.setattr(K, 'conv_chain8x8', orig)
    from ctgan_amd.engine import GraphedTrainer
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(F, 'PREP_ASYNC', mode)
        lib.delete_all_params(); lib.set_seed(4)
        R.configure(DIM_G=64, DIM_D=64, BATCH_SIZE=8)
        R.build_params()
        tr = R.Trainer(seed=11)
        eng = GraphedTrainer(tr, use_graphs=True)
        assert eng.graphed, eng.graph_error
        g = torch.Generator().manual_seed(2)
        batches = [(torch.randint(0, 256, (8, 3072), dtype=torch.int32, generator=g).cuda(), torch.randint(0, 10, (8,), dtype=torch.int32, generator=g).cuda()) for _ in range(5)]
        it = iter(batches * 3)
        outs = [eng.train_iteration(k, lambda: next(it))['cost'].item() for k in range(1, 4)]
        torch.cuda.synchronize()
        res[mode] = (outs, tr.d_opt.theta.clone(), tr.g_opt.theta.clone())
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2])


@pytest.mark.parametrize('split_mode', [True, False])
def test_data_gradient_with_per_range_dropout_masks_equals_one_dropout_per_range(split_mode):
    """ctgan_conv2d16_dgrad_ex / ctgan_conv2d_dgrad_ex with sample ranges (round 5: the merged backward carries the rows of the dropout
    passes and of the penalty pass in one data gradient, each range with the mask of its own forward dropout): the epilogue form
    against the plain data gradient followed by ctgan_dropout_rng on each range's own rows - same Philox draws, bit for bit."""
    import ctgan_amd.kernels as K
    from ctgan_amd.kernels import ConvGeom
    old = K.X3_HYBRID
    K.X3_HYBRID = split_mode
    try:
        g = torch.Generator().manual_seed(4)
        N, C = 256, 128
        geom = ConvGeom(C, 8, 8, C, 3, 3, 1)
        cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)       # noqa: E731
        gy = cl(torch.randn(N, C, 8, 8, generator=g))
        mask = cl(torch.randn(N, C, 8, 8, generator=g))
        resid = cl(torch.randn(N, C, 8, 8, generator=g))
        w = (torch.randn(3, 3, C, C, generator=g) * 0.05).cuda()
        ctr = torch.full((1,), 3, dtype=torch.int64, device='cuda')
        drop = {'ranges': [(192, (0.5, 77, 5, ctr)), (256, (0.8, 77, 9, ctr))]}
        got = K.conv_dgrad(gy, w, geom, N, mask=mask, resid=resid, drop=drop)
        name = K.last_kernel()
        assert ('conv16x3' in name) == split_mode, name
        ref = K.conv_dgrad(gy, w, geom, N, mask=mask, resid=resid)
        ref = K._dropout_ranges(ref, drop)
        assert torch.equal(got, ref)
        # the ranges really differ: range 2 keeps ~80 %, range 1 ~50 %
        z = (got == 0).float()
        assert 0.4 < z[:192].mean().item() < 0.6 and 0.1 < z[192:].mean().item() < 0.3
    finally:
        K.X3_HYBRID = old


def test_generator_output_stage_with_batchnorm_on_load_equals_the_separate_launches():
    """Under no_grad the output stage tanh(Conv2D(relu(Batchnorm(h)))) of TF/CT_gan_cifar_resnet.py:164-166 runs as the moments + ONE launch of the
    many -> few pixel kernel (batch norm while the input is staged, tanh in the epilogue; ctgan_epilogue_ext.in_bn_*): the same arithmetic as
    ctgan_bn_apply + the conv + ctgan_tanh_fwd - compared bit for bit (and to 1e-6 should the compiler contract the two kernels differently),
    for one tower and for the five towers of an iteration's fake batches."""
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.kernels as K
    import ctgan_amd.tflib as lib
    lib.delete_all_params(); lib.set_device(None); lib.set_seed(3)
    R.configure(DIM_G=128, DIM_D=128, BATCH_SIZE=64)
    K.debug_x3_hk(0)          # (bit for bit: both forms on the pixel-tiled halo kernel - without the norm on load the 64-row 8x8 conv would ride conv16x3hk_kernel, another summation order)
    try:
        dev = lib._dev()
        g = torch.Generator().manual_seed(4)
        for n, groups in ((64, 1), (320, 5)):
            z = torch.randn(n, 128, generator=g).to(dev)
            labels = torch.randint(0, 10, (n,), generator=g, dtype=torch.int32).to(dev)
            with torch.no_grad():
                R.Generator(n, labels, noise=z, groups=groups)                 # creates the parameters
                with torch.no_grad():
                    for nm in ('Generator.OutputN.scale', 'Generator.OutputN.offset', 'Generator.Output.Biases'):
                        p = lib.param(nm); p.add_((0.2 * torch.randn(p.shape, generator=g)).to(dev))
                lib.bump_epoch()
                assert R.OUTPUT_STAGE_FUSION
                a = R.Generator(n, labels, noise=z, groups=groups); ka = K.last_kernel()
                R.OUTPUT_STAGE_FUSION = False
                try:
                    b = R.Generator(n, labels, noise=z, groups=groups)
                finally:
                    R.OUTPUT_STAGE_FUSION = True
            assert ka == 'fewch_m2f(bn)', ka
            assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-6
            assert torch.equal(a, b), float((a - b).abs().max())
    finally:
        K.debug_x3_hk(1)
        R.configure(); lib.delete_all_params()
