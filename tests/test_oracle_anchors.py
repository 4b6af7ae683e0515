"""Known-answer anchors that pin the CPU oracle.

The reference ships no tests or golden vectors ("parity unpinned"); these are the self-checks its
own source implies.
"""
import numpy as np
import pytest
import torch

from oracle import nets, np_conv, steps, tf_ops
from oracle import tflib_ref as ops

torch.manual_seed(0)
F64 = torch.float64


# ---- TF SAME padding rule, the cases SURVEY 8(a) row A1 enumerates
@pytest.mark.parametrize("n,k,s,exp", [
    (32, 3, 1, (32, 1, 1)), (32, 1, 1, (32, 0, 0)), (32, 5, 2, (16, 1, 2)), (28, 5, 2, (14, 1, 2)),
    (7, 5, 2, (4, 2, 2)), (16, 3, 2, (8, 0, 1)), (8, 5, 1, (8, 2, 2)), (14, 5, 2, (7, 1, 2)),
])
def test_same_pads(n, k, s, exp):
    assert tf_ops.same_pads(n, k, s) == exp


# ---- torch formulation == independent numpy tap-loop restatement
@pytest.mark.parametrize("H,k,s,ci,co", [(8, 3, 1, 3, 4), (8, 5, 2, 2, 3), (7, 5, 2, 3, 2), (6, 1, 1, 4, 2),
                                          (9, 3, 2, 2, 2), (4, 5, 1, 1, 2)])
def test_conv_same_vs_numpy(H, k, s, ci, co):
    x = torch.randn(2, ci, H, H + 1, dtype=F64)
    w = torch.randn(k, k, ci, co, dtype=F64)
    a = tf_ops.conv2d_same(x, w, s).numpy()
    b = np_conv.conv2d_same_np(x.numpy(), w.numpy(), s)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("H,k,ci,co", [(4, 5, 3, 2), (7, 5, 2, 3), (8, 5, 1, 1), (3, 3, 2, 2), (5, 1, 2, 2)])
def test_conv_transpose_same_vs_numpy(H, k, ci, co):
    x = torch.randn(2, ci, H, H + 1, dtype=F64)
    w = torch.randn(k, k, co, ci, dtype=F64)
    a = tf_ops.conv2d_transpose_same(x, w, 2).numpy()
    b = np_conv.conv2d_transpose_same_np(x.numpy(), w.numpy(), 2)
    assert a.shape == (2, co, 2 * H, 2 * (H + 1))
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


def test_conv_transpose_is_adjoint_of_conv():
    """tf.nn.conv2d_transpose is defined as the gradient of conv2d: <conv(u),v> == <u,convT(v)>."""
    u = torch.randn(2, 3, 16, 16, dtype=F64)        # large side, channels = deconv output_dim
    v = torch.randn(2, 4, 8, 8, dtype=F64)          # small side, channels = deconv input_dim
    w = torch.randn(5, 5, 3, 4, dtype=F64)          # [k,k,out,in] of the deconv == HWIO of the conv
    lhs = (tf_ops.conv2d_same(u, w, 2) * v).sum()
    rhs = (u * tf_ops.conv2d_transpose_same(v, w, 2)).sum()
    assert abs(lhs - rhs) < 1e-9 * abs(lhs)


def test_pytorch_padding_shortcut_is_wrong():
    """conv_transpose2d(padding=2, output_padding=1) is shifted by a pixel."""
    v = torch.randn(1, 2, 4, 4, dtype=F64)
    w = torch.randn(5, 5, 3, 2, dtype=F64)
    ours = tf_ops.conv2d_transpose_same(v, w, 2)
    naive = torch.nn.functional.conv_transpose2d(v, w.permute(3, 2, 0, 1), stride=2, padding=2, output_padding=1)
    assert (ours - naive).abs().max() > 1e-3


# ---- algebraic identities (anchor 4)
def test_upsample_is_nearest_and_pool_is_avgpool():
    x = torch.randn(2, 3, 4, 5, dtype=F64)
    up = tf_ops.upsample2(x)
    assert torch.equal(up, x.repeat_interleave(2, 2).repeat_interleave(2, 3))
    y = torch.randn(2, 3, 6, 8, dtype=F64)
    assert torch.allclose(tf_ops.mean_pool2(y), torch.nn.functional.avg_pool2d(y, 2), atol=1e-15)


def test_dropout_formula():
    x = torch.randn(4, 5, dtype=F64)
    u = torch.rand(4, 5, dtype=F64)
    y = tf_ops.dropout(x, 0.8, u)
    keep = (u >= 0.2 - 1e-16)
    assert torch.allclose(y, torch.where(u + 0.8 >= 1.0, x / 0.8, torch.zeros_like(x)))
    assert tf_ops.dropout(x, 1.0, None) is x
    assert keep.any()


def test_ct_zero_when_no_dropout_and_clamp():
    d = torch.randn(8, dtype=F64); f = torch.randn(8, 16, dtype=F64)
    assert steps.ct_term(d, d, f, f) == 0
    d2 = d + 0.1; f2 = f + 0.2
    ct0 = steps.ct_term(d, d2, f, f2, 2.0, 0.0)
    exp = (2.0 * 0.01 + 0.2 * 0.04)
    assert abs(ct0 - exp) < 1e-12
    assert steps.ct_term(d, d2, f, f2, 2.0, 1.0) == 0           # M above CT clamps to zero
    assert abs(steps.ct_term(d, d2, f, f2, 2.0, 0.01) - (exp - 0.01)) < 1e-12


def test_tf_adam_form():
    th = torch.tensor([1.0], dtype=F64); g = torch.tensor([0.5], dtype=F64)
    m = torch.zeros(1, dtype=F64); v = torch.zeros(1, dtype=F64)
    th1, m1, v1 = tf_ops.tf_adam_step(th, g, m, v, 1, 1e-3, 0.5, 0.9)
    # t=1: m=0.25, v=0.025, lr_t = 1e-3*sqrt(0.1)/0.5
    exp = 1.0 - (1e-3 * np.sqrt(0.1) / 0.5) * 0.25 / (np.sqrt(0.025) + 1e-8)
    assert abs(th1.item() - exp) < 1e-15
    # differs from torch.optim.Adam's epsilon placement only at O(eps)
    assert abs(th1.item() - (1.0 - 1e-3)) < 1e-6


# ---- shape closure (anchor 1) + parameter counts (anchor 2) + init statistics (anchor 3)
def test_resnet_shapes_and_param_counts():
    reg = ops.Registry(dtype=torch.float32)
    cfg = nets.ResnetCfg()
    lab = torch.arange(2, dtype=torch.int32)
    x = nets.resnet_generator(reg, cfg, 2, lab, torch.randn(2, 128))
    assert x.shape == (2, 3072) and x.abs().max() <= 1
    d, f, a = nets.resnet_discriminator(reg, cfg, x, lab, 1., 1., 1.)
    assert d.shape == (2,) and f.shape == (2, 128) and a.shape == (2, 10)
    nD = sum(p.numel() for n, p in reg.trainable_with_name('Discriminator.'))
    nG = sum(p.numel() for n, p in reg.trainable_with_name('Generator'))
    assert nD == 1055115
    assert nG == 1202691 + 15360 + 256 == 1218307
    names = dict(reg.params_with_name('Generator'))
    assert 'Generator.OutputN.moving_mean' in names and 'Generator.OutputN.moving_variance' in names
    assert reg['Generator.1.N1.scale'].shape == (10, 128)
    assert reg['Discriminator.1.Conv1.Filters'].shape == (3, 3, 3, 128)       # HWIO
    assert reg['Generator.Input.W'].shape == (128, 2048)
    # sharing: a second call creates nothing new
    n_before = len(reg)
    nets.resnet_discriminator(reg, cfg, x, lab, 1., 1., 1.)
    assert len(reg) == n_before


def test_init_statistics():
    reg = ops.Registry(dtype=torch.float32)
    x = torch.zeros(1, 128, 8, 8)
    ops.Conv2D(reg, 'a', 128, 128, 3, x)                    # he_init: sigma = sqrt(4/(fi+fo))
    w = reg['a.Filters']
    sigma = np.sqrt(4. / (128 * 9 + 128 * 9))
    assert abs(w.std().item() - sigma) < 0.02 * sigma
    assert w.abs().max().item() <= sigma * np.sqrt(3) + 1e-7
    ops.Conv2D(reg, 'b', 128, 128, 1, x, he_init=False)
    sigma = np.sqrt(2. / (128 + 128))
    assert abs(reg['b.Filters'].std().item() - sigma) < 0.03 * sigma
    ops.Conv2D(reg, 'c', 128, 256, 5, x, stride=2)          # fan_out / stride^2
    sigma = np.sqrt(4. / (128 * 25 + 256 * 25 / 4))
    assert abs(reg['c.Filters'].std().item() - sigma) < 0.02 * sigma
    ops.Deconv2D(reg, 'd', 256, 128, 5, x[:, :, :4, :4].repeat(1, 2, 1, 1))
    sigma = np.sqrt(4. / (256 * 25 / 4 + 128 * 25))
    assert reg['d.Filters'].shape == (5, 5, 128, 256)
    assert abs(reg['d.Filters'].std().item() - sigma) < 0.02 * sigma
    ops.Linear(reg, 'e', 128, 2048, torch.zeros(1, 128))    # None -> glorot
    sigma = np.sqrt(2. / (128 + 2048))
    assert abs(reg['e.W'].std().item() - sigma) < 0.02 * sigma
    assert reg['a.Biases'].abs().sum() == 0 and reg['e.b'].abs().sum() == 0


def test_dcgan_shape_closure():
    reg = ops.Registry(dtype=torch.float32)
    x = nets.cifar_generator(reg, 2, torch.randn(2, 128), DIM=16)
    assert x.shape == (2, 3072)
    u = [torch.rand(2, 16, 16, 16), torch.rand(2, 32, 8, 8), torch.rand(2, 64, 4, 4)]
    d, f = nets.cifar_discriminator(reg, x, u, DIM=16)
    assert d.shape == (2,) and f.shape == (2, 4 * 4 * 4 * 16)
    assert reg['Generator.BN1.scale'].shape == (1, 4 * 4 * 4 * 16)           # non-fused axes=[0] path
    reg2 = ops.Registry(dtype=torch.float32)
    y = nets.mnist_generator(reg2, 2, torch.randn(2, 128), DIM=8)
    assert y.shape == (2, 784) and y.min() >= 0 and y.max() <= 1
    u = [torch.rand(2, 8, 14, 14), torch.rand(2, 16, 7, 7), torch.rand(2, 32, 4, 4)]
    d, f = nets.mnist_discriminator(reg2, y, u, DIM=8)
    assert d.shape == (2,) and f.shape == (2, 4 * 4 * 4 * 8)


def test_resnet_d_and_g_step_run_small():
    """End-to-end oracle D/G step at reduced width: finite losses, CT>0 with dropout, grads for all."""
    reg = ops.Registry(dtype=F64, seed=3)
    cfg = nets.ResnetCfg(DIM_G=8, DIM_D=8)
    B = 4
    g = torch.Generator().manual_seed(1)
    real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
    rnd = steps.make_rnd_resnet_d(B, 8, g)
    out = steps.resnet_d_losses(reg, cfg, real, labels, rnd, B=B)
    assert torch.isfinite(out['cost']) and out['ct'] > 0 and out['gp'] > 0
    optD = steps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    before = reg['Discriminator.2.Conv1.Filters'].clone()
    o2 = steps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=0, B=B)
    assert set(o2['grads']) == set(optD.names)
    assert not torch.equal(before, reg['Discriminator.2.Conv1.Filters'])
    optG = steps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
    rg = steps.make_rnd_resnet_g(B, 8, g)
    o3 = steps.resnet_g_step(reg, cfg, optG, rg, iteration=1, B=B)
    assert torch.isfinite(o3['cost']) and set(o3['grads']) == set(optG.names)
    # second D step decreases nothing pathological and Adam t advanced
    assert optD.t == 1 and optG.t == 1


# ---- second restatement of the ResNet critic + its loss scalars (oracle/np_critic.py: numpy only, hand-derived gradient) ----------
def _np_critic_setup(dim=4, B=2, seed=3):
    from oracle import np_critic
    cfg = nets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    reg = ops.Registry(dtype=torch.float64, seed=seed)
    g = torch.Generator().manual_seed(seed)
    real_int = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
    rnd = steps.make_rnd_resnet_d(B, dim, g)
    ref = steps.resnet_d_losses(reg, cfg, real_int, labels, rnd, B=B)           # creates the parameters (reference init)
    with torch.no_grad():                                                        # biases are zero at init: make them count
        for n, p in reg.items():
            if n.endswith('.Biases') or n.endswith('.b'):
                p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * 0.1)
    ref = steps.resnet_d_losses(reg, cfg, real_int, labels, rnd, B=B)
    P = {n: p.detach().numpy() for n, p in reg.items()}
    nrnd = {k: ([t.numpy() for t in v] if isinstance(v, list) else v.numpy()) for k, v in rnd.items()}
    return np_critic, cfg, reg, P, ref, nrnd, B


def test_numpy_restatement_of_the_critic_agrees_with_the_torch_oracle():
    """Two restatements written separately from the reference text (torch ops + autograd vs numpy tap loops + a hand-derived
    backward) must produce the same critic outputs, the same dD/dx_hat and the same WGAN / CT / GP scalars (fp64: 1e-10)."""
    np_critic, cfg, reg, P, ref, rnd, B = _np_critic_setup()
    out = np_critic.critic_scalars(P, ref['real'].detach().numpy(), ref['fake'].detach().numpy(), rnd, B)
    for k in ('wgan_only', 'ct', 'gp'):
        a, b = out['wgan' if k == 'wgan_only' else k], ref[k].item()
        assert abs(a - b) <= 1e-10 * max(1.0, abs(b)), (k, a, b)
    np.testing.assert_allclose(out['d_real'], ref['d_real'].detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out['d_fake'], ref['d_fake'].detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out['slopes'], ref['slopes'].detach().numpy(), rtol=1e-9)
    np.testing.assert_allclose(out['gp_grads'], ref['gp_grads'].detach().numpy(), rtol=1e-8, atol=1e-12)
    assert out['ct'] > 0 and out['gp'] > 0            # the draws exercise both terms


def test_hand_derived_critic_gradient_matches_central_differences():
    """The numpy backward is pinned by the numpy forward alone: directional derivatives along random directions (ReLU / dropout
    kinks are measure-zero; h = 1e-6 in fp64)."""
    np_critic, cfg, reg, P, ref, rnd, B = _np_critic_setup(seed=5)
    x = (ref['real'] + 0.3 * (ref['fake'] - ref['real'])).detach().numpy()
    kps = (0.8, 0.5, 0.5)
    D, _, gx = np_critic.critic(P, x, kps, rnd['u_gp'], want_grad=True)
    rs = np.random.default_rng(0)
    for _ in range(4):
        v = rs.standard_normal(x.shape)
        h = 1e-6
        fd = (np_critic.critic(P, x + h * v, kps, rnd['u_gp'])[0] - np_critic.critic(P, x - h * v, kps, rnd['u_gp'])[0]) / (2 * h)
        an = (gx * v).sum(axis=1)
        np.testing.assert_allclose(an, fd, rtol=2e-5, atol=1e-9)


def test_numpy_critic_clean_pass_and_consistency_anchor():
    """No dropout (keep 1): both passes coincide, so the consistency term of identical passes is exactly 0 in the second
    restatement as well, and D of the numpy critic equals the torch oracle's clean pass."""
    np_critic, cfg, reg, P, ref, rnd, B = _np_critic_setup(seed=9)
    rf = torch.cat([ref['real'], ref['fake']], 0).detach()
    with torch.no_grad():
        dc, fc, _ = nets.resnet_discriminator(reg, cfg, rf, None, 1.0, 1.0, 1.0, None)
    d, f = np_critic.critic(P, rf.numpy(), (1.0, 1.0, 1.0), None)
    np.testing.assert_allclose(d, dc.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(f, fc.numpy(), rtol=1e-10, atol=1e-12)
