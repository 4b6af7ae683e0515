"""The committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py) still
describe the oracle: guards the checker itself against drift.  CPU only."""
import numpy as np
import torch

from oracle import nets, steps, tf_ops
from oracle import tflib_ref as ops
from tests import golden_util as G

F64 = torch.float64
T = lambda a: torch.from_numpy(np.asarray(a)).to(F64)   # noqa: E731


def test_ops_fixture_reproduces():
    z = G.load('ops.npz')
    for i in range(6):
        N, C, H, W, K, k, s = [int(v) for v in z['conv%d_cfg' % i]]
        y = tf_ops.bias_add_nchw(tf_ops.conv2d_same(T(z['conv%d_x' % i]), T(z['conv%d_w' % i]), s), T(z['conv%d_b' % i]))
        np.testing.assert_allclose(y.numpy(), z['conv%d_y' % i], rtol=1e-12, atol=1e-12)
    for i in range(3):
        y = tf_ops.bias_add_nchw(tf_ops.conv2d_transpose_same(T(z['deconv%d_x' % i]), T(z['deconv%d_w' % i]), 2), T(z['deconv%d_b' % i]))
        np.testing.assert_allclose(y.numpy(), z['deconv%d_y' % i], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(tf_ops.dropout(T(z['ew_x']), 0.8, T(z['ew_u'])).numpy(), z['ew_drop08'], rtol=1e-12)
    np.testing.assert_allclose(steps.ct_term(T(z['ct_d']), T(z['ct_d_']), T(z['ct_f']), T(z['ct_f_']), 2.0, 0.5).numpy(), z['ct_M05'], rtol=1e-12)
    th = T(z['adam_theta'][0]); m = torch.zeros_like(th); v = torch.zeros_like(th)
    for t in range(1, 4):
        th, m, v = tf_ops.tf_adam_step(th, T(z['adam_g'][t - 1]), m, v, t, 2e-4 * (1 - t / 10.), 0.5, 0.9)
        np.testing.assert_allclose(th.numpy(), z['adam_theta'][t], rtol=1e-12)


def test_resnet_trace_reproduces():
    z = G.load('resnet_trace.npz')
    dim, B, iters = [int(v) for v in z['cfg']]
    reg = ops.Registry(dtype=F64)
    for n, w in G.trace_weights(z).items():
        tr = not n.endswith(('.moving_mean', '.moving_variance'))
        reg[n] = w.double().requires_grad_(tr)
        if not tr:
            reg.non_trainable.add(n)
    cfg = nets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    optD = steps.TFAdam(reg, [str(n) for n in z['d_names']], 0.0, 0.9)
    optG = steps.TFAdam(reg, [str(n) for n in z['g_names']], 0.0, 0.9)
    for it in range(iters):
        real, labels, rnd = G.trace_d_inputs(z, it, F64)
        o = steps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=it, B=B)
        for k in ('cost', 'ct', 'gp', 'acgan'):
            assert abs(o[k].item() - float(z['it%d.d.out.%s' % (it, k)])) < 1e-9
        o = steps.resnet_g_step(reg, cfg, optG, G.trace_g_inputs(z, it, F64), iteration=it + 1, B=B)
        assert abs(o['cost'].item() - float(z['it%d.g.out.cost' % it])) < 1e-9
