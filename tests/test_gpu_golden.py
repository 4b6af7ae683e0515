"""HIP path against the committed golden fixtures (no oracle import: pure data)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import golden_util as G  # noqa: E402


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def cl(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).float().cuda()
    out = torch.empty((t.shape[0], t.shape[2], t.shape[3], t.shape[1]), device='cuda').permute(0, 3, 1, 2)
    out.copy_(t)
    return out


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().cuda()


def test_ops_against_golden():
    import ctgan_amd.functional as F
    import ctgan_amd.kernels as K
    z = G.load('ops.npz')
    for i in range(6):
        N, C, H, W, Ko, k, s = [int(v) for v in z['conv%d_cfg' % i]]
        geom = K.ConvGeom(C, H, W, Ko, k, k, s, False)
        x, w, b, gy = cl(z['conv%d_x' % i]), dv(z['conv%d_w' % i]), dv(z['conv%d_b' % i]), cl(z['conv%d_gy' % i])
        assert rel(K.conv_fwd(x, w, b, geom).cpu().numpy(), z['conv%d_y' % i]) < 2e-5
        assert rel(K.conv_dgrad(gy, w, geom, N).cpu().numpy(), z['conv%d_gx' % i]) < 2e-5
        assert rel(K.conv_wgrad(x, gy, geom).cpu().numpy(), z['conv%d_gw' % i]) < 3e-5
    for i in range(3):
        y = F.conv2d_transpose(cl(z['deconv%d_x' % i]), dv(z['deconv%d_w' % i]), dv(z['deconv%d_b' % i]))
        assert rel(y.cpu().numpy(), z['deconv%d_y' % i]) < 2e-5
    x, u = cl(z['ew_x']), cl(z['ew_u'])
    assert rel(K.dropout(x, u, 0.8).cpu().numpy(), z['ew_drop08']) < 1e-6
    assert rel(K.dropout(x, u, 0.5).cpu().numpy(), z['ew_drop05']) < 1e-6
    assert rel(K.lrelu_fwd(x, 0.2).cpu().numpy(), z['ew_lrelu']) < 1e-6
    assert rel(K.pool2(x, 0.25).cpu().numpy(), z['ew_pool']) < 1e-6
    assert rel(K.upsample2(x, 1.0).cpu().numpy(), z['ew_up']) < 1e-7
    lab = torch.from_numpy(z['bn_labels']).cuda()
    y, _, _, _ = K.bn_fwd(cl(z['bn_x']), dv(z['cbn_scale']), dv(z['cbn_offset']), lab, 1, False)
    assert rel(y.cpu().numpy(), z['cbn_y']) < 2e-5
    y, _, _, _ = K.bn_fwd(cl(z['bn_x']), dv(z['bn_scale']).view(1, -1), dv(z['bn_offset']).view(1, -1), None, 1, False)
    assert rel(y.cpu().numpy(), z['bn_y']) < 2e-5
    y, _, _, _ = K.bn_fwd(dv(z['bn0_x']), dv(z['bn0_scale']), dv(z['bn0_offset']), None, 1, False)
    assert rel(y.cpu().numpy(), z['bn0_y']) < 2e-5
    ct, _ = K.ct_fwd(dv(z['ct_d']), dv(z['ct_d_']), dv(z['ct_f']), dv(z['ct_f_']), 2.0, 0.0)
    assert rel(ct.item(), z['ct_M0']) < 1e-5
    ct, _ = K.ct_fwd(dv(z['ct_d']), dv(z['ct_d_']), dv(z['ct_f']), dv(z['ct_f_']), 2.0, 0.5)
    assert rel(ct.item(), z['ct_M05']) < 1e-5
    gp, _ = K.gp_fwd(dv(z['gp_g']), 10.0)
    assert rel(gp.item(), z['gp_val']) < 1e-5
    th = dv(z['adam_theta'][0]); m = torch.zeros_like(th); v = torch.zeros_like(th)
    state = torch.tensor([0.0, 0.5, 0.9, 0.0], device='cuda')
    for t in range(1, 4):
        state[0] = 2e-4 * (1 - t / 10.)
        K.adam_step(th, dv(z['adam_g'][t - 1]), m, v, state, 0.5, 0.9)
        K.adam_advance(state, 0.5, 0.9)
        assert rel(th.cpu().numpy(), z['adam_theta'][t]) < 1e-6


def test_resnet_trace_against_golden():
    """Free-running replay of the recorded 2-iteration trace: losses within the north-star 1e-3."""
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    z = G.load('resnet_trace.npz')
    dim, B, iters = [int(v) for v in z['cfg']]
    lib.delete_all_params(); lib.set_device(None)
    try:
        R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
        lib.load_state_dict(G.trace_weights(z), strict=False)
        assert [n for n, _ in lib.named_params_with_name('Discriminator.', True)] == [str(n) for n in z['d_names']]
        tr = R.Trainer(seed=0)
        for it in range(iters):
            real, labels, rnd = G.trace_d_inputs(z, it, device='cuda')
            o = tr.d_step(real, labels, rnd, iteration=it)
            for k in ('cost', 'ct', 'gp', 'acgan', 'wgan_only'):
                want = float(z['it%d.d.out.%s' % (it, k)])
                assert abs(o[k].item() - want) <= 1e-3 * abs(want) + 1e-6, (it, k, o[k].item(), want)
            assert rel(o['fake'].cpu().numpy(), z['it%d.d.out.fake' % it]) < 2e-4
            gn = np.array([o['grads'][n].norm().item() for n in [str(s) for s in z['d_names']]])
            assert np.all(np.abs(gn - z['it%d.d.gradnorm' % it]) <= 2e-3 * z['it%d.d.gradnorm' % it] + 1e-6)
            o = tr.g_step(G.trace_g_inputs(z, it, device='cuda'), iteration=it + 1)
            want = float(z['it%d.g.out.cost' % it])
            assert abs(o['cost'].item() - want) <= 1e-3 * abs(want) + 5e-4
    finally:
        lib.delete_all_params(); R.configure()
