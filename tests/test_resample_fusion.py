"""ConvMeanPool / UpsampleConv as single stride-2 (transposed) convs with the spread 4x4 filter
(functional.conv2d_mean_pool / upsample_conv2d) against the reference formulation
(TF/CT_gan_cifar_resnet.py:89-92, :100-107) computed by the oracle's TF-semantics ops:
values, first-order gradients and the double backward of the gradient-penalty path.
Host logic on the torch-CPU stand-in kernels; the GPU twin is in test_gpu_kernels.py."""
import numpy as np
import pytest
import torch

from oracle import tf_ops


def rel(a, b):
    a = a.detach().double(); b = b.detach().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _ref_pool(x, w, b):
    y = tf_ops.bias_add_nchw(tf_ops.conv2d_same(x, w, 1), b)
    return (y[:, :, ::2, ::2] + y[:, :, 1::2, ::2] + y[:, :, ::2, 1::2] + y[:, :, 1::2, 1::2]) / 4.


def _ref_up(x, w, b):
    return tf_ops.bias_add_nchw(tf_ops.conv2d_same(tf_ops.upsample2(x), w, 1), b)


@pytest.mark.parametrize('mode', ['pool', 'up'])
@pytest.mark.parametrize('k', [3, 1, 5])
def test_fused_resample_conv_matches_reference_formulation(cpu_kernels, mode, k):
    import ctgan_amd.functional as F
    g = torch.Generator().manual_seed(5 + k)
    N, C, Ko, H = 3, 8, 12, 8
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
    b = torch.randn(Ko, generator=g)
    xd = x.clone().requires_grad_(True); wd = w.clone().requires_grad_(True); bd = b.clone().requires_grad_(True)
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True); br = b.double().requires_grad_(True)
    if mode == 'pool':
        y = F.conv2d_mean_pool(xd, wd, bd)
        yr = _ref_pool(xr, wr, br)
    else:
        y = F.upsample_conv2d(xd, wd, bd)
        yr = _ref_up(xr, wr, br)
    assert tuple(y.shape) == tuple(yr.shape)
    assert rel(y, yr) < 1e-5
    gy = torch.randn(yr.shape, generator=g)
    got = torch.autograd.grad(y, [xd, wd, bd], gy, create_graph=True)
    ref = torch.autograd.grad(yr, [xr, wr, br], gy.double(), create_graph=True)
    for a, c in zip(got, ref):
        assert rel(a, c) < 1e-5
    # gradient-penalty shape: d/dw and d/dgy-free second order of <dL/dx, v>
    v = torch.randn(x.shape, generator=g)
    s = (got[0] * v).sum(); sr = (ref[0] * v.double()).sum()
    (gw2,) = torch.autograd.grad(s, [wd]); (gw2r,) = torch.autograd.grad(sr, [wr])
    assert rel(gw2, gw2r) < 1e-5


def test_spread_fold_are_adjoint(cpu_kernels):
    import ctgan_amd.kernels as K
    g = torch.Generator().manual_seed(1)
    w = torch.randn(3, 3, 5, 7, generator=g)
    for flip in (False, True):
        w4 = K.filter_spread(w, 0.25, flip)
        u = torch.randn(w4.shape, generator=g)
        assert abs((w4 * u).sum().item() - (w * K.filter_fold(u, 0.25, flip)).sum().item()) < 1e-4
