"""tflib.ops.layernorm.Layernorm against the oracle restatement of TF/tflib/ops/layernorm.py:
values, first-order gradients and the second-order path of a gradient penalty (d/d theta of || d y / d x ||).
Host logic on the torch-CPU stand-in kernels; the GPU twin is in test_gpu_kernels.py."""
import numpy as np
import pytest
import torch

from oracle import tflib_ref as oref


def _rel(a, b):
    a = a.detach().double(); b = b.detach().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _check(lib, dev, shape, tol, relu=False):
    from ctgan_amd.tflib.ops import layernorm as ln
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g) * 1.7 + 0.3
    C = shape[1]
    lib.param('L.scale', (torch.rand(C, generator=g) + 0.5).numpy())
    lib.param('L.offset', torch.randn(C, generator=g).numpy())
    xd = x.to(dev).requires_grad_(True)
    y = ln.Layernorm('L', list(range(1, len(shape))), xd, relu=relu)
    reg = oref.Registry(dtype=torch.float64)
    for n in ('L.scale', 'L.offset'):
        reg[n] = lib._params[n].detach().cpu().double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    yr = oref.Layernorm(reg, 'L', list(range(1, len(shape))), xr)
    if relu:
        yr = torch.relu(yr)          # Normalize -> nonlinearity of the critics' blocks, fused into the Layernorm kernels
    assert tuple(y.shape) == tuple(yr.shape) and _rel(y.cpu(), yr) < tol
    sc, of = lib._params['L.scale'], lib._params['L.offset']
    gy = torch.randn(*shape, generator=g)
    got = torch.autograd.grad(y, [xd, sc, of], gy.to(dev), create_graph=True)
    ref = torch.autograd.grad(yr, [xr, reg['L.scale'], reg['L.offset']], gy.double(), create_graph=True)
    for a, b in zip(got, ref):
        assert _rel(a.cpu(), b) < tol
    # gradient-penalty shape: lambda * mean((||dy/dx|| - 1)^2) differentiated w.r.t. x and the scale
    def gp(gx):
        n = gx.reshape(gx.shape[0], -1).pow(2).sum(dim=1).sqrt() if gx.dtype == torch.float64 else None
        return n
    pen_ref = ((ref[0].reshape(shape[0], -1).pow(2).sum(dim=1).sqrt() - 1) ** 2).mean()
    g2_ref = torch.autograd.grad(pen_ref, [xr, reg['L.scale']])
    import ctgan_amd.functional as F
    pen, _ = F.gradient_penalty(got[0].reshape(shape[0], -1), 1.0)
    g2 = torch.autograd.grad(pen, [xd, sc])
    assert abs(pen.item() - pen_ref.item()) < tol * max(1.0, abs(pen_ref.item()))
    for a, b in zip(g2, g2_ref):
        assert _rel(a.cpu(), b) < 20 * tol


@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('shape', [(5, 8, 4, 4), (3, 16, 8, 8), (7, 24)])
def test_layernorm_values_gradients_and_double_backward(cpu_kernels, shape, relu):
    import ctgan_amd.tflib as lib
    _check(lib, 'cpu', shape, 2e-5, relu)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(6, 128, 8, 8), (3, 64, 16, 16), (9, 40), (2, 128, 32, 32), (3, 1024, 8, 8), (5, 256)])
@pytest.mark.parametrize('relu', [False, True])
def test_layernorm_on_gpu(shape, relu):
    import ctgan_amd.tflib as lib
    lib.delete_all_params(); lib.set_device(None)
    try:
        _check(lib, 'cuda', shape, 3e-5, relu)
    finally:
        lib.delete_all_params()
