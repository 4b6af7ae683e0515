"""Parity of every HIP kernel family against the CPU oracle (fp64 truth), through the C-ABI.
Runs on the MI355X box only (`-m gpu`)."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_ops  # noqa: E402


@pytest.fixture(scope='module')
def K():
    """This module tests the fp32 MFMA family itself (kernel names, bit-equality between its variants): the routing of large stride-1
    layers to the split mode (kernels.X3_HYBRID, covered by test_gpu_kernels16.py and the step / loop parity tests) is off here."""
    import ctgan_amd.kernels as K
    old, K.X3_HYBRID = K.X3_HYBRID, False
    yield K
    K.X3_HYBRID = old


def dev(t):
    return t.to('cuda')


def cl(t):
    """channels-last copy on device"""
    d = t.to('cuda')
    out = torch.empty((d.shape[0], d.shape[2], d.shape[3], d.shape[1]), device='cuda', dtype=d.dtype).permute(0, 3, 1, 2)
    out.copy_(d)
    return out


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# (N, C, H, W, K, k, stride, x_layout, up)   -- covers vector + generic loaders, every tile shape
CONV_CASES = [
    (4, 128, 8, 8, 128, 3, 1, 'cl', False),      # hot 3x3, 32x128 tile
    (16, 128, 16, 16, 128, 3, 1, 'cl', False),    # 64x128
    (64, 128, 32, 32, 128, 3, 1, 'cl', False),    # 128x128 tile (M=65536)
    (3, 3, 32, 32, 128, 3, 1, 'nchw', False),     # first critic conv: generic A from NCHW image
    (3, 3, 16, 16, 128, 1, 1, 'cl', False),       # 1x1 shortcut from pooled image
    (5, 128, 16, 16, 128, 1, 1, 'cl', False),     # 1x1 shortcut
    (3, 128, 32, 32, 3, 3, 1, 'cl', False),       # generator output conv (K=3: generic B)
    (4, 128, 8, 8, 128, 3, 1, 'cl', True),        # UpsampleConv: x_up gather, physical 4x4
    (2, 3, 32, 32, 128, 5, 2, 'nchw', False),     # DCGAN critic layer 1 (stride 2, asymmetric pad)
    (2, 128, 16, 16, 256, 5, 2, 'cl', False),     # DCGAN critic layer 2
    (3, 64, 7, 7, 96, 5, 2, 'cl', False),         # odd size, pad (2,2); K=96 -> partial N tile
    (8, 128, 32, 32, 128, 4, 2, 'cl', False),     # ConvMeanPool as one 4x4 stride-2 conv; dgrad = 4 output phases of 2x2 taps
    (3, 128, 16, 16, 128, 4, 2, 'cl', False),     # same, small M
    (5, 64, 8, 12, 96, 4, 2, 'cl', False),        # phases with a partial N tile, H != W
    (2, 32, 8, 8, 64, 5, 2, 'cl', False),         # phase dgrad, odd filter (3/2-tap phases zero-padded), table-driven kernel
    (2, 1, 28, 28, 64, 5, 2, 'cl', False),        # MNIST layer 1
    (7, 40, 6, 5, 24, 3, 1, 'cl', False),         # nothing aligned: fully generic
    (130, 128, 1, 1, 2048, 1, 1, 'cl', False),    # Linear 128->2048 as a 1x1 conv
    (64, 8192, 1, 1, 1, 1, 1, 'cl', False),       # DCGAN critic head (GEMV)
    (192, 128, 1, 1, 10, 1, 1, 'cl', False),      # ACGAN head: small-N linear kernels
    (70, 128, 1, 1, 1, 1, 1, 'cl', False),        # critic output head
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'N%d_C%d_H%dx%d_K%d_k%d_s%d_%s%s' % (c[:8] + ('_up' if c[8] else '',)))
def test_conv_fwd_dgrad_wgrad(K, case):
    N, C, H, W, Ko, k, st, layout, up = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 1000)   # hash() of a str is per-process random
    Hp, Wp = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(N, C, Hp, Wp, generator=g)
    w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
    b = torch.randn(Ko, generator=g)
    xin = tf_ops.upsample2(x.double()) if up else x.double()
    ref = tf_ops.bias_add_nchw(tf_ops.conv2d_same(xin, w.double(), st), b.double())
    geom = K.ConvGeom(C, H, W, Ko, k, k, st, up)
    xd = cl(x) if layout == 'cl' else dev(x)
    y = K.conv_fwd(xd, dev(w), dev(b), geom)
    name_fwd = K.last_kernel()
    assert tuple(y.shape) == tuple(ref.shape)
    assert relerr(y, ref) < 2e-5, name_fwd
    # fused epilogue: + resid, relu ; NCHW output
    r = torch.randn(ref.shape, generator=g)
    y2 = K.conv_fwd(xd, dev(w), dev(b), geom, resid=cl(r), relu=True)
    assert relerr(y2, torch.relu(ref + r.double())) < 2e-5
    y3 = K.conv_fwd(xd, dev(w), None, geom, out_strides=tuple(torch.empty(ref.shape).stride()))
    assert y3.is_contiguous() and relerr(y3, ref - b.double().view(1, -1, 1, 1)) < 2e-5

    # dgrad / wgrad against autograd of the oracle
    gy = torch.randn(ref.shape, generator=g)
    xin_ = xin.clone().requires_grad_(True)
    w_ = w.double().clone().requires_grad_(True)
    out = tf_ops.conv2d_same(xin_, w_, st)
    gx_ref, gw_ref = torch.autograd.grad(out, [xin_, w_], gy.double())
    gw = K.conv_wgrad(xd, cl(gy), geom)
    assert relerr(gw, gw_ref) < 3e-5, K.last_kernel()
    gw_b = K.conv_wgrad(xd, dev(gy), geom)                     # NCHW-strided dy
    assert relerr(gw_b, gw_ref) < 3e-5, K.last_kernel()
    gw_c, gb_c = K.conv_wgrad(xd, cl(gy), geom, with_bias=True)  # bias gradient fused where the pipelined kernel applies
    gb_ref = gy.double().sum(dim=(0, 2, 3))
    gb_scale = gy.double().abs().sum(dim=(0, 2, 3)).max().item()       # a sum of N(0,1) draws may cancel to ~0
    assert relerr(gw_c, gw_ref) < 3e-5 and (gb_c.cpu().double() - gb_ref).abs().max().item() < 1e-6 * gb_scale, K.last_kernel()
    if not up:
        gx = K.conv_dgrad(cl(gy), dev(w), geom, N)
        assert relerr(gx, gx_ref) < 2e-5, K.last_kernel()
        gx2 = K.conv_dgrad(dev(gy), dev(w), geom, N, out_strides=tuple(x.stride()), bias=None)
        assert gx2.is_contiguous() and relerr(gx2, gx_ref) < 2e-5


@pytest.mark.parametrize('N,H,C,Ko,k', [(64, 32, 128, 128, 4), (16, 16, 128, 128, 4), (3, 16, 128, 256, 5), (2, 8, 64, 96, 4)])
def test_stride2_dgrad_phase_decomposition(K, N, H, C, Ko, k):
    """Stride-2 data gradient as four stride-1 convs (one per output parity, one launch): pipelined and
    table-driven kernels, pre-repacked phase filter, bias / mask / resid epilogue, NCHW output."""
    g = torch.Generator().manual_seed(N + H + k)
    geom = K.ConvGeom(C, H, H, Ko, k, k, 2, False)
    w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
    gy = torch.randn(N, Ko, H // 2, H // 2, generator=g)
    x_ = torch.zeros(N, C, H, H, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(tf_ops.conv2d_same(x_, w.double(), 2), x_, gy.double())
    dx = K.conv_dgrad(cl(gy), dev(w), geom, N)
    name = K.last_kernel()
    assert 'ph4' in name, name
    assert relerr(dx, ref) < 2e-5, name
    wt = K.repack_filter(dev(w), geom)
    b = torch.randn(C, generator=g); m = torch.randn(ref.shape, generator=g); r = torch.randn(ref.shape, generator=g)
    dx2 = K.conv_dgrad(cl(gy), dev(w), geom, N, bias=dev(b), wt=wt, mask=cl(m), resid=cl(r))
    ref2 = torch.where(m.double() > 0, ref + b.double().view(1, -1, 1, 1), torch.zeros_like(ref)) + r.double()
    assert relerr(dx2, ref2) < 2e-5
    dx3 = K.conv_dgrad(dev(gy), dev(w), geom, N, out_strides=tuple(torch.empty(ref.shape).stride()))
    assert dx3.is_contiguous() and relerr(dx3, ref) < 2e-5
    K.debug_force_generic(True)
    try:
        dx4 = K.conv_dgrad(cl(gy), dev(w), geom, N)
        assert 'igemm_fwd<' in K.last_kernel() and 'ph4' in K.last_kernel()
    finally:
        K.debug_force_generic(False)
    assert relerr(dx4, ref) < 2e-5
    if 'pipe' in name and ',k1' in name:
        assert torch.equal(dx, dx4)          # same taps, same order


@pytest.mark.parametrize('mode', ['pool', 'up'])
@pytest.mark.parametrize('N,C,Ko,H', [(6, 128, 128, 16), (3, 64, 96, 8)])
def test_fused_resample_convs_on_gpu(K, mode, N, C, Ko, H):
    """functional.conv2d_mean_pool / upsample_conv2d (spread 4x4 filter, stride-2 conv / phase dgrad) against the
    reference formulation conv->pool / upsample->conv of the oracle: values, gradients, GP-style double backward."""
    import ctgan_amd.functional as F
    g = torch.Generator().manual_seed(N + C + H)
    x = torch.randn(N, C, H, H, generator=g); w = torch.randn(3, 3, C, Ko, generator=g) / np.sqrt(9 * C)
    b = torch.randn(Ko, generator=g)
    xd = cl(x).requires_grad_(True); wd = dev(w).requires_grad_(True); bd = dev(b).requires_grad_(True)
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True); br = b.double().requires_grad_(True)
    if mode == 'pool':
        y = F.conv2d(xd, wd, bd, pool=True)
        yr = tf_ops.bias_add_nchw(tf_ops.conv2d_same(xr, wr, 1), br)
        yr = (yr[:, :, ::2, ::2] + yr[:, :, 1::2, ::2] + yr[:, :, ::2, 1::2] + yr[:, :, 1::2, 1::2]) / 4.
    else:
        y = F.conv2d(xd, wd, bd, x_up=True)
        yr = tf_ops.bias_add_nchw(tf_ops.conv2d_same(tf_ops.upsample2(xr), wr, 1), br)
    assert tuple(y.shape) == tuple(yr.shape) and relerr(y, yr) < 2e-5
    gy = torch.randn(yr.shape, generator=g)
    got = torch.autograd.grad(y, [xd, wd, bd], cl(gy), create_graph=True)
    ref = torch.autograd.grad(yr, [xr, wr, br], gy.double(), create_graph=True)
    for a, c in zip(got, ref):
        assert relerr(a, c) < 3e-5
    v = torch.randn(x.shape, generator=g)
    (gw2,) = torch.autograd.grad((got[0] * cl(v)).sum(), [wd]); (gw2r,) = torch.autograd.grad((ref[0] * v.double()).sum(), [wr])
    assert relerr(gw2, gw2r) < 3e-5


def test_filter_spread_fold(K):
    g = torch.Generator().manual_seed(3)
    w = torch.randn(3, 3, 40, 24, generator=g)
    for flip in (False, True):
        ref = torch.zeros(4, 4, 40, 24, dtype=torch.float64)
        for a in (0, 1):
            for b in (0, 1):
                ref[a:a + 3, b:b + 3] += w.double()
        ref = ref * 0.25
        if flip:
            ref = torch.flip(ref, (0, 1)).permute(0, 1, 3, 2)
        got = K.filter_spread(dev(w), 0.25, flip)
        assert tuple(got.shape) == tuple(ref.shape) and relerr(got, ref) < 1e-6
        u = torch.randn(ref.shape, generator=g)
        ud = u.double()
        if flip:
            ud = torch.flip(ud, (0, 1)).permute(0, 1, 3, 2)
        fref = sum(ud[a:a + 3, b:b + 3] for a in (0, 1) for b in (0, 1)) * 0.5
        assert relerr(K.filter_fold(dev(u), 0.5, flip), fref) < 1e-6


@pytest.mark.parametrize('N,C,H,Ko,k,st,layout', [(5, 3, 32, 128, 3, 1, 'nchw'), (64, 3, 32, 128, 3, 1, 'nchw'), (4, 3, 16, 128, 1, 1, 'cl'),
                                                    (3, 128, 32, 3, 3, 1, 'cl'), (33, 128, 32, 3, 3, 1, 'cl'), (6, 64, 8, 3, 3, 1, 'cl'),
                                                    (4, 1, 28, 64, 5, 2, 'cl'), (2, 256, 8, 3, 3, 1, 'cl')])
def test_few_channel_direct_kernels(K, N, C, H, Ko, k, st, layout):
    """csrc/fewch.hip: convs with <= 4 channels on one side (critic conv 1 and its shortcut, generator output conv,
    MNIST first conv) run on direct FMA kernels for forward, data gradient and weight gradient (+ bias)."""
    g = torch.Generator().manual_seed(N * 7 + C + H)
    geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
    assert K.fewch_handles(geom)
    x = torch.randn(N, C, H, H, generator=g); w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
    b = torch.randn(Ko, generator=g)
    xd = cl(x) if layout == 'cl' else dev(x)
    ref = tf_ops.bias_add_nchw(tf_ops.conv2d_same(x.double(), w.double(), st), b.double())
    y = K.conv_fwd(xd, dev(w), dev(b), geom)
    assert 'fewch' in K.last_kernel(), K.last_kernel()
    assert relerr(y, ref) < 2e-5
    if C <= 4:
        r = torch.randn(ref.shape, generator=g)
        y2 = K.conv_fwd(xd, dev(w), dev(b), geom, resid=cl(r), relu=True, relu_in=True)
        ref2 = torch.relu(tf_ops.bias_add_nchw(tf_ops.conv2d_same(torch.relu(x.double()), w.double(), st), b.double()) + r.double())
        assert 'fewch' in K.last_kernel() and relerr(y2, ref2) < 2e-5
    else:
        y3 = K.conv_fwd(xd, dev(w), dev(b), geom, out_strides=tuple(torch.empty(ref.shape).stride()))      # NCHW output (G.Output)
        assert 'fewch' in K.last_kernel() and y3.is_contiguous() and relerr(y3, ref) < 2e-5
    gy = torch.randn(ref.shape, generator=g)
    x_ = x.double().clone().requires_grad_(True); w_ = w.double().clone().requires_grad_(True)
    gx_ref, gw_ref = torch.autograd.grad(tf_ops.conv2d_same(x_, w_, st), [x_, w_], gy.double())
    gyd = cl(gy) if C <= 4 else dev(gy)          # the few-channel side may be plain NCHW
    gw, gb = K.conv_wgrad(xd, gyd, geom, with_bias=True)
    assert 'fewch_wgrad' in K.last_kernel(), K.last_kernel()
    gb_scale = gy.double().abs().sum(dim=(0, 2, 3)).max().item()
    assert relerr(gw, gw_ref) < 3e-5 and (gb.cpu().double() - gy.double().sum(dim=(0, 2, 3))).abs().max().item() < 1e-6 * gb_scale
    gw2 = K.conv_wgrad(xd, gyd, geom)
    assert torch.equal(gw, gw2)
    if st == 1:
        gx = K.conv_dgrad(gyd, dev(w), geom, N, out_strides=tuple(x.stride()) if layout == 'nchw' else None)
        assert 'fewch' in K.last_kernel(), K.last_kernel()
        assert relerr(gx, gx_ref) < 2e-5
    # the table-driven GEMM kernels remain the cross-check
    K.debug_force_generic(True)
    try:
        y_gen = K.conv_fwd(xd, dev(w), dev(b), geom)
        assert 'fewch' not in K.last_kernel()
    finally:
        K.debug_force_generic(False)
    assert relerr(y, y_gen) < 1e-5


def test_conv_vector_and_generic_paths_agree_bitwise(K):
    """The vector loaders only change how tiles reach LDS: same MFMA order => identical bits.
    A misaligned (offset-by-one-element) view forces the scalar-gather path on the same values."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 128, 8, 8, generator=g)
    w = torch.randn(3, 3, 128, 128, generator=g) * 0.03
    geom = K.ConvGeom(128, 8, 8, 128, 3, 3, 1, False)
    xa = cl(x)
    K.debug_force_generic(True)
    try:
        y_vec = K.conv_fwd(xa, dev(w), None, geom)
        k_vec = K.last_kernel()
    finally:
        K.debug_force_generic(False)
    buf = torch.empty(xa.numel() + 1, device='cuda')
    xb = buf[1:].view(4, 8, 8, 128).permute(0, 3, 1, 2)
    xb.copy_(xa)
    wbuf = torch.empty(w.numel() + 1, device='cuda')
    wb = wbuf[1:].view(3, 3, 128, 128)
    wb.copy_(dev(w))
    y_gen = K.conv_fwd(xb, wb, None, geom)
    k_gen = K.last_kernel()
    assert 'avec,bvec' in k_vec and 'agen,bgen' in k_gen, (k_vec, k_gen)
    assert torch.equal(y_vec, y_gen)


# (N, H, up, stride-2 dgrad?) -> every pipelined tile configuration, incl. the K-split one
@pytest.mark.parametrize('N,H,up', [(128, 32, False), (64, 16, False), (24, 16, False), (64, 8, False), (7, 8, False),
                                     (128, 8, True), (3, 4, False)])
def test_pipelined_conv_matches_table_driven_kernel(K, N, H, up):
    """igemm_fwd_pipe (double-buffered LDS, B prefetch ring, clamped-address padding, XCD tile order,
    optional in-block K split) against the simple table-driven kernel on the same inputs."""
    g = torch.Generator().manual_seed(N * 100 + H)
    Hp = H // 2 if up else H
    x = cl(torch.randn(N, 128, Hp, Hp, generator=g))
    w = dev(torch.randn(3, 3, 128, 128, generator=g) * 0.03)
    b = dev(torch.randn(128, generator=g))
    geom = K.ConvGeom(128, H, H, 128, 3, 3, 1, up)
    r = cl(torch.randn(N, 128, H, H, generator=g))
    y = K.conv_fwd(x, w, b, geom, resid=r, relu=True)
    name = K.last_kernel()
    assert 'igemm_fwd_pipe' in name
    K.debug_force_generic(True)
    try:
        y_ref = K.conv_fwd(x, w, b, geom, resid=r, relu=True)
        assert 'igemm_fwd<' in K.last_kernel()
        gy = cl(torch.randn(N, 128, H, H, generator=g))
        dx_ref = None if up else K.conv_dgrad(gy, w, geom, N)
    finally:
        K.debug_force_generic(False)
    if ',k1' in name:
        assert torch.equal(y, y_ref), name              # same summation order => same bits
    else:
        assert relerr(y, y_ref) < 1e-5, name             # K halves summed separately
    if not up:
        dx = K.conv_dgrad(gy, w, geom, N)
        assert 'igemm_fwd_pipe' in K.last_kernel()
        assert relerr(dx, dx_ref) < 1e-5


@pytest.mark.parametrize('N,H,C,Ko', [(128, 32, 128, 128), (32, 16, 128, 128), (64, 16, 128, 128), (64, 8, 128, 128),
                                      (5, 8, 64, 128), (3, 4, 256, 512), (9, 8, 128, 256)])
def test_pipelined_wgrad_matches_table_driven_kernel(K, N, H, C, Ko):
    g = torch.Generator().manual_seed(N + H + C)
    x = cl(torch.randn(N, C, H, H, generator=g)); gy = cl(torch.randn(N, Ko, H, H, generator=g))
    geom = K.ConvGeom(C, H, H, Ko, 3, 3, 1, False)
    dw, db = K.conv_wgrad(x, gy, geom, with_bias=True)
    name = K.last_kernel()
    assert 'igemm_wgrad_pipe' in name and 'bias' in name
    dw2 = K.conv_wgrad(x, gy, geom)
    assert torch.equal(dw, dw2)                                   # bias row does not perturb the weights
    K.debug_force_generic(True)
    try:
        ref = K.conv_wgrad(x, gy, geom)
        assert 'igemm_wgrad<' in K.last_kernel()
    finally:
        K.debug_force_generic(False)
    assert torch.equal(dw, ref), name                              # same split plan, same MFMA order => same bits
    assert relerr(db, gy.double().sum(dim=(0, 2, 3))) < 2e-5


@pytest.mark.parametrize('N,C,H,Ko,k,st,layout', [(5, 3, 32, 128, 3, 1, 'nchw'), (3, 3, 32, 128, 5, 2, 'nchw'),
                                                    (4, 1, 28, 64, 5, 2, 'cl'), (2, 3, 7, 16, 3, 1, 'cl')])
def test_few_channel_conv_via_im2col(K, N, C, H, Ko, k, st, layout):
    """functional.conv2d routes C<=4 convs through im2col + 1x1 MFMA convs; fwd, dgrad (col2im), wgrad,
    bias grad and the double backward (GP path) against oracle autograd."""
    import ctgan_amd.functional as F
    g = torch.Generator().manual_seed(N + C + H)
    x = torch.randn(N, C, H, H, generator=g); w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
    b = torch.randn(Ko, generator=g)
    xd = (cl(x) if layout == 'cl' else dev(x)).requires_grad_(True)
    wd = dev(w).requires_grad_(True); bd = dev(b).requires_grad_(True)
    y = F.conv2d(xd, wd, bd, stride=st)
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True); br = b.double().requires_grad_(True)
    yr = tf_ops.bias_add_nchw(tf_ops.conv2d_same(xr, wr, st), br)
    assert relerr(y, yr) < 2e-5
    gy = torch.randn(yr.shape, generator=g)
    got = torch.autograd.grad(y, [xd, wd, bd], cl(gy), create_graph=True)
    ref = torch.autograd.grad(yr, [xr, wr, br], gy.double(), create_graph=True)
    for a, c in zip(got, ref):
        assert relerr(a, c) < 3e-5
    # second order: d/dw of <dL/dx, v>
    v = torch.randn(x.shape, generator=g)
    vd = cl(v) if layout == 'cl' else dev(v)
    (gw2,) = torch.autograd.grad((got[0] * vd).sum(), wd)
    (gw2r,) = torch.autograd.grad((ref[0] * v.double()).sum(), wr)
    assert relerr(gw2, gw2r) < 3e-5


@pytest.mark.parametrize('N,C,H,Ko,k', [(192, 96, 16, 128, 1), (64, 96, 16, 128, 1), (256, 32, 8, 64, 3)])
def test_split_k_reduction_with_four_lanes_per_float4_gives_the_same_bits(K, N, C, H, Ko, k):
    """splitk_reduce_lanes_kernel (small filter, many slabs: the 1x1 weight gradient of the im2col'd first critic conv, TF/CT_gan_cifar.py:84) sums the
    slabs in the order of splitk_reduce_kernel - lane j = that kernel's accumulator j - so the weight and bias gradients are bit-identical."""
    g = torch.Generator().manual_seed(N + C)
    geom = K.ConvGeom(C, H, H, Ko, k, k, 1, False)
    x = cl(torch.randn(N, C, H, H, generator=g)); gy = cl(torch.randn(N, Ko, H, H, generator=g))
    dw, db = K.conv_wgrad(x, gy, geom, with_bias=True)
    K.lib.ctgan_debug_reduce_lanes(0)
    try:
        dw0, db0 = K.conv_wgrad(x, gy, geom, with_bias=True)
    finally:
        K.lib.ctgan_debug_reduce_lanes(1)
    assert torch.equal(dw, dw0) and torch.equal(db, db0), K.last_kernel()
    assert relerr(db, gy.double().sum(dim=(0, 2, 3))) < 2e-5


@pytest.mark.parametrize('N,C,H,W,k,st,cpad,layout', [(256, 3, 32, 32, 5, 2, 96, 'nchw'), (64, 3, 32, 32, 5, 2, 96, 'cl'), (4, 1, 28, 28, 5, 2, 32, 'nchw'),
                                                          (2, 3, 7, 9, 3, 1, 32, 'cl'), (3, 4, 9, 6, 3, 2, 36, 'nchw'), (5, 2, 8, 8, 1, 1, 4, 'nchw')])
def test_im2col_band_kernel_and_col2im_pixel_kernel(K, N, C, H, W, k, st, cpad, layout):
    """ctgan_im2col (band kernel: LDS rows, 16-byte stores) is pure data movement - exact against the slicing definition - and ctgan_col2im
    (one thread per pixel over the taps that exist) is its adjoint: against the scatter-add definition to fp32 summation error, and
    <im2col(x), c> = <x, col2im(c)>.  The first critic conv of TF/CT_gan_cifar.py:84 / TF/CT_gan_mnist.py:92 (5x5, stride 2) and odd shapes."""
    from tests import cpu_kernels as M
    geom = K.ConvGeom(C, H, W, 8, k, k, st, False)
    g = torch.Generator().manual_seed(N * 7 + C + H)
    x = torch.randn(N, C, H, W, generator=g)
    xd = cl(x) if layout == 'cl' else dev(x)
    cols = K.im2col(xd, geom, cpad)
    assert torch.equal(cols.cpu(), M.im2col(x, geom, cpad))
    c = torch.randn(N, cpad, geom.P, geom.Q, generator=g)
    ref = M.col2im(c.double(), geom, N)
    for strides in (None, (C * H * W, H * W, W, 1)):
        dx = K.col2im(cl(c), geom, N, strides)
        assert relerr(dx, ref) < 2e-6
    lhs = (cols.double().cpu() * c.double()).sum().item()
    rhs = (x.double() * K.col2im(cl(c), geom, N).double().cpu()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


@pytest.mark.parametrize('N,C,H,Ko,k', [(64, 128, 32, 128, 3), (16, 128, 8, 128, 3), (3, 40, 6, 24, 3), (8, 128, 16, 128, 1)])
def test_fused_relu_in_and_mask_epilogue(K, N, C, H, Ko, k):
    """conv(relu(x)) with the ReLU applied while staging the tile; its data gradient with the ReLU mask
    (+ residual) in the epilogue; its weight gradient with relu-on-load - vs the unfused composition."""
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, H, H, generator=g); w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
    b = torch.randn(Ko, generator=g); gy = torch.randn(N, Ko, H, H, generator=g); r = torch.randn(N, C, H, H, generator=g)
    geom = K.ConvGeom(C, H, H, Ko, k, k, 1, False)
    xd, wd, bd, gyd, rd = cl(x), dev(w), dev(b), cl(gy), cl(r)
    xr = K.lrelu_fwd(xd, 0.0)
    y = K.conv_fwd(xd, wd, bd, geom, relu_in=True)
    assert torch.equal(y, K.conv_fwd(xr, wd, bd, geom)), K.last_kernel()
    dx = K.conv_dgrad(gyd, wd, geom, N, mask=xd, resid=rd)
    ref = K.axpby(K.lrelu_bwd(K.conv_dgrad(gyd, wd, geom, N), xd, 0.0), rd, 1.0, 1.0)
    assert torch.equal(dx, ref), K.last_kernel()
    dw, db = K.conv_wgrad(xd, gyd, geom, with_bias=True, relu_x=True)
    dw2, db2 = K.conv_wgrad(xr, gyd, geom, with_bias=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), K.last_kernel()


@pytest.mark.parametrize('N,C,H,Ko,k,st', [(128, 128, 32, 128, 3, 1), (192, 128, 32, 128, 4, 2), (320, 128, 16, 128, 3, 1),
                                           (192, 3, 32, 128, 3, 1), (128, 128, 32, 3, 3, 1), (384, 128, 8, 128, 3, 1)])
def test_full_size_conv_properties(K, N, C, H, Ko, k, st):
    """Size-independent properties at the sizes the step runs (too large for the fp64 oracle in seconds): the data gradient
    is the adjoint of the forward, <conv(x,w), gy> = <x, dgrad(gy,w)> = <w, wgrad(x,gy)>; linearity in x; and batch-split
    invariance (rows of a batch are independent: the forward of the whole batch equals the forwards of its halves)."""
    g = torch.Generator().manual_seed(N + C + H + Ko + k)
    geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
    mk_x = (lambda n: dev(torch.randn(n, C, H, H, generator=g))) if C <= 4 else (lambda n: cl(torch.randn(n, C, H, H, generator=g)))
    mk_y = (lambda n: dev(torch.randn(n, Ko, geom.P, geom.Q, generator=g))) if Ko <= 4 else (lambda n: cl(torch.randn(n, Ko, geom.P, geom.Q, generator=g)))
    x, x2, gy = mk_x(N), mk_x(N), mk_y(N)
    w = dev(torch.randn(k, k, C, Ko, generator=g) * 0.05)
    y = K.conv_fwd(x, w, None, geom)
    gx = K.conv_dgrad(gy, w, geom, N)
    gw = K.conv_wgrad(x, gy, geom)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    lhs = dot(y, gy)
    scale = float(y.double().norm() * gy.double().norm())
    assert abs(lhs - dot(x, gx)) < 1e-5 * scale and abs(lhs - dot(w, gw)) < 1e-5 * scale
    y12 = K.conv_fwd(K.axpby(x, x2, 0.75, -1.5), w, None, geom)
    y2 = K.conv_fwd(x2, w, None, geom)
    assert relerr(y12, 0.75 * y.double() - 1.5 * y2.double()) < 2e-5
    h = N // 2
    # (another batch size may pick another tile / K-split configuration: same sums in another fp32 order)
    assert relerr(K.conv_fwd(x[:h], w, None, geom), y[:h]) < 1e-5 and relerr(K.conv_fwd(x[h:], w, None, geom), y[h:]) < 1e-5


def test_conv_is_deterministic(K):
    g = torch.Generator().manual_seed(8)
    x = cl(torch.randn(32, 128, 16, 16, generator=g)); gy = cl(torch.randn(32, 128, 16, 16, generator=g))
    w = dev(torch.randn(3, 3, 128, 128, generator=g))
    geom = K.ConvGeom(128, 16, 16, 128, 3, 3, 1, False)
    a = K.conv_wgrad(x, gy, geom).clone(); b = K.conv_wgrad(x, gy, geom)
    assert 'split' in K.last_kernel() and torch.equal(a, b)
    assert torch.equal(K.conv_fwd(x, w, None, geom), K.conv_fwd(x, w, None, geom))


def test_deconv_matches_tf_conv2d_transpose(K):
    import ctgan_amd.functional as F
    g = torch.Generator().manual_seed(9)
    for (n, ci, co, h, w_) in [(2, 256, 128, 4, 4), (3, 128, 64, 7, 7), (2, 64, 3, 16, 16), (2, 64, 1, 14, 14)]:
        x = torch.randn(n, ci, h, w_, generator=g)
        w = torch.randn(5, 5, co, ci, generator=g) / np.sqrt(25 * ci / 4)
        b = torch.randn(co, generator=g)
        ref = tf_ops.bias_add_nchw(tf_ops.conv2d_transpose_same(x.double(), w.double(), 2), b.double())
        xd = cl(x).requires_grad_(True); wd = dev(w).requires_grad_(True); bd = dev(b).requires_grad_(True)
        y = F.conv2d_transpose(xd, wd, bd)
        assert tuple(y.shape) == (n, co, 2 * h, 2 * w_) and relerr(y, ref) < 2e-5
        gy = torch.randn(ref.shape, generator=g)
        x_ = x.double().requires_grad_(True); w_r = w.double().requires_grad_(True); b_ = b.double().requires_grad_(True)
        r = tf_ops.bias_add_nchw(tf_ops.conv2d_transpose_same(x_, w_r, 2), b_)
        gr = torch.autograd.grad(r, [x_, w_r, b_], gy.double())
        gp = torch.autograd.grad(y, [xd, wd, bd], cl(gy))
        for a, c in zip(gp, gr):
            assert relerr(a, c) < 3e-5


def test_elementwise(K):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 12, 6, 6, generator=g); u = torch.rand(5, 12, 6, 6, generator=g)
    xd, ud = cl(x), cl(u)
    assert torch.equal(K.lrelu_fwd(xd, 0.0).cpu(), torch.relu(x))
    assert relerr(K.lrelu_fwd(xd, 0.2), tf_ops.leaky_relu(x.double())) < 1e-7
    assert relerr(K.lrelu_bwd(ud, xd, 0.2), torch.where(x > 0, u, 0.2 * u)) < 1e-7
    for keep in (0.8, 0.5):
        assert relerr(K.dropout(xd, ud, keep), tf_ops.dropout(x, keep, u)) < 1e-6
        assert relerr(K.dropout(xd, dev(u), keep), tf_ops.dropout(x, keep, u)) < 1e-6       # layout repack of u
    assert relerr(K.tanh_fwd(xd), torch.tanh(x.double())) < 1e-6
    assert relerr(K.sigmoid_fwd(xd), torch.sigmoid(x.double())) < 1e-6
    y = torch.tanh(x)
    assert relerr(K.tanh_bwd(ud, cl(y)), u.double() * (1 - y.double() ** 2)) < 1e-6
    assert relerr(K.axpby(xd, ud, 2.0, -0.5), 2 * x.double() - 0.5 * u.double()) < 1e-6
    assert relerr(K.axpby(xd, None, 3.0, 0.0), 3 * x.double()) < 1e-7
    assert torch.equal(K.to_nchw(xd).cpu(), x) and K.to_nchw(xd).is_contiguous()
    assert torch.equal(K.to_channels_last(dev(x)).cpu(), x)
    xe = torch.randn(3, 5, 8, 10, generator=g)
    assert relerr(K.pool2(cl(xe), 0.25), tf_ops.mean_pool2(xe.double())) < 1e-6
    assert relerr(K.pool2(dev(xe), 0.25), tf_ops.mean_pool2(xe.double())) < 1e-6          # NCHW input
    assert torch.equal(K.upsample2(cl(xe), 1.0).cpu(), tf_ops.upsample2(xe))
    assert relerr(K.spatial_sum(cl(xe), 1.0 / 80), xe.double().mean(dim=(2, 3))) < 1e-6
    gg = torch.randn(3, 5, generator=g)
    assert relerr(K.spatial_bcast(dev(gg), 8, 10, 0.5), (0.5 * gg)[:, :, None, None].expand(3, 5, 8, 10)) < 1e-7
    xi = torch.randint(0, 256, (6, 3072), generator=g, dtype=torch.int32)
    nz = torch.rand(6, 3072, generator=g) / 128
    assert relerr(K.real_prep(dev(xi), dev(nz), 256.0), 2 * (xi.double() / 256. - .5) + nz.double()) < 1e-6
    assert relerr(K.real_prep(dev(xi), None, 255.0), 2 * (xi.double() / 255. - .5)) < 1e-6
    a = torch.rand(6, 1, generator=g); f = torch.randn(6, 3072, generator=g); r = torch.randn(6, 3072, generator=g)
    assert relerr(K.interpolate(dev(r), dev(f), dev(a)), r.double() + a.double() * (f.double() - r.double())) < 1e-6
    big = torch.randn(1000, 130, generator=g)
    assert relerr(K.colsum_channels(cl(big.view(10, 100, 130).permute(0, 2, 1).unsqueeze(3).contiguous())),
                  big.double().sum(0)) < 1e-5


@pytest.mark.parametrize('n,c,h,groups,cond,relu', [(8, 128, 4, 1, True, True), (6, 16, 8, 2, True, True),
                                                     (4, 128, 32, 1, False, True), (8, 200, 1, 1, False, False),
                                                     (64, 128, 16, 2, True, True)])
def test_batchnorm(K, n, c, h, groups, cond, relu):
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, h, h, generator=g) * 2 + 3.0             # |mean| >> 0: exercises the fp64 stats
    nl = 10 if cond else 1
    scale = torch.rand(nl, c, generator=g) + 0.5; offset = torch.randn(nl, c, generator=g)
    labels = torch.randint(0, 10, (n,), generator=g, dtype=torch.int32) if cond else None
    gy = torch.randn(n, c, h, h, generator=g)
    xr = x.double().requires_grad_(True); sr = scale.double().requires_grad_(True); orr = offset.double().requires_grad_(True)
    per = n // groups
    outs = []
    for gi in range(groups):
        xs = xr[gi * per:(gi + 1) * per]
        mean, var = tf_ops.moments(xs, [0, 2, 3])
        lab = labels[gi * per:(gi + 1) * per].long() if cond else torch.zeros(per, dtype=torch.long)
        outs.append(tf_ops.batch_normalization(xs, mean, var, orr[lab][:, :, None, None], sr[lab][:, :, None, None], 1e-5))
    ref = torch.cat(outs)
    if relu:
        ref = torch.relu(ref)
    gr = torch.autograd.grad(ref, [xr, sr, orr], gy.double())
    y, mean, rstd, x4 = K.bn_fwd(cl(x), dev(scale), dev(offset), dev(labels) if cond else None, groups, relu)
    assert relerr(y, ref) < 2e-5
    gx, gs, go = K.bn_bwd(cl(gy), x4, mean, rstd, dev(scale), dev(offset), dev(labels) if cond else None, groups, relu)
    assert relerr(gx, gr[0]) < 1e-4 and relerr(gs, gr[1]) < 1e-4 and relerr(go, gr[2]) < 1e-4


def test_loss_heads(K):
    g = torch.Generator().manual_seed(5)
    gr = torch.randn(64, 3072, generator=g) * 0.02
    s = gr.double().norm(dim=1)
    gp, slopes = K.gp_fwd(dev(gr), 10.0)
    assert relerr(slopes, s) < 1e-6 and relerr(gp, 10 * ((s - 1) ** 2).mean()) < 1e-6
    grd = gr.double().requires_grad_(True)
    ref = torch.autograd.grad(10 * ((grd.norm(dim=1) - 1) ** 2).mean() * 0.7, grd)[0]
    assert relerr(K.gp_bwd(dev(gr), slopes, dev(torch.tensor(0.7)), 10.0), ref) < 1e-5
    d, d_ = torch.randn(64, generator=g), torch.randn(64, generator=g)
    f, f_ = torch.randn(64, 128, generator=g), torch.randn(64, 128, generator=g)
    from oracle import steps
    for M in (0.0, 3.0):
        leaves = [t.double().requires_grad_(True) for t in (d, d_, f, f_)]
        ref = steps.ct_term(*leaves, 2.0, M)
        refg = torch.autograd.grad(ref * 1.3, leaves)
        ct, ct_i = K.ct_fwd(dev(d), dev(d_), dev(f), dev(f_), 2.0, M)
        assert relerr(ct, ref) < 1e-6
        got = K.ct_bwd(dev(d), dev(d_), dev(f), dev(f_), ct_i, dev(torch.tensor(1.3)), 2.0, M)
        for a, b in zip(got, refg):
            assert relerr(a, b) < 1e-5
    logits = torch.randn(64, 10, generator=g) * 3
    labels = torch.randint(0, 10, (64,), generator=g, dtype=torch.int32)
    lr = logits.double().requires_grad_(True)
    ref = tf_ops.sparse_softmax_ce(lr, labels).mean()
    loss, probs, nc = K.softmax_ce_fwd(dev(logits), dev(labels))
    assert relerr(loss, ref) < 1e-6 and nc.item() == (logits.argmax(1) == labels.long()).sum().item()
    assert relerr(K.softmax_ce_bwd(probs, dev(labels), dev(torch.tensor(2.0))), torch.autograd.grad(ref * 2, lr)[0]) < 1e-5
    x = torch.randn(128, generator=g)
    assert relerr(K.mean_diff_fwd(dev(x), 64, 64, -1.0, 1.0), x[64:].double().mean() - x[:64].double().mean()) < 1e-5
    assert relerr(K.mean_diff_fwd(dev(x), 128, 0, -1.0, 0.0), -x.double().mean()) < 1e-5
    gx = K.mean_diff_bwd(dev(torch.tensor(2.0)), 64, 64, -1.0, 1.0).cpu()
    assert torch.allclose(gx[:64], torch.full((64,), -2.0 / 64)) and torch.allclose(gx[64:], torch.full((64,), 2.0 / 64))


def test_tf_adam_kernel(K):
    g = torch.Generator().manual_seed(6)
    n = 100003
    th = torch.randn(n, generator=g); m = torch.zeros(n); v = torch.zeros(n)
    thd, md, vd = dev(th).clone(), dev(m).clone(), dev(v).clone()
    state = torch.tensor([1e-3, 0.5, 0.9, 0.0], device='cuda')
    ref_t, ref_m, ref_v = th.double(), m.double(), v.double()
    for t in range(1, 4):
        gr = torch.randn(n, generator=g)
        ref_t, ref_m, ref_v = tf_ops.tf_adam_step(ref_t, gr.double(), ref_m, ref_v, t, 1e-3, 0.5, 0.9)
        K.adam_step(thd, dev(gr), md, vd, state, 0.5, 0.9)
        K.adam_advance(state, 0.5, 0.9)
        assert relerr(thd, ref_t) < 1e-6 and relerr(md, ref_m) < 1e-6 and relerr(vd, ref_v) < 1e-6
    assert abs(state[1].item() - 0.5 ** 4) < 1e-7


def test_packed_update_and_step_advance_are_bit_identical_to_the_separate_launches(K):
    """adam_step_packed (gradient bucket + Adam in one launch) + step_advance (beta powers + Philox counter in one launch) against
    pack + adam_step + adam_advance + rng_advance, bit for bit, over several steps (a None gradient, unaligned sizes)."""
    sizes = [128 * 128 * 9, 37, 128, 3 * 3 * 3 * 128, 1, 4096]
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    n = sum(sizes)

    def fresh():
        g = torch.Generator().manual_seed(3)
        th = torch.randn(n, generator=g)
        return [dev(th), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'),
                dev(torch.tensor([2e-4, 0.5, 0.9, 0.0])), torch.zeros(1, dtype=torch.int64, device='cuda')]
    A, B = fresh(), fresh()                                  # [theta, m, v, flat, state, ctr]
    gen = torch.Generator().manual_seed(11)
    for t in range(1, 5):
        srcs = [dev(torch.randn(k, generator=gen)) if (i != 2 or t % 2) else None for i, k in enumerate(sizes)]
        K.pack(srcs, offs, sizes, A[3])
        K.adam_step(A[0], A[3], A[1], A[2], A[4], 0.5, 0.9, 1e-8, 0.5)
        K.adam_advance(A[4], 0.5, 0.9)
        K.rng_advance(A[5], 1)
        K.adam_step_packed(srcs, offs, sizes, B[3], B[0], B[1], B[2], B[4], 0.5, 0.9, 1e-8, 0.5)
        K.step_advance(B[4], 0.5, 0.9, B[5], 1)
        for a, x in zip(A, B):
            assert torch.equal(a, x)
        assert int(B[5]) == t


def _philox_np(seed, sid, step, nblk):
    """numpy Philox4x32-10 with the kernel's counter layout (known-answer checked below)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = np.zeros((nblk, 4), dtype=np.uint64)
    c[:, 0] = np.arange(nblk); c[:, 1] = sid; c[:, 2] = step & 0xffffffff; c[:, 3] = step >> 32
    k0, k1 = seed & 0xffffffff, seed >> 32
    for _ in range(10):
        p0 = M0 * c[:, 0]; p1 = M1 * c[:, 2]
        n0 = (p1 >> 32) ^ c[:, 1] ^ k0; n1 = p1 & 0xffffffff
        n2 = (p0 >> 32) ^ c[:, 3] ^ k1; n3 = p0 & 0xffffffff
        c = np.stack([n0, n1, n2, n3], 1) & 0xffffffff
        k0 = (k0 + W0) & 0xffffffff; k1 = (k1 + W1) & 0xffffffff
    return c.astype(np.uint32)


def test_philox_known_answer_and_streams(K):
    # Random123 known-answer vector: counter=0, key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    assert [hex(v) for v in _philox_np(0, 0, 0, 1)[0]] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    ctr = torch.zeros(1, dtype=torch.int64, device='cuda')
    out = torch.empty(1001, device='cuda')
    K.rng_uniform(out, 2024, 5, ctr, 0.0, 1.0)
    exp = (_philox_np(2024, 5, 0, 251).reshape(-1)[:1001] >> 8).astype(np.float32) / np.float32(16777216.0)
    assert np.array_equal(out.cpu().numpy(), exp)                                  # bit-exact
    a = out.clone()
    K.rng_advance(ctr, 1)
    K.rng_uniform(out, 2024, 5, ctr, 0.0, 1.0)
    assert not torch.equal(a, out) and ctr.item() == 1
    exp1 = (_philox_np(2024, 5, 1, 251).reshape(-1)[:1001] >> 8).astype(np.float32) / np.float32(16777216.0)
    assert np.array_equal(out.cpu().numpy(), exp1)
    big = torch.empty(1 << 20, device='cuda')
    K.rng_uniform(big, 1, 2, ctr, 0.0, 1.0)
    assert 0 <= big.min() and big.max() < 1 and abs(big.mean().item() - 0.5) < 2e-3
    K.rng_normal(big, 1, 3, ctr)
    assert abs(big.mean().item()) < 5e-3 and abs(big.std().item() - 1) < 5e-3
    lab = torch.empty(100000, dtype=torch.int32, device='cuda')
    K.rng_labels(lab, 10, 1, 4, ctr)
    assert lab.min() == 0 and lab.max() == 9
    assert np.allclose(np.bincount(lab.cpu().numpy(), minlength=10) / 1e5, 0.1, atol=0.01)


def test_errors_surface_as_python_exceptions(K):
    with pytest.raises(RuntimeError):
        K.lrelu_fwd(torch.zeros(4), 0.0)                   # CPU tensor: no fallback
    x = torch.zeros(4, device='cuda')
    with pytest.raises(ValueError):
        K.dropout(x, x, 0.0)                               # keep must be in (0,1]


def test_filter_batch_matches_single_kernels(K):
    """ctgan_filter_batch (all derived filters of a weight update in one launch) against the stand-alone kernels."""
    g = torch.Generator().manual_seed(9)
    w3 = dev(torch.randn(3, 3, 64, 96, generator=g)); w4 = dev(torch.randn(4, 4, 32, 64, generator=g)); w5 = dev(torch.randn(5, 5, 32, 64, generator=g))
    g3 = K.ConvGeom(64, 8, 8, 96, 3, 3, 1, False); g4 = K.ConvGeom(32, 8, 8, 64, 4, 4, 2, False); g5 = K.ConvGeom(32, 8, 8, 64, 5, 5, 2, False)
    assert K.dgrad_filter_kind(g3) == K.FILTER_ROTATE and K.dgrad_filter_kind(g4) == K.FILTER_PHASES
    jobs = []
    for w, gg in ((w3, g3), (w4, g4), (w5, g5)):
        kind = K.dgrad_filter_kind(gg)
        jobs.append((w, torch.empty(K.filter_job_shape(kind, *w.shape), device='cuda'), kind, gg.pad_t, gg.pad_l, 1.0))
    for flip, kind in ((False, K.FILTER_SPREAD), (True, K.FILTER_SPREAD_FLIP)):
        jobs.append((w3, torch.empty(K.filter_job_shape(kind, *w3.shape), device='cuda'), kind, 0, 0, 0.25))
    jobs = jobs * 6                                  # > CTGAN_FILTER_BATCH jobs: several launches
    jobs = [(s, torch.empty_like(d), k, a, b, c) for (s, d, k, a, b, c) in jobs]
    K.filter_batch(jobs)
    for (src, dst, kind, pt, pl, scale), gg in zip(jobs[:3], (g3, g4, g5)):
        assert torch.equal(dst, K.repack_filter(src, gg).reshape(-1))
    assert torch.equal(jobs[3][1], K.filter_spread(w3, 0.25, False)) and torch.equal(jobs[4][1], K.filter_spread(w3, 0.25, True))
    assert torch.equal(jobs[-1][1], jobs[4][1]) and torch.equal(jobs[-3][1].reshape(-1), jobs[2][1].reshape(-1))
    # composed jobs: the data-gradient layout OF a spread filter straight from the parameter (job.pre) is bit-identical to
    # the layout job run on the materialised spread filter; odd channel counts exercise the ragged 32 x 32 tiles
    for C, Ko in ((64, 96), (40, 72)):
        w = dev(torch.randn(3, 3, C, Ko, generator=g))
        for flip, pre in ((False, K.FILTER_SPREAD), (True, K.FILTER_SPREAD_FLIP)):
            sp = K.filter_spread(w, 0.25, flip)                               # [4,4,C,K] or [4,4,K,C]
            Ce, Ke = sp.shape[2], sp.shape[3]
            gs = K.ConvGeom(Ce, 8, 8, Ke, 4, 4, 2, False)
            for kind in (K.FILTER_PHASES, K.FILTER_ROTATE):
                ref = torch.empty(K.filter_job_shape(kind, *sp.shape), device='cuda')
                K.filter_batch([(sp, ref, kind, gs.pad_t, gs.pad_l, 1.0)])
                got = torch.empty_like(ref)
                K.filter_batch([(w, got, kind, gs.pad_t, gs.pad_l, 1.0, pre, 0.25)])
                assert torch.equal(got, ref), (C, Ko, flip, kind)
                if kind == K.dgrad_filter_kind(gs):
                    assert torch.equal(ref, K.repack_filter(sp, gs).reshape(-1))


def test_filter_fold_batch_matches_single_folds(K):
    """ctgan_filter_fold_batch: every fold of a step in one launch = the stand-alone fold kernel, job by job."""
    g = torch.Generator().manual_seed(19)
    jobs, refs = [], []
    for (R, S, C, Ko, flip) in ((3, 3, 128, 128, False), (3, 3, 128, 128, True), (1, 1, 128, 64, False), (3, 3, 40, 24, True)) * 5:
        w4 = dev(torch.randn(*((R + 1, S + 1, Ko, C) if flip else (R + 1, S + 1, C, Ko)), generator=g))
        out = torch.empty(R, S, C, Ko, device='cuda')
        jobs.append((w4, 0.25, flip, out)); refs.append(K.filter_fold(w4, 0.25, flip))
    K.filter_fold_batch(jobs)                       # 20 jobs: two launches
    for (w4, sc, flip, out), ref in zip(jobs, refs):
        assert torch.equal(out, ref)


def test_fused_philox_dropout_equals_draw_then_dropout(K):
    """ctgan_dropout_rng regenerates the mask from (seed, site, device counter): bit-identical to drawing the
    uniform tensor first; backward and double backward reuse the same mask."""
    import ctgan_amd.functional as F
    from ctgan_amd.rng import DeviceRNG
    g = torch.Generator().manual_seed(2)
    x = cl(torch.randn(6, 128, 8, 8, generator=g))
    ctr = torch.full((1,), 5, dtype=torch.int64, device='cuda')
    u = K.rng_uniform(K.empty_cl(6, 128, 8, 8, 'cuda'), 77, 3, ctr)
    assert torch.equal(K.dropout_rng(x, 0.8, 77, 3, ctr), K.dropout(x, u, 0.8))
    xs = dev(torch.randn(1001, generator=g))                      # ragged tail, 1-D
    us = K.rng_uniform(torch.empty(1001, device='cuda'), 77, 4, ctr)
    assert torch.equal(K.dropout_rng(xs, 0.5, 77, 4, ctr), K.dropout(xs, us, 0.5))
    rng = DeviceRNG(seed=77, rank=0, device='cuda'); rng.ctr.fill_(5)
    rng.begin_step(); rng._site = 3
    xr = x.clone().requires_grad_(True)
    y = F.dropout(xr, 0.8, rng=rng)
    assert torch.equal(y, K.dropout(x, u, 0.8))
    gy = dev(torch.randn(6, 128, 8, 8, generator=g)).requires_grad_(True)   # NCHW-contiguous: other physical order than x
    (gx,) = torch.autograd.grad(y, xr, gy, create_graph=True)
    m = K.dropout(torch.ones_like(x), u, 0.8)
    assert relerr(gx, gy.detach() * m) < 1e-6
    v = cl(torch.randn(6, 128, 8, 8, generator=g))
    (ggy,) = torch.autograd.grad(gx, gy, v)                        # double backward: the same mask again
    assert relerr(ggy, v * m) < 1e-6


@pytest.mark.parametrize('with_a', [True, False])
def test_fused_critic_heads(K, with_a):
    """ctgan_critic_heads_fwd/bwd against the separate reference formulas (:244-248, :288-291) in fp64 autograd."""
    import ctgan_amd.functional as F
    g = torch.Generator().manual_seed(8)
    B, nf, ncls = 16, 128, 10
    d = torch.randn(3 * B, generator=g); f = torch.randn(3 * B, nf, generator=g); a = torch.randn(3 * B, ncls, generator=g)
    lab = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32)
    lam2, M, scale = 2.0, 0.3, 0.7
    dr = d.double().requires_grad_(True); fr = f.double().requires_grad_(True); ar = a.double().requires_grad_(True)
    wgan = dr[B:2 * B].mean() - dr[:B].mean()
    ct_i = lam2 * (dr[:B] - dr[2 * B:]) ** 2 + 0.1 * lam2 * ((fr[:B] - fr[2 * B:]) ** 2).mean(dim=1)
    ct = torch.clamp(ct_i - M, min=0).mean()
    ac = torch.nn.functional.cross_entropy(ar[:B], lab.long()) if with_a else torch.zeros((), dtype=torch.float64)
    cost = wgan + ct + (scale * ac if with_a else 0)
    dd = dev(d).requires_grad_(True); fd = dev(f).requires_grad_(True); ad = dev(a).requires_grad_(True)
    gpv = dev(torch.tensor(0.37)).requires_grad_(True)
    got = F.critic_heads(dd, fd, ad if with_a else None, dev(lab), B, lam2, M, scale if with_a else 0.0, gp=gpv)
    assert abs(got[4].item() - (wgan + ct + 0.37).item()) < 2e-5
    (g_gp,) = torch.autograd.grad(got[0], gpv, retain_graph=True)
    assert abs(g_gp.item() - 1.0) < 1e-6
    cost = cost + 0.37
    for x, y in zip(got, (cost, wgan, ct, ac)):
        assert abs(x.item() - y.item()) < 2e-5 * max(1.0, abs(y.item()))
    ins = [dd, fd] + ([ad] if with_a else []); rins = [dr, fr] + ([ar] if with_a else [])
    gg = torch.autograd.grad(got[0], ins, retain_graph=True); gr = torch.autograd.grad(cost, rins, retain_graph=True)
    for x, y in zip(gg, gr):
        assert relerr(x, y) < 2e-5
    # gradients of the individual heads (n_gout = 4 path)
    w = [0.3, -1.1, 0.6, 0.9]
    comb = sum(wi * gi for wi, gi in zip(w, got)); combr = w[0] * cost + w[1] * wgan + w[2] * ct + w[3] * ac
    gg = torch.autograd.grad(comb, ins); gr = torch.autograd.grad(combr, rins)
    for x, y in zip(gg, gr):
        assert relerr(x, y) < 2e-5
    acc = K.accuracy2(dev(a[:2 * B].contiguous()), dev(lab), B)
    am = a[:2 * B].argmax(dim=1)
    assert abs(acc[0].item() - (am[:B] == lab.long()).float().mean().item()) < 1e-6
    assert abs(acc[1].item() - (am[B:] == lab.long()).float().mean().item()) < 1e-6


def test_critic_prep_and_rows_cat_dropout_equal_their_compositions(K):
    """ctgan_critic_prep = rng_uniform + real_prep + rng_uniform + interpolate + concat; ctgan_rows_cat_dropout = concat +
    dropout_rng; ctgan_rows_cat_bwd = the concat's adjoint - all on the same Philox streams."""
    g = torch.Generator().manual_seed(12)
    B, d = 7, 3072
    xi = dev(torch.randint(0, 256, (B, d), generator=g, dtype=torch.int32)); fake = dev(torch.rand(B, d, generator=g) * 2 - 1)
    ctr = torch.full((1,), 9, dtype=torch.int64, device='cuda')
    rf, interp, both = K.critic_prep(xi, fake, 2024, 5, 6, ctr, 0.0, 1. / 128, 256.0)
    assert both.shape == (3 * B, d) and both.data_ptr() == rf.data_ptr()
    deq = K.rng_uniform(torch.empty(B, d, device='cuda'), 2024, 5, ctr, 0.0, 1. / 128)
    real = K.real_prep(xi, deq, 256.0)
    alpha = K.rng_uniform(torch.empty(B, 1, device='cuda'), 2024, 6, ctr)
    assert relerr(rf[:B], real) < 1e-7 and torch.equal(rf[B:], fake)
    assert relerr(interp, K.interpolate(real, fake, alpha)) < 1e-6
    h = cl(torch.randn(10, 128, 8, 8, generator=g))
    for keep in (0.8, 1.0):
        y = K.rows_cat_dropout(h, 4, keep, 77, 3, ctr)
        cat = K.to_channels_last(torch.cat([h, h[:4]], 0))
        ref = K.dropout_rng(cat, keep, 77, 3, ctr) if keep < 1 else cat
        assert y.shape == ref.shape and torch.equal(y, ref)
    gy = cl(torch.randn(14, 128, 8, 8, generator=g))
    gh = K.rows_cat_bwd(gy, 10, 4)
    ref = gy[:10].clone(); ref[:4] += gy[10:]
    assert torch.equal(gh, ref)


def test_row_range_dropout_in_conv_epilogue_and_row_gather(K):
    """ctgan_epilogue_ext row ranges: each sample range of one conv launch gets the dropout its own tensor would get
    (bit-identical to dropout_rng on the range's rows); ctgan_rows_gather_dropout builds [x ; x[:k] | x | x[j:]] with
    per-group dropout in one launch."""
    g = torch.Generator().manual_seed(31)
    ctr = torch.full((1,), 4, dtype=torch.int64, device='cuda')
    geom = K.ConvGeom(128, 8, 8, 128, 3, 3, 1, False)
    x = cl(torch.randn(40, 128, 8, 8, generator=g)); w = dev(torch.randn(3, 3, 128, 128, generator=g) * 0.03); b = dev(torch.randn(128, generator=g))
    r = cl(torch.randn(40, 128, 8, 8, generator=g))
    s_a, s_b = (0.5, 99, 7, ctr), (0.8, 99, 11, ctr)
    for relu in (False, True):
        y0 = K.conv_fwd(x, w, b, geom, resid=r, relu=relu, relu_in=True)
        y = K.conv_fwd(x, w, b, geom, resid=r, relu=relu, relu_in=True, drop={'ranges': [(18, s_a), (30, None), (40, s_b)]})
        assert 'igemm_fwd_pipe' in K.last_kernel()
        assert torch.equal(y[:18], K.dropout_rng(y0[:18], *s_a))
        assert torch.equal(y[18:30], y0[18:30])
        assert torch.equal(y[30:], K.dropout_rng(y0[30:], *s_b))
    h = cl(torch.randn(12, 128, 8, 8, generator=g))
    out = K.rows_gather_dropout(h, [(0, 8, 0.8, 7, 0), (0, 4, 0.8, 7, 0), (0, 8, 1.0, 0, 12), (8, 4, 0.8, 11, 20)], 99, ctr)
    main = K.dropout_rng(K.to_channels_last(torch.cat([h[:8], h[:4]], 0)), 0.8, 99, 7, ctr)
    assert out.shape[0] == 24 and torch.equal(out[:12], main) and torch.equal(out[12:20], h[:8])
    assert torch.equal(out[20:], K.dropout_rng(h[8:], 0.8, 99, 11, ctr))


@pytest.mark.parametrize('B,nf,H,with_a', [(16, 128, 8, True), (5, 32, 4, True), (7, 64, 8, False), (3, 256, 2, True)])
def test_fused_critic_tail_heads(K, B, nf, H, with_a):
    """F.critic_tail_heads (reduce_mean + both Linear heads + loss heads, TF/CT_gan_cifar_resnet.py:179-186,244-248,288-291)
    and F.gp_head_grad against the same graph in fp64 torch autograd: losses, critic outputs, the gradient w.r.t. the last
    conv's result (mask and 1/keep included), the head weight gradients; and the gp head's double backward."""
    import ctgan_amd.functional as F
    g = torch.Generator().manual_seed(B * 1000 + nf)
    ncls, keep = 10, 0.5
    z = torch.randn(3 * B, nf, H, H, generator=g)
    mask = (torch.rand(3 * B, nf, H, H, generator=g) < keep).float() / keep
    y = torch.relu(z) * mask                                    # = relu(dropout(z)), what the last conv's epilogue writes
    w_out = torch.randn(nf, 1, generator=g) * 0.1; b_out = torch.randn(1, generator=g)
    w_ac = torch.randn(nf, ncls, generator=g) * 0.1; b_ac = torch.randn(ncls, generator=g)
    lab = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32)
    lam2, M, scale = 2.0, 0.05, 0.7
    zr = z.double().requires_grad_(True)
    P = [t.double().requires_grad_(True) for t in (w_out, b_out, w_ac, b_ac)]
    fr = (torch.relu(zr) * mask.double()).mean(dim=(2, 3))
    dr = (fr @ P[0]).reshape(-1) + P[1]
    ar = fr @ P[2] + P[3]
    wgan = dr[B:2 * B].mean() - dr[:B].mean()
    ct_i = lam2 * (dr[:B] - dr[2 * B:]) ** 2 + 0.1 * lam2 * ((fr[:B] - fr[2 * B:]) ** 2).mean(dim=1)
    ct = torch.clamp(ct_i - M, min=0).mean()
    ac = torch.nn.functional.cross_entropy(ar[:B], lab.long()) if with_a else torch.zeros((), dtype=torch.float64)
    cost = wgan + ct + 0.37 + (scale * ac if with_a else 0)
    yd = cl(y).requires_grad_(True)
    Pd = [dev(t).requires_grad_(True) for t in (w_out, b_out, w_ac, b_ac)]
    gpv = dev(torch.tensor(0.37)).requires_grad_(True)
    got = F.critic_tail_heads(yd, Pd[0], Pd[1], Pd[2] if with_a else None, Pd[3] if with_a else None, dev(lab), B, lam2, M,
                              scale if with_a else 0.0, 1.0 / keep, gp=gpv)
    for x, ref in zip(got[:4], (cost, wgan, ct, ac)):
        assert abs(x.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    assert relerr(got[5], dr) < 1e-5
    ins = [yd, Pd[0], Pd[1]] + ([Pd[2], Pd[3]] if with_a else [])
    rins = [zr, P[0], P[1]] + ([P[2], P[3]] if with_a else [])
    gg = torch.autograd.grad(got[0], ins, retain_graph=True); gr = torch.autograd.grad(cost, rins, retain_graph=True)
    scale_all = max(float(t.abs().max()) for t in gr)
    for x, ref in zip(gg, gr):
        assert float((x.cpu().double() - ref).norm()) <= 2e-5 * float(ref.norm()) + 1e-6 * scale_all
    w = [0.3, -1.1, 0.6, 0.9]                                    # the four heads differentiated separately (n_gout = 4)
    comb = sum(wi * gi for wi, gi in zip(w, got[:4])); combr = w[0] * cost + w[1] * wgan + w[2] * ct + w[3] * ac
    gg = torch.autograd.grad(comb, ins); gr = torch.autograd.grad(combr, rins, retain_graph=True)
    for x, ref in zip(gg, gr):
        assert float((x.cpu().double() - ref).norm()) <= 2e-5 * float(ref.norm()) + 1e-6 * scale_all
    # relu inside the mean (clean pass)
    f2, _, a2 = K.tail_heads_fwd(cl(z), None, None, dev(w_ac), dev(b_ac), relu=True)
    assert relerr(f2, torch.relu(z).mean(dim=(2, 3))) < 1e-6 and relerr(a2, torch.relu(z).mean(dim=(2, 3)) @ w_ac + b_ac) < 1e-5
    # gradient-penalty head: gz = d(mean_hw(y) . w_out)/dz and its adjoint w.r.t. w_out
    gz = F.gp_head_grad(yd.detach(), Pd[0], 1.0 / keep)
    (gz_ref,) = torch.autograd.grad(dr.sum(), zr, retain_graph=True)
    assert relerr(gz, gz_ref) < 1e-6
    v = torch.randn(3 * B, nf, H, H, generator=g)
    (gw,) = torch.autograd.grad(gz, Pd[0], cl(v))
    gw_ref = ((v.double() * (y > 0).double()).sum(dim=(0, 2, 3)) / (keep * H * H)).reshape(nf, 1)
    assert relerr(gw, gw_ref) < 1e-5


@pytest.mark.parametrize('B,nf,H', [(64, 128, 8), (5, 32, 4)])
def test_penalty_mean_and_clean_accuracies_folded_into_the_heads_launches(K, B, nf, H):
    """ctgan_tail_critic_heads_fwd2: gp from the slopes (written into the slot gp_fwd(defer_mean=True) left) and the clean pass's class
    head + accuracies inside the two launches of the fused heads - bit-identical to gp_fwd + tail_critic_heads_fwd + tail_heads_fwd +
    accuracy2 as separate launches."""
    g = torch.Generator().manual_seed(B + nf)
    ncls = 10
    y = cl(torch.relu(torch.randn(3 * B, nf, H, H, generator=g)))
    yc = cl(torch.relu(torch.randn(2 * B, nf, H, H, generator=g)))
    w_out, b_out = dev(torch.randn(nf, 1, generator=g) * 0.1), dev(torch.randn(1, generator=g))
    w_ac, b_ac = dev(torch.randn(nf, ncls, generator=g) * 0.1), dev(torch.randn(ncls, generator=g))
    lab = dev(torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32))
    grads = dev(torch.randn(B, 3072, generator=g) * 0.02)
    gp0, sl0 = K.gp_fwd(grads, 10.0)
    ref = K.tail_critic_heads_fwd(y, B, w_out, b_out, w_ac, b_ac, lab, gp0.reshape(1), 2.0, 0.0, 1.0)
    _, _, a_c = K.tail_heads_fwd(yc, None, None, w_ac, b_ac, relu=False)
    acc0 = K.accuracy2(a_c.contiguous(), lab, B)
    gp1, sl1 = K.gp_fwd(grads, 10.0, defer_mean=True)
    got = K.tail_critic_heads_fwd(y, B, w_out, b_out, w_ac, b_ac, lab, gp1.reshape(1), 2.0, 0.0, 1.0, slopes=sl1, gp_lambda=10.0,
                                  y_clean=yc, clean_relu=False)
    assert torch.equal(sl0, sl1) and torch.equal(gp0, gp1) and float(gp0) > 0
    for a, b in zip(ref[:6], got[:6]):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
    assert torch.equal(got[6], acc0) and 0.0 <= float(acc0[0]) <= 1.0


@pytest.mark.parametrize('C,H,Ko,k,st,Ns', [(128, 8, 128, 3, 1, (12, 4)), (128, 16, 128, 4, 2, (8, 4, 2)), (64, 8, 96, 1, 1, (5,)),
                                            (128, 32, 128, 3, 1, (64, 16))])
def test_multi_segment_wgrad(K, C, H, Ko, k, st, Ns):
    """ctgan_conv2d_wgrad_multi: one launch over several (x, dy) pairs of one filter = the sum of the separate weight
    gradients; per-segment relu-on-load and bias flags."""
    g = torch.Generator().manual_seed(C + H + sum(Ns))
    geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
    segs, ref_w, ref_b = [], 0, 0
    for i, n in enumerate(Ns):
        x = cl(torch.randn(n, C, H, H, generator=g)); gy = cl(torch.randn(n, Ko, geom.P, geom.Q, generator=g))
        relu_x, with_bias = (i % 2 == 0), (i == 0)
        segs.append((x, gy, relu_x, with_bias))
        r = K.conv_wgrad(x, gy, geom, with_bias=with_bias, relu_x=relu_x)
        ref_w = ref_w + (r[0] if with_bias else r).double()
        if with_bias:
            ref_b = ref_b + r[1].double()
    dw = torch.empty(k, k, C, Ko, device='cuda'); db = torch.empty(Ko, device='cuda')
    K.conv_wgrad_multi(segs, geom, dw, db)
    assert 'igemm_wgrad_pipe' in K.last_kernel()
    assert relerr(dw, ref_w) < 1e-5 and relerr(db, ref_b) < 1e-5
    dw2 = torch.empty_like(dw)
    K.conv_wgrad_multi([(x, gy, r, False) for x, gy, r, _ in segs], geom, dw2, None)
    assert torch.equal(dw, dw2)                                    # the bias row does not perturb the weights
    K.conv_wgrad_multi(segs, geom, dw2, db)
    assert torch.equal(dw, dw2)                                    # deterministic


@pytest.mark.parametrize('C,Ko,k', [(3, 128, 3), (3, 128, 1), (128, 3, 3)])
def test_few_channel_wgrad_two_segments(K, C, Ko, k):
    """ctgan_conv2d_wgrad_multi on a few-channel conv: both uses of the filter (two passes of a step) in one launch of the
    direct kernel = the sum of the separate weight gradients; per-segment relu-on-load and bias flags."""
    g = torch.Generator().manual_seed(C * 7 + k)
    geom = K.ConvGeom(C, 32, 32, Ko, k, k, 1, False)
    assert K.fewch_handles(geom)
    segs, ref_w, ref_b = [], 0, 0
    for i, n in enumerate((12, 5)):
        x = dev(torch.randn(n, C, 32, 32, generator=g)) if C <= 4 else cl(torch.randn(n, C, 32, 32, generator=g))
        gy = cl(torch.randn(n, Ko, 32, 32, generator=g)) if Ko > 4 else dev(torch.randn(n, Ko, 32, 32, generator=g))
        relu_x, with_bias = (i == 1), (i == 0)
        segs.append((x, gy, relu_x, with_bias))
        r = K.conv_wgrad(x, gy, geom, with_bias=with_bias, relu_x=relu_x)
        ref_w = ref_w + (r[0] if with_bias else r).double()
        if with_bias:
            ref_b = ref_b + r[1].double()
    dw = torch.empty(k, k, C, Ko, device='cuda'); db = torch.empty(Ko, device='cuda')
    K.conv_wgrad_multi(segs, geom, dw, db)
    assert 'fewch_wgrad' in K.last_kernel()
    assert relerr(dw, ref_w) < 1e-5 and relerr(db, ref_b) < 1e-5
    dw1 = torch.empty_like(dw)
    K.conv_wgrad_multi(segs[:1], geom, dw1, db)
    r0 = K.conv_wgrad(segs[0][0], segs[0][1], geom, with_bias=True, relu_x=False)
    assert torch.equal(dw1, r0[0]) and torch.equal(db, r0[1])


def test_grouped_wgrad_equals_separate_launches(K):
    """ctgan_conv2d_wgrad_group: the queued weight gradients of a step (different filters, geometries and tile
    configurations, several segments each) from one launch per tile configuration + one reduction launch are
    bit-identical to one ctgan_conv2d_wgrad_multi call per filter."""
    g = torch.Generator().manual_seed(77)
    cases = [(128, 8, 128, 3, 1, (64, 64, 32)), (128, 8, 128, 3, 1, (96,)), (128, 16, 128, 4, 2, (48, 16)), (128, 16, 128, 2, 2, (40,)),
             (128, 16, 128, 3, 1, (64, 8)), (128, 32, 128, 4, 2, (32, 8)), (64, 8, 96, 1, 1, (5,)), (128, 8, 128, 3, 1, (7, 3)),
             (128, 4, 128, 3, 1, (16,)), (128, 8, 128, 3, 1, (33,)), (128, 8, 128, 3, 1, (34,))]
    groups, refs = [], []
    for ci, (C, H, Ko, k, st, Ns) in enumerate(cases):
        geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
        segs = []
        for i, n in enumerate(Ns):
            x = cl(torch.randn(n, C, H, H, generator=g)); gy = cl(torch.randn(n, Ko, geom.P, geom.Q, generator=g))
            segs.append((x, gy, (i + ci) % 2 == 0, i == 0 and ci % 3 != 2))
        has_b = any(sg[3] for sg in segs)
        dw = torch.empty(k, k, C, Ko, device='cuda'); db = torch.empty(Ko, device='cuda') if has_b else None
        dw_r = torch.empty_like(dw); db_r = torch.empty_like(db) if has_b else None
        K.conv_wgrad_multi(segs, geom, dw_r, db_r)
        groups.append((segs, geom, dw, db)); refs.append((dw_r, db_r))
    x3g, K.X3_WGRAD_GROUP = K.X3_WGRAD_GROUP, 0              # the fp32 family's grouped launch (the split-mode one: test below)
    try:
        K.conv_wgrad_group(groups)
    finally:
        K.X3_WGRAD_GROUP = x3g
    assert 'igemm_wgrad_pipe_group' in K.last_kernel()
    for (segs, geom, dw, db), (dw_r, db_r) in zip(groups, refs):
        assert torch.equal(dw, dw_r), (geom.H, geom.R)
        if db is not None:
            assert torch.equal(db, db_r)


@pytest.mark.parametrize('N,H', [(5, 16), (130, 8), (64, 32)])
def test_conv_epilogue_reads_low_resolution_residual(K, N, H):
    """CTGAN_RESID_UP: the residual operand is the [N,K,P/2,Q/2] tensor, added through a nearest-2x upsample inside the
    conv epilogue (generator 'up' block shortcut) - bit-identical to adding the materialised upsample."""
    g = torch.Generator().manual_seed(N + H)
    geom = K.ConvGeom(128, H, H, 128, 3, 3, 1, False)
    x = cl(torch.randn(N, 128, H, H, generator=g)); w = dev(torch.randn(3, 3, 128, 128, generator=g) * 0.03)
    b = dev(torch.randn(128, generator=g)); r = cl(torch.randn(N, 128, H // 2, H // 2, generator=g))
    y = K.conv_fwd(x, w, b, geom, resid=r, resid_up=True, relu_in=True)
    assert 'igemm_fwd_pipe' in K.last_kernel()
    y_ref = K.conv_fwd(x, w, b, geom, resid=K.upsample2(r, 1.0), relu_in=True)
    assert torch.equal(y, y_ref)


def test_adam_leaves_elements_with_a_non_finite_gradient_untouched():
    """ctgan_adam_step / ctgan_adam_step_packed: an inf or NaN gradient element (fp16-mode overflow, degenerate warm-up input) must not
    reach theta, m or v - not even at lr = 0, where 0 * inf = NaN; finite elements take the unchanged path (bit-identical)."""
    import ctgan_amd.tflib as lib
    from ctgan_amd.optim import FlatAdam
    lib.delete_all_params(); lib.set_device(None)
    try:
        p = lib.param('T.w', np.linspace(-1, 1, 64, dtype='float32'))
        ref = lib.param('U.w', np.linspace(-1, 1, 64, dtype='float32'))
        opt, opt_ref = FlatAdam([('T.w', p)], 0.5, 0.9), FlatAdam([('U.w', ref)], 0.5, 0.9)
        g = torch.linspace(1, 2, 64, device='cuda')
        bad = g.clone(); bad[3] = float('inf'); bad[10] = float('nan'); bad[40] = -float('inf')
        for lr in (0.0, 1e-3):
            opt.set_lr(lr); opt_ref.set_lr(lr)
            opt.update([bad], 1.0); opt_ref.update([g], 1.0)
        torch.cuda.synchronize()
        # the skips are counted: 3 elements x 2 packed updates; the two-launch form (gather + step) counts the same way
        assert opt.skipped() == 6 and opt_ref.skipped() == 0
        opt.gather_grads([bad]); opt.step(1.0)
        opt_ref.gather_grads([g]); opt_ref.step(1.0)
        assert opt.skipped() == 9 and opt_ref.skipped() == 0
        ok = torch.ones(64, dtype=torch.bool, device='cuda'); ok[[3, 10, 40]] = False
        assert torch.isfinite(opt.theta).all() and torch.isfinite(opt.m).all() and torch.isfinite(opt.v).all()
        assert torch.equal(opt.theta[ok], opt_ref.theta[ok]) and torch.equal(opt.m[ok], opt_ref.m[ok])
        init = torch.from_numpy(np.linspace(-1, 1, 64, dtype='float32'))
        assert torch.equal(opt.theta[~ok].cpu(), init[~ok.cpu()]) and (opt.m[~ok] == 0).all() and (opt.v[~ok] == 0).all()
    finally:
        lib.delete_all_params()


@pytest.mark.parametrize('N,C,dgrad', [(3, 128, False), (64, 128, True), (5, 64, True), (2, 256, False)])
def test_many_to_few_one_pixel_per_lane_kernel(K, N, C, dgrad):
    """m2f_px_kernel (csrc/fewch.hip, round 5): the 3x3 many -> few convs on 32-pixel rows with one output pixel per lane and the filter
    as scalar operands - the generator's output conv (forward, NCHW result, filter [r,s,many,few]: the few index has unit stride) and the
    data gradient of the first critic conv (filter [r,s,few,many]: the many index has unit stride) - against the fp64 oracle and
    against the row-ring kernel it replaces, with bias / relu-on-load where the entry point has them."""
    g = torch.Generator().manual_seed(N + C)
    H = 32
    if dgrad:
        geom = K.ConvGeom(3, H, H, C, 3, 3, 1, False)
        w = torch.randn(3, 3, 3, C, generator=g) / np.sqrt(27)
        gy = torch.randn(N, C, H, H, generator=g)
        x_ = torch.zeros(N, 3, H, H, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(tf_ops.conv2d_same(x_, w.double(), 1), x_, gy.double())
        run = lambda: K.conv_dgrad(cl(gy), dev(w), geom, N, out_strides=(3 * H * H, H * H, H, 1))      # noqa: E731
        sym = 'm2f_px_kernel<3>'
    else:
        geom = K.ConvGeom(C, H, H, 3, 3, 3, 1, False)
        w = torch.randn(3, 3, C, 3, generator=g) / np.sqrt(9 * C)
        b = torch.randn(3, generator=g)
        x = torch.randn(N, C, H, H, generator=g)
        ref = tf_ops.bias_add_nchw(tf_ops.conv2d_same(torch.relu(x.double()), w.double(), 1), b.double())
        run = lambda: K.conv_fwd(cl(x), dev(w), dev(b), geom, out_strides=(3 * H * H, H * H, H, 1), relu_in=True)      # noqa: E731
        sym = 'm2f_px_kernel<3>'
    y = run()
    assert K.last_symbol() == sym, K.last_symbol()
    assert y.is_contiguous() and relerr(y, ref) < 2e-5
    K.debug_m2f_px(False)
    try:
        y_ring = run()
        assert 'm2f_px' not in K.last_symbol()
    finally:
        K.debug_m2f_px(True)
    assert relerr(y, y_ring) < 2e-6


def test_launches_folded_into_their_neighbours_for_the_hand_scheduled_critic_step(K):
    """Round 5 (critic_schedule.py): (i) ctgan_tail_heads_bwd_gp = tail_heads_bwd + gp_head_grad for the penalty rows in one launch, bit for bit;
    (ii) ctgan_gp_head_wgrad_acc adds onto a finished gradient; (iii) ctgan_gp_finish = upsample2 + add + per-sample norms."""
    g = torch.Generator().manual_seed(9)
    B, nf, hw = 8, 128, 64
    y = cl(torch.randn(4 * B, nf, 8, 8, generator=g))
    w_out = dev(torch.randn(nf, 1, generator=g) * 0.1); b_out = dev(torch.zeros(1))
    w_ac = dev(torch.randn(nf, 10, generator=g) * 0.1); b_ac = dev(torch.zeros(10))
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32).cuda()
    out5, f, d, a, ct_i, probs, _ = K.tail_critic_heads_fwd(y[:3 * B], B, w_out, b_out, w_ac, b_ac, labels, None, 2.0, 0.0, 1.0)
    one = torch.ones(1, device='cuda')
    ref = K.tail_heads_bwd(y[:3 * B], d, f, probs, labels, ct_i, one, B, 2.0, 0.0, 1.0, 2.0, w_out, w_ac)
    gz_ref = K.gp_head_grad(y[3 * B:], w_out, 2.0)
    gy = K.empty_cl(4 * B, nf, 8, 8, 'cuda')
    got = K.tail_heads_bwd(y[:3 * B], d, f, probs, labels, ct_i, one, B, 2.0, 0.0, 1.0, 2.0, w_out, w_ac, out=gy[:3 * B], y_gp=y[3 * B:], out_gp=gy[3 * B:])
    assert torch.equal(gy[:3 * B], ref[0]) and torch.equal(gy[3 * B:], gz_ref)
    for a_, b_ in zip(got[1:], ref[1:]):
        assert torch.equal(a_, b_)
    # (ii)
    gg = cl(torch.randn(B, nf, 8, 8, generator=g))
    gw0 = dev(torch.randn(nf, 1, generator=g))
    part = K.gp_head_wgrad(gg, y[3 * B:], 2.0, w_out)
    acc = K.gp_head_wgrad(gg, y[3 * B:], 2.0, w_out, add_to=gw0.clone())
    assert torch.allclose(acc, gw0 + part, rtol=1e-6, atol=1e-7)
    # (iii)
    ga = torch.randn(B, 3, 32, 32, generator=g).cuda()
    gs = cl(torch.randn(B, 3, 16, 16, generator=g))
    want = ga + K.upsample2(gs, 0.25)
    _, s_ref = K.gp_fwd(want.reshape(B, -1).contiguous(), 10.0, True)
    ga2 = ga.clone()
    slopes = K.gp_finish(ga2, gs, 0.25)
    assert torch.allclose(ga2, want, rtol=1e-6, atol=1e-7) and torch.allclose(slopes, s_ref, rtol=1e-6)


def test_clock_probe_reports_a_plausible_shader_clock_and_never_outlives_its_cap():
    """K.ClockProbe (bench.py's roofline leg): s_memtime cycles / s_memrealtime over a bracketed region on a side stream.  Around real
    launches the clock is a shader clock of this part (0.1 .. 2.5 GHz) and the probe ends on the region's flag; a probe whose region never
    ends (the flag is not set) gives up at its cap instead of hanging."""
    import time
    import ctgan_amd.kernels as K
    from ctgan_amd.kernels import ConvGeom
    g = ConvGeom(128, 16, 16, 128, 3, 3, 1)
    x = K.empty_cl(192, 128, 16, 16, 'cuda').normal_()
    w = torch.randn(3, 3, 128, 128, device='cuda') * 0.05
    K.conv_fwd(x, w, None, g)
    torch.cuda.synchronize()
    with K.ClockProbe() as p:
        for _ in range(8):
            K.conv_fwd(x, w, None, g)
    torch.cuda.synchronize()
    mhz, us, seen = p.result()
    assert seen and 100.0 < mhz < 2500.0 and 20.0 < us < 50000.0, (mhz, us, seen)
    # a region that never signals: the probe returns after its cap (5 ms here), not at the flag
    pr = K.ClockProbe(cap_ms=5.0)
    pr.__enter__()
    t0 = time.time()
    torch.cuda.synchronize()
    assert time.time() - t0 < 2.0
    mhz, us, seen = pr.result()
    assert not seen and 4000.0 < us < 50000.0
