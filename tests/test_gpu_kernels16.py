"""The 16-bit matrix-core conv family (csrc/igemm16.hip: bf16 / fp16 operands, fp32 accumulation) through the C-ABI.

Two kinds of checks per geometry and operator (forward, data gradient, weight gradient):
  * EXACT-PRODUCT check: with operands that are representable in the 16-bit format the products are exact in fp32, so the
    16-bit kernels must agree with the fp64 oracle to fp32 summation error (2e-5) - any indexing / packing / phase /
    transposition mistake shows at full size of the error, rounding plays no part;
  * ROUNDING check: with generic fp32 operands the result must sit within the format's rounding model of the fp64 truth:
    relative L2 error <= 2 * 2^-(mantissa bits + 1) (two rounded operands per product; errors average over K).
BASELINE.json configs[1] (DCGAN 5x5 stride-2 convs / transposed convs) and configs[4] (3x3, 3x3 stride 2, 4x4 stride-2
resampling convs at 64..1024 channels) geometries.
"""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tf_ops  # noqa: E402


@pytest.fixture(scope='module')
def K():
    import ctgan_amd.kernels as K
    yield K
    K.set_mma_dtype(None)


def cl(t):
    d = t.to('cuda')
    out = torch.empty((d.shape[0], d.shape[2], d.shape[3], d.shape[1]), device='cuda', dtype=d.dtype).permute(0, 3, 1, 2)
    out.copy_(d)
    return out


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def q16(t, dt):
    return t.to(torch.bfloat16 if dt == 'bf16' else torch.float16).float()


EPS = {'bf16': 2.0 ** -9, 'f16': 2.0 ** -12}

# (N, C, H, W, K, k, stride)
CASES = [
    (4, 128, 16, 16, 256, 5, 2),      # DCGAN critic layer 2 (:92); its data gradient = Deconv2D 256->128 of the generator (:70)
    (8, 256, 8, 8, 512, 5, 2),        # DCGAN critic layer 3 / Deconv2D 512->256
    (64, 128, 16, 16, 256, 5, 2),     # the same at batch 64: 128x128 tiles
    (4, 64, 16, 16, 128, 3, 1),       # 3x3 stride 1, C = 64
    (3, 32, 8, 8, 32, 3, 1),          # C = 32: the 32-deep K slice variant; K = 32: partial kout tile
    (2, 96, 8, 12, 160, 3, 1),        # C = 96 (32-deep slices), K = 160 (partial tiles), H != W
    (2, 64, 16, 16, 64, 1, 1),        # 1x1 shortcut
    (6, 128, 8, 8, 128, 4, 2),        # 4x4 stride 2 = fused ConvMeanPool / UpsampleConv filter; data gradient: 4 phases of 2x2 taps
    (2, 128, 32, 32, 256, 3, 2),      # 3x3 stride 2, SAME pad (0,1): config[4] critic 'down' conv2 (LS/wgan_LSUN_Bedrooms128.py:126-131)
    (2, 1024, 8, 8, 1024, 3, 1),      # config[4] critic 1024-channel block
    (70, 128, 1, 1, 2048, 1, 1),      # Linear 128 -> 2048 (1x1 conv on a 1x1 image)
]


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'N%d_C%d_H%dx%d_K%d_k%d_s%d' % c)
def test_conv16_fwd_dgrad_wgrad(K, case, dt):
    N, C, H, W, Ko, k, st = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 1000)
    geom = K.ConvGeom(C, H, W, Ko, k, k, st, False)
    for exact in (True, False):
        x = torch.randn(N, C, H, W, generator=g)
        w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
        b = torch.randn(Ko, generator=g)
        gy = torch.randn(N, Ko, geom.P, geom.Q, generator=g)
        if exact:
            x, w, gy = q16(x, dt), q16(w, dt), q16(gy, dt)
        xr = x.double().requires_grad_(True)
        wr = w.double().requires_grad_(True)
        ref = tf_ops.bias_add_nchw(tf_ops.conv2d_same(xr, wr, st), b.double())
        gx_ref, gw_ref = torch.autograd.grad(ref, [xr, wr], gy.double())
        tol = (lambda got, want, what: (relerr(got, want) < 2e-5, relerr(got, want), what)) if exact else \
              (lambda got, want, what: (rel_l2(got, want) < 2 * EPS[dt], rel_l2(got, want), what))
        xd, wd, bd, gyd = cl(x), w.cuda(), b.cuda(), cl(gy)
        with K.mma_dtype(dt):
            y = K.conv_fwd(xd, wd, bd, geom)
            assert K.last_kernel().startswith('conv16'), K.last_kernel()
            ok, e, what = tol(y, ref, 'fwd'); assert ok, (what, e, K.last_kernel())
            # epilogue: relu(conv(relu(x)) + b + resid)
            r = torch.randn(ref.shape, generator=g)
            y2 = K.conv_fwd(xd, wd, bd, geom, resid=cl(r), relu=True, relu_in=True)
            ref2 = torch.relu(tf_ops.bias_add_nchw(tf_ops.conv2d_same(torch.relu(x.double()), w.double(), st), b.double()) + r.double())
            ok, e, what = tol(y2, ref2, 'fwd+epilogue'); assert ok, (what, e)
            gx = K.conv_dgrad(gyd, wd, geom, N)
            assert K.last_kernel().startswith('conv16'), K.last_kernel()
            ok, e, what = tol(gx, gx_ref, 'dgrad'); assert ok, (what, e, K.last_kernel())
            # dgrad epilogue: (conv^T + bias) masked + resid  (bias has the data gradient's channel count)
            bc = torch.randn(C, generator=g); m = torch.randn(N, C, H, W, generator=g); rr = torch.randn(N, C, H, W, generator=g)
            gx2 = K.conv_dgrad(gyd, wd, geom, N, bias=bc.cuda(), mask=cl(m), resid=cl(rr))
            ref_gx2 = torch.where(m.double() > 0, gx_ref + bc.double().view(1, -1, 1, 1), torch.zeros_like(gx_ref)) + rr.double()
            ok, e, what = tol(gx2, ref_gx2, 'dgrad+epilogue'); assert ok, (what, e)
            if C % 64 == 0 and geom.Q % 4 == 0:
                gw, gb = K.conv_wgrad(xd, gyd, geom, with_bias=True)
                ok, e, what = tol(gw, gw_ref, 'wgrad'); assert ok, (what, e)
                assert relerr(gb, gy.double().sum(dim=(0, 2, 3))) < 2e-5
                gw2 = K.conv_wgrad(xd, gyd, geom, relu_x=True)
                xr2 = x.double()
                wr2 = w.double().requires_grad_(True)
                (gw2_ref,) = torch.autograd.grad(tf_ops.conv2d_same(torch.relu(xr2), wr2, st), [wr2], gy.double())
                ok, e, what = tol(gw2, gw2_ref, 'wgrad relu_x'); assert ok, (what, e)


# launches large enough for the 128x128 tiles of the split mode (>= 192 tiles): the halo-patch kernel (stride 1, whole-row tiles) and
# the single-stage slice kernel (stride 2)
X3_CASES = CASES + [
    (24, 32, 32, 32, 128, 3, 1),      # 32-wide images: tile = 4 rows, 6 x 34 patch
    (96, 32, 16, 16, 128, 3, 1),      # 16-wide images: tile = 8 rows (half an image), 10 x 18 patch
    (96, 32, 16, 16, 128, 5, 1),      # 5x5 taps: too large a patch beside an LDS filter stage - the fragment-streaming halo kernel on 64-pixel tiles
    (384, 32, 8, 8, 128, 3, 1),       # 8x8 images: tile = two whole images, each with its own 10 x 10 halo block
    (96, 64, 32, 32, 128, 4, 2),      # stride 2 (the folded ConvMeanPool filter): single-stage 128x128 slice kernel; data gradient in 4 phases
    # launches whose 128-pixel tiles cannot fill the chip: the 64- and 32-pixel tiles of the fragment-streaming halo kernel
    (64, 32, 16, 16, 128, 3, 1),      # 16-wide, 64-pixel tiles = 4 rows (256 tiles)
    (12, 64, 32, 32, 128, 3, 1),      # 32-wide, 64-pixel tiles = 2 rows (192 tiles)
    (48, 32, 8, 8, 128, 3, 1),        # 8x8 images, 32-pixel tiles = half an image (96 tiles)
    (160, 32, 8, 8, 256, 3, 1),       # 8x8 images, 32-pixel tiles = half an image; two kout tiles (640 tiles)
    (192, 128, 8, 8, 128, 3, 1),      # 8x8 images, 32-pixel tiles, forward and data gradient (384 workgroups)
    # the folded ConvMeanPool / UpsampleConv filters on the stride-2 halo kernels (csrc/conv16s2.h): data gradient = four phases from one dy patch
    (128, 128, 32, 32, 128, 4, 2),    # dy 16x16: 64-position tiles (4 rows + halo), 512 workgroups of eight waves
    (128, 128, 16, 16, 128, 4, 2),    # dy 8x8: 32-position tiles (half an image)
    (40, 256, 32, 32, 128, 4, 2),     # two channel tiles on the output side of the data gradient, 4 chunks on its reduction side
]
X3_S2_DGRAD = {(128, 128, 32, 32, 128, 4, 2): 'conv16x3sf<64x128', (128, 128, 16, 16, 128, 4, 2): 'conv16x3p<4x32x128', (40, 256, 32, 32, 128, 4, 2): 'conv16x3sf<64x128'}      # (from 768 workgroups: conv16x3sf, one phase per workgroup)
# the stride-2 forward: conv16x3sf_kernel (filter fragments from L2, round 5) at 64- / 32-position tiles; reductions beyond 2,304 terms keep the slice kernel
X3_S2_FWD = {(96, 64, 32, 32, 128, 4, 2): 'conv16x3sf<64x128', (128, 128, 32, 32, 128, 4, 2): 'conv16x3sf<64x128', (128, 128, 16, 16, 128, 4, 2): 'conv16x3<',
             (40, 256, 32, 32, 128, 4, 2): 'conv16x3<'}
X3_SMALL_TILE = {(64, 32, 16, 16, 128, 3, 1): '64x128', (12, 64, 32, 32, 128, 3, 1): '64x128', (48, 32, 8, 8, 128, 3, 1): '32x128',
                 (160, 32, 8, 8, 256, 3, 1): '32x128', (192, 128, 8, 8, 128, 3, 1): '32x128', (384, 32, 8, 8, 128, 3, 1): '32x128'}
# round 6: 128-channel launches of at most 256 workgroups of 64 pixels x 64 kout run one channel chunk per wave (conv16x3hk_kernel): none of these
# cases (192 rows of 8x8 are 384 workgroups) - test_f32x3_chunk_per_wave_halo_kernel_against_fp64_and_the_pixel_tiled_kernels covers that kernel
X3_CHUNK_PER_WAVE = set()


@pytest.mark.parametrize('case', X3_CASES, ids=lambda c: 'N%d_C%d_H%dx%d_K%d_k%d_s%d' % c)
def test_f32x3_split_mode_is_as_accurate_as_the_fp32_mfma_family(K, case):
    """'f32x3': every fp32 operand split exactly into three bf16 terms, six bf16 MFMAs per product, fp32 accumulation.  With generic
    fp32 operands its distance to the fp64 truth must be fp32 rounding noise - no larger than twice that of the fp32 MFMA family on
    the same inputs (floor 3e-7: both sit at a few 1e-8 .. 1e-7) - forward, data gradient (with epilogues) and weight gradient."""
    N, C, H, W, Ko, k, st = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 1000 + 7)
    geom = K.ConvGeom(C, H, W, Ko, k, k, st, False)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)
    b = torch.randn(Ko, generator=g)
    gy = torch.randn(N, Ko, geom.P, geom.Q, generator=g)
    r = torch.randn(N, Ko, geom.P, geom.Q, generator=g)
    bc = torch.randn(C, generator=g); m = torch.randn(N, C, H, W, generator=g); rr = torch.randn(N, C, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    ref = tf_ops.bias_add_nchw(tf_ops.conv2d_same(xr, wr, st), b.double())
    gx_ref, gw_ref = torch.autograd.grad(ref, [xr, wr], gy.double())
    ref2 = torch.relu(tf_ops.bias_add_nchw(tf_ops.conv2d_same(torch.relu(x.double()), w.double(), st), b.double()) + r.double())
    ref_gx2 = torch.where(m.double() > 0, gx_ref + bc.double().view(1, -1, 1, 1), torch.zeros_like(gx_ref)) + rr.double()
    xd, wd, bd, gyd = cl(x), w.cuda(), b.cuda(), cl(gy)

    def run():
        out = {'fwd': K.conv_fwd(xd, wd, bd, geom)}
        kern = {'fwd': K.last_kernel()}
        out['fwd+epi'] = K.conv_fwd(xd, wd, bd, geom, resid=cl(r), relu=True, relu_in=True)
        out['dgrad'] = K.conv_dgrad(gyd, wd, geom, N)
        kern['dgrad'] = K.last_kernel()
        out['dgrad+epi'] = K.conv_dgrad(gyd, wd, geom, N, bias=bc.cuda(), mask=cl(m), resid=cl(rr))
        out['wgrad'] = K.conv_wgrad(xd, gyd, geom)
        kern['wgrad'] = K.last_kernel()
        return out, kern
    want = {'fwd': ref, 'fwd+epi': ref2, 'dgrad': gx_ref, 'dgrad+epi': ref_gx2, 'wgrad': gw_ref}
    with K.mma_dtype('f32x3'):
        got3, kern3 = run()
    hybrid, K.X3_HYBRID = K.X3_HYBRID, False
    try:
        got1, kern1 = run()                                  # the fp32 MFMA family
    finally:
        K.X3_HYBRID = hybrid
    assert kern1['fwd'].startswith('igemm') or kern1['fwd'].startswith('fewch'), kern1
    assert kern3['fwd'].startswith('conv16x3') and kern3['dgrad'].startswith('conv16x3'), kern3
    if case in X3_CASES[len(CASES):]:
        want_kernel = 'conv16x3h' if st == 1 else 'conv16x3<128x'       # (stride 2: the slice kernel, 128 kout x 128 or 64 pixels)
        if case in X3_SMALL_TILE:
            want_kernel = 'conv16x3hf<' + X3_SMALL_TILE[case]
        if case in X3_CHUNK_PER_WAVE:
            want_kernel = 'conv16x3hk<'
        assert kern3['fwd'].startswith(X3_S2_FWD.get(case, want_kernel)) and (C % 128 != 0 or kern3['dgrad'].startswith(X3_S2_DGRAD.get(case, want_kernel))), kern3
    pq = geom.P * geom.Q
    if C % 128 == 0 and Ko % 128 == 0 and geom.Q % 4 == 0 and not (pq & (pq - 1)) and not (geom.Q & (geom.Q - 1)):
        assert kern3['wgrad'].startswith('wgrad16x3') or kern3['wgrad'].startswith('reduce16'), kern3
    for what in want:
        e3, e1 = rel_l2(got3[what], want[what]), rel_l2(got1[what], want[what])
        assert e3 <= max(2.0 * e1, 3e-7), (what, e3, e1)
        m3, m1 = relerr(got3[what], want[what]), relerr(got1[what], want[what])
        assert m3 <= max(3.0 * m1, 2e-6), (what, m3, m1)      # (floor: 2e-6 of the largest element - a 3,200-term fp32 chain in any order)


@pytest.mark.parametrize('case', [(128, 128, 32, 32, 128), (192, 128, 16, 16, 256), (40, 256, 32, 32, 64), (48, 128, 16, 16, 128)],
                         ids=lambda c: 'N%d_C%d_H%dx%d_K%d' % c)
def test_f32x3_stride2_halo_data_gradient_equals_the_slice_kernel(K, case):
    """conv16x3p_kernel (csrc/conv16s2.h): the four output-parity phases of the folded 4x4 / stride-2 data gradient from ONE staged dy
    patch.  Same products as the slice kernel in another fp32 summation order: the two must agree to 2e-6 of the largest element, with
    and without the bias / mask / residual epilogue, and the plain result must match the fp64 oracle's data gradient."""
    N, C, H, W, Ko = case
    g = torch.Generator().manual_seed(sum(case))
    geom = K.ConvGeom(C, H, W, Ko, 4, 4, 2, False)
    w = (torch.randn(4, 4, C, Ko, generator=g) / np.sqrt(16 * Ko)).cuda()
    gy = cl(torch.randn(N, Ko, geom.P, geom.Q, generator=g))
    bc = torch.randn(C, generator=g).cuda(); m = cl(torch.randn(N, C, H, W, generator=g)); rr = cl(torch.randn(N, C, H, W, generator=g))

    def run():
        a = K.conv_dgrad(gy, w, geom, N); ka = K.last_kernel()
        b = K.conv_dgrad(gy, w, geom, N, bias=bc, mask=m, resid=rr)
        return a, b, ka
    with K.mma_dtype('f32x3'):
        a1, b1, k1 = run()
        K.debug_x3_s2halo(False)
        try:
            a0, b0, k0 = run()
        finally:
            K.debug_x3_s2halo(True)
    assert k1.startswith(('conv16x3p<', 'conv16x3sf<')) and k0.startswith('conv16x3<'), (k1, k0)      # (8x8 dy grids from 768 workgroups: one phase per workgroup on conv16x3sf)
    for x1, x0 in ((a1, a0), (b1, b0)):
        assert float((x1 - x0).abs().max()) <= 2e-6 * float(x0.abs().max())
    xr = torch.zeros(N, C, H, W, dtype=torch.float64, requires_grad=True)
    gx, = torch.autograd.grad(tf_ops.conv2d_same(xr, w.cpu().double(), 2), [xr], gy.cpu().double())
    assert relerr(a1, gx) <= 3e-6


@pytest.mark.parametrize('case', [(192, 128, 32, 32, 128, 4), (320, 128, 32, 32, 128, 4), (96, 128, 32, 32, 128, 4), (384, 128, 16, 16, 128, 4), (96, 128, 16, 16, 256, 2),
                                  (96, 64, 32, 32, 128, 4), (384, 256, 16, 16, 128, 3), (192, 128, 16, 16, 128, 4), (64, 128, 32, 32, 128, 4)], ids=lambda c: 'N%d_C%d_H%dx%d_K%d_k%d' % c)
def test_f32x3_strided_forward_with_filter_fragments_from_l2_equals_the_slice_kernel(K, case):
    """conv16x3sf_kernel (round 5): the stride-2 forward of the split mode - the folded ConvMeanPool / MeanPoolConv filters of
    TF/CT_gan_cifar_resnet.py:89-98 (4x4 / 2x2, stride 2) and a 3x3 stride-2 conv (LS/wgan_LSUN_Bedrooms128.py:113) - with the filter operand
    streamed from L2 in fragment order and the pixel operand alone in LDS, at 128- / 64- / 32-position tiles and (the last two cases: 192 / 256 tiles)
    with the chunks split over two workgroups per tile + the slab epilogue.  Same products as the slice kernel
    in another fp32 summation order: the two agree to 3e-6 of the largest element, plain and with relu-on-load + bias + residual and with the
    out-mask epilogue, and the plain result matches the fp64 oracle."""
    N, C, H, W, Ko, k = case
    g = torch.Generator().manual_seed(sum(case))
    geom = K.ConvGeom(C, H, W, Ko, k, k, 2, False)
    w = (torch.randn(k, k, C, Ko, generator=g) / np.sqrt(k * k * C)).cuda()
    x = cl(torch.randn(N, C, H, W, generator=g))
    b = torch.randn(Ko, generator=g).cuda(); rr = cl(torch.randn(N, Ko, geom.P, geom.Q, generator=g)); mk = cl(torch.randn(N, Ko, geom.P, geom.Q, generator=g))

    def run():
        a = K.conv_fwd(x, w, None, geom); ka = K.last_kernel()
        c = K.conv_fwd(x, w, b, geom, resid=rr, relu_in=True); kc = K.last_kernel()
        d = K.conv_fwd(x, w, b, geom, mask=mk)
        e = K.conv_fwd(x, w, b, geom, relu=True)
        return (a, c, d, e), ka, kc
    with K.mma_dtype('f32x3'):
        r1, k1, k1c = run()
        K.debug_x3_s2fwd(False)
        try:
            r0, k0, _ = run()
        finally:
            K.debug_x3_s2fwd(True)
    assert k1.startswith('conv16x3sf<') and k1c.startswith('conv16x3sf<') and k0.startswith('conv16x3<'), (k1, k1c, k0)
    for x1, x0 in zip(r1, r0):      # (two fp32 summation orders of up to 2,304 terms; the worst of 6 M elements: 2.1e-6 of the largest)
        assert float((x1 - x0).abs().max()) <= 3e-6 * float(x0.abs().max()), k1
    assert float((r1[3] < 0).sum()) == 0
    ref = tf_ops.conv2d_same(x.cpu().double(), w.cpu().double(), 2)
    assert relerr(r1[0], ref) <= 3e-6, k1


@pytest.mark.parametrize('case', [(192, 128, 32, 32, 128), (192, 128, 16, 16, 256), (320, 128, 16, 16, 128)], ids=lambda c: 'N%d_C%d_H%dx%d_K%d' % c)
def test_f32x3_four_phase_data_gradient_one_phase_per_workgroup_equals_the_four_phase_kernel_bitwise(K, case):
    """The data gradient of the folded 4x4 / stride-2 filters on conv16x3sf_kernel (one output-parity phase per workgroup, its 2x2 taps staged
    slice by slice, filter fragments from L2 in the FRAG image's (chunk, phase, tap) order) against conv16x3p_kernel (four phases from one dy
    patch): the same products accumulated in the same order - identical bits, plain and with bias + mask + residual."""
    N, C, H, W, Ko = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    geom = K.ConvGeom(C, H, W, Ko, 4, 4, 2, False)
    w = (torch.randn(4, 4, C, Ko, generator=g) / np.sqrt(16 * Ko)).cuda()
    gy = cl(torch.randn(N, Ko, geom.P, geom.Q, generator=g))
    bc = torch.randn(C, generator=g).cuda(); m = cl(torch.randn(N, C, H, W, generator=g)); rr = cl(torch.randn(N, C, H, W, generator=g))
    out = {}
    with K.mma_dtype('f32x3'):
        try:
            for code in (-1, 0):
                K.lib.ctgan_debug_x3_s2dgrad_sf(code)
                a = K.conv_dgrad(gy, w, geom, N); ka = K.last_kernel()
                out[code] = (a, K.conv_dgrad(gy, w, geom, N, bias=bc, mask=m, resid=rr), ka)
        finally:
            K.lib.ctgan_debug_x3_s2dgrad_sf(0)
    assert out[-1][2].startswith('conv16x3p<') and out[0][2].startswith('conv16x3sf<'), (out[-1][2], out[0][2])
    assert torch.equal(out[-1][0], out[0][0]) and torch.equal(out[-1][1], out[0][1])


@pytest.mark.parametrize('case', [(24, 128, 32, 32, 128, 3, 1), (96, 128, 16, 16, 256, 3, 1), (384, 128, 8, 8, 128, 3, 1)],
                         ids=lambda c: 'N%d_C%d_H%dx%d_K%d_k%d_s%d' % c)
def test_f32x3_halo_kernel_with_filter_fragments_from_l2_equals_the_lds_staged_one_bitwise(K, case):
    """conv16x3hf (filter fragments streamed from L2 in MFMA-fragment order, waves split over kout) performs the same MFMAs in the same
    order per accumulator element as conv16x3h (filter staged through LDS): forward with every epilogue and the data gradient must be
    BIT-identical (whatever the tile height: the order per accumulator element does not depend on it) - any error in the fragment-order packed image or the wave -> channel map shows as a mismatch."""
    N, C, H, W, Ko, k, st = case
    g = torch.Generator().manual_seed(5)
    geom = K.ConvGeom(C, H, W, Ko, k, k, st, False)
    x, w, b = cl(torch.randn(N, C, H, W, generator=g)), (torch.randn(k, k, C, Ko, generator=g) / 17).cuda(), torch.randn(Ko, generator=g).cuda()
    r, gy = cl(torch.randn(N, Ko, H, W, generator=g)), cl(torch.randn(N, Ko, H, W, generator=g))
    m, rr, bc = cl(torch.randn(N, C, H, W, generator=g)), cl(torch.randn(N, C, H, W, generator=g)), torch.randn(C, generator=g).cuda()

    def run(v):
        K.debug_x3_halo_version(v)
        K.debug_x3_hk(0)                  # (this test is about the two pixel-tiled kernels: the chunk-per-wave kernel stays out)
        try:
            with K.mma_dtype('f32x3'):
                a = K.conv_fwd(x, w, b, geom, resid=r, relu=True, relu_in=True)
                name = K.last_kernel()
                c = K.conv_fwd(x, w, None, geom)
                d = K.conv_dgrad(gy, w, geom, N, bias=bc, mask=m, resid=rr)
                return a, c, d, name, K.last_kernel()
        finally:
            K.debug_x3_halo_version(0)
            K.debug_x3_hk(1)
    a1, c1, d1, n1, nd1 = run(1)
    a2, c2, d2, n2, nd2 = run(2)
    assert n1.startswith('conv16x3h<128x128') and nd1.startswith('conv16x3h<128x128'), (n1, nd1)
    assert n2.startswith('conv16x3hf<') and nd2.startswith('conv16x3hf<'), (n2, nd2)       # (its preferred tile height for the image size)
    assert torch.equal(a1, a2) and torch.equal(c1, c2) and torch.equal(d1, d2)


@pytest.mark.parametrize('case', [(64, 8, 128), (192, 8, 128), (256, 8, 128), (384, 8, 128), (64, 16, 128), (128, 8, 256), (32, 16, 128)],
                         ids=lambda c: 'N%d_H%d_K%d' % c)
def test_f32x3_chunk_per_wave_halo_kernel_against_fp64_and_the_pixel_tiled_kernels(K, case):
    """conv16x3hk_kernel (round 6): 64-pixel x 64-kout tiles, every wave one 32-channel chunk of C = 128, the four partial sums added in the
    fixed order chunk 0..3 - the launches that cannot fill the chip with pixel tiles (the 8x8 layers at 64-384 rows, the 16x16 layers at 64
    rows; TF/CT_gan_cifar_resnet.py:109-141,174-178 at the shapes of the critic's blocks 2-4 and of the penalty's double backward).  Forward
    (bias, ReLU on load, residual, ReLU), data gradient (bias, mask, residual), the epilogue dropout with sample ranges and the residual
    through the nearest-2x upsample: against fp64 (the split mode's bound: no worse than 2x the fp32 MFMA family, floor 3e-7 / 2e-6) and
    against the kernels the same launches ran on before (another summation order inside fp32: 2e-6 of the largest element; the same
    Philox draws: identical keep / drop pattern)."""
    N, Hh, Ko = case
    C = 128
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 1000 + 3)
    geom = K.ConvGeom(C, Hh, Hh, Ko, 3, 3, 1, False)
    x = torch.randn(N, C, Hh, Hh, generator=g)
    w = torch.randn(3, 3, C, Ko, generator=g) / np.sqrt(9 * C)
    b = torch.randn(Ko, generator=g)
    r = torch.randn(N, Ko, Hh, Hh, generator=g)
    xd, wd, bd, rd = cl(x), w.cuda(), b.cuda(), cl(r)
    ref_f = tf_ops.bias_add_nchw(tf_ops.conv2d_same(x.double(), w.double(), 1), b.double())
    ref_f2 = torch.relu(tf_ops.bias_add_nchw(tf_ops.conv2d_same(torch.relu(x.double()), w.double(), 1), b.double()) + r.double())
    # the data gradient reduces over the conv's OUTPUT channels: a conv Ko -> 128, so that its reduction side has the 128 channels the kernel takes
    wT = torch.randn(3, 3, Ko, C, generator=g) / np.sqrt(9 * C)          # a conv Ko -> C (C = 128 output channels): its dgrad reduces over C = 128 and yields Ko channels
    geomD = K.ConvGeom(Ko, Hh, Hh, C, 3, 3, 1, False)
    gyD = torch.randn(N, C, Hh, Hh, generator=g)
    xr = torch.zeros(N, Ko, Hh, Hh, dtype=torch.float64, requires_grad=True)
    (gx_ref,) = torch.autograd.grad(tf_ops.conv2d_same(xr, wT.double(), 1), [xr], gyD.double())
    bc = torch.randn(Ko, generator=g); m = torch.randn(N, Ko, Hh, Hh, generator=g); rr = torch.randn(N, Ko, Hh, Hh, generator=g)
    ref_g2 = torch.where(m.double() > 0, gx_ref + bc.double().view(1, -1, 1, 1), torch.zeros_like(gx_ref)) + rr.double()
    ctr = torch.tensor([5], dtype=torch.int64, device='cuda')
    drop = {'ranges': [(N // 2, (0.5, 99, 3, ctr)), (N, (0.8, 99, 4, ctr))]} if (N // 2 * Hh * Hh) % 128 == 0 else (0.5, 99, 3, ctr)
    up = cl(torch.randn(N, Ko, Hh // 2, Hh // 2, generator=g))

    def run():
        out, names = {}, {}
        out['fwd'] = K.conv_fwd(xd, wd, bd, geom); names['fwd'] = K.last_kernel()
        out['fwd+epi'] = K.conv_fwd(xd, wd, bd, geom, resid=rd, relu=True, relu_in=True)
        out['dgrad'] = K.conv_dgrad(cl(gyD), wT.cuda(), geomD, N); names['dgrad'] = K.last_kernel()
        out['dgrad+epi'] = K.conv_dgrad(cl(gyD), wT.cuda(), geomD, N, bias=bc.cuda(), mask=cl(m), resid=cl(rr))
        out['fwd+drop'] = K.conv_fwd(xd, wd, bd, geom, relu=True, relu_in=True, drop=drop); names['drop'] = K.last_kernel()
        out['fwd+up'] = K.conv_fwd(xd, wd, bd, geom, resid=up, resid_up=True); names['up'] = K.last_kernel()
        return out, names
    with K.mma_dtype('f32x3'):
        K.debug_x3_hk(2)
        try:
            got, names = run()
        finally:
            K.debug_x3_hk(1)
        K.debug_x3_hk(0)
        try:
            old, names0 = run()
        finally:
            K.debug_x3_hk(1)
    assert all(v.startswith('conv16x3hk<') for v in names.values()), names
    assert not any(v.startswith('conv16x3hk<') for v in names0.values()), names0
    hybrid, K.X3_HYBRID = K.X3_HYBRID, False
    try:
        f1 = {'fwd': K.conv_fwd(xd, wd, bd, geom), 'fwd+epi': K.conv_fwd(xd, wd, bd, geom, resid=rd, relu=True, relu_in=True),
              'dgrad': K.conv_dgrad(cl(gyD), wT.cuda(), geomD, N),
              'dgrad+epi': K.conv_dgrad(cl(gyD), wT.cuda(), geomD, N, bias=bc.cuda(), mask=cl(m), resid=cl(rr))}      # the fp32 MFMA family
    finally:
        K.X3_HYBRID = hybrid
    want = {'fwd': ref_f, 'fwd+epi': ref_f2, 'dgrad': gx_ref, 'dgrad+epi': ref_g2}
    for what in want:
        e3, e1 = rel_l2(got[what], want[what]), rel_l2(f1[what], want[what])
        assert e3 <= max(2.0 * e1, 3e-7), (what, e3, e1)
        m3, m1 = relerr(got[what], want[what]), relerr(f1[what], want[what])
        assert m3 <= max(3.0 * m1, 2e-6), (what, m3, m1)
    for what in got:
        assert relerr(got[what], old[what]) < 2e-6, (what, relerr(got[what], old[what]), names[what if what in names else 'fwd'], names0)
    assert torch.equal(got['fwd+drop'] == 0, old['fwd+drop'] == 0) or ((got['fwd+drop'] == 0) != (old['fwd+drop'] == 0)).float().mean().item() < 1e-5
    assert 0.2 < (got['fwd+drop'] == 0).float().mean().item() < 0.95


@pytest.mark.parametrize('N,split', [(64, 0), (256, 192), (8, 8)])
def test_chain8x8_equals_the_separate_launches_and_fp64(K, N, split):
    """ctgan_conv2d16_chain8x8 (round 6: one image per workgroup through up to four 3x3 convs on 8 x 8 x 128 images, csrc/chain8x8.hip) with
    both programs the critic step runs: (B) the data-gradient chain of blocks 4-3 - conv^T, mask; conv^T, mask, + residual, x ranged dropout;
    twice - and (C) the penalty's double backward - x dropout, masked; conv, mask; conv + residual, x dropout, masked; conv, mask; conv +
    residual.  Against the same chain as separate launches (conv_dgrad / conv_fwd with their epilogues, dropout_rng_mask: another summation
    order inside fp32, the same Philox draws - identical zero patterns) and, without the dropouts, against fp64."""
    C, H = 128, 8
    g = torch.Generator().manual_seed(N + 3)
    geom = K.ConvGeom(C, H, H, C, 3, 3, 1, False)
    ws = [(torch.randn(3, 3, C, C, generator=g) / np.sqrt(9 * C)).cuda() for _ in range(4)]
    for w in ws:
        K._STABLE_PTRS.add(w.data_ptr())
    x = cl(torch.randn(N, C, H, H, generator=g))
    masks = [cl(torch.randn(N, C, H, H, generator=g)) for _ in range(4)]
    ctr = torch.tensor([7], dtype=torch.int64, device='cuda')
    seed, sA0, sA1, sB0, sB1 = 99, 3, 4, 5, 6
    with K.mma_dtype('f32x3'):
        # ---- (B) the backward chain
        lo = split if split else N
        dr = lambda a, b: {'ranges': [(lo, (0.5, seed, a, ctr)), (N, (0.5, seed, b, ctr))]} if split and split < N else (0.5, seed, a if split else b, ctr)
        got = K.conv_chain8x8(x, [{'save': 1},
                                  {'w': ws[0], 'op': 1, 'mask': masks[0], 'out': True},
                                  {'w': ws[1], 'op': 1, 'mask': masks[1], 'resid': 1, 'drop': 1, 'save': 2, 'out': True},
                                  {'w': ws[2], 'op': 1, 'mask': masks[2], 'out': True},
                                  {'w': ws[3], 'op': 1, 'mask': masks[3], 'resid': 2, 'drop': 2, 'out': True}],
                              drops=[(0.5, sA0, sA1, split if split < N else N), (0.5, sB0, sB1, split if split < N else N)], seed=seed, ctr=ctr)
        assert K.last_kernel().startswith('chain8x8')
        K.debug_x3_hk(0)
        try:
            a1 = K.conv_dgrad(x, ws[0], geom, N, mask=masks[0])
            a2 = K.conv_dgrad(a1, ws[1], geom, N, mask=masks[1], resid=x, drop=dr(sA0, sA1))
            a3 = K.conv_dgrad(a2, ws[2], geom, N, mask=masks[2])
            a4 = K.conv_dgrad(a3, ws[3], geom, N, mask=masks[3], resid=a2, drop=dr(sB0, sB1))
        finally:
            K.debug_x3_hk(1)
        for i, (u, v) in enumerate(zip(got, (a1, a2, a3, a4))):
            assert torch.equal(u == 0, v == 0) or ((u == 0) != (v == 0)).float().mean().item() < 1e-5, i
            assert relerr(u, v) < 4e-6 * (i + 1), (i, relerr(u, v))
        # ---- (C) the double backward (one stream per dropout: n_split = 0)
        gotc = K.conv_chain8x8(x, [{'drop': 1, 'save': 1, 'post_mask': masks[0], 'out': True},
                                   {'w': ws[0], 'op': 0, 'mask': masks[1], 'out': True},
                                   {'w': ws[1], 'op': 0, 'resid': 1, 'drop': 2, 'save': 2, 'post_mask': masks[2], 'out': True},
                                   {'w': ws[2], 'op': 0, 'mask': masks[3], 'out': True},
                                   {'w': ws[3], 'op': 0, 'resid': 2, 'out': True}],
                               drops=[(0.8, sA0, sA0, 0), (0.5, sB0, sB0, 0)], seed=seed, ctr=ctr)
        K.debug_x3_hk(0)
        try:
            r3, u = K.dropout_rng_mask(x, masks[0], 0.8, seed, sA0, ctr, want_dropped=True)
            u1 = K.conv_fwd(u, ws[0], None, geom, mask=masks[1])
            z = K.conv_fwd(u1, ws[1], None, geom, resid=r3)
            r4, u2 = K.dropout_rng_mask(z, masks[2], 0.5, seed, sB0, ctr, want_dropped=True)
            u3 = K.conv_fwd(u2, ws[2], None, geom, mask=masks[3])
            u4 = K.conv_fwd(u3, ws[3], None, geom, resid=r4)
        finally:
            K.debug_x3_hk(1)
        for i, (a, b) in enumerate(zip(gotc, (u, u1, u2, u3, u4))):
            assert torch.equal(a == 0, b == 0) or ((a == 0) != (b == 0)).float().mean().item() < 1e-5, i
            assert relerr(a, b) < 4e-6 * (i + 1), (i, relerr(a, b))
        assert torch.equal(gotc[0], u)                        # the pre step is elementwise: bit for bit
        # ---- fp64: two convs, masks and a residual, no dropout
        got2 = K.conv_chain8x8(x, [{'save': 1}, {'w': ws[0], 'op': 0, 'mask': masks[0], 'out': True}, {'w': ws[1], 'op': 1, 'resid': 1, 'out': True}])
    xd = x.cpu().double()
    y1 = torch.where(masks[0].cpu().double() > 0, tf_ops.conv2d_same(xd, ws[0].cpu().double(), 1), torch.zeros_like(xd))
    yr = y1.clone().requires_grad_(True)
    zin = torch.zeros_like(xd, requires_grad=True)
    (gx,) = torch.autograd.grad(tf_ops.conv2d_same(zin, ws[1].cpu().double(), 1), [zin], y1)       # conv^T(y1, w1)
    assert rel_l2(got2[0], y1) < 3e-7 and rel_l2(got2[1], gx + xd) < 5e-7, (rel_l2(got2[0], y1), rel_l2(got2[1], gx + xd))


def test_f32x3_halo_kernel_adds_the_upsampled_residual(K):
    """UpsampleConv shortcut (CTGAN_RESID_UP): the dense [N, K, P/2, Q/2] residual is added through a nearest-2x upsample inside the
    halo-patch kernel's epilogue - equal to the fp32 family's result on the same inputs to fp32 rounding."""
    N, C, H, Ko = 24, 32, 32, 128
    geom = K.ConvGeom(C, H, H, Ko, 3, 3, 1, False)
    g = torch.Generator().manual_seed(11)
    x, w, b = cl(torch.randn(N, C, H, H, generator=g)), (torch.randn(3, 3, C, Ko, generator=g) / 17).cuda(), torch.randn(Ko, generator=g).cuda()
    r = cl(torch.randn(N, Ko, H // 2, H // 2, generator=g))
    hybrid, K.X3_HYBRID = K.X3_HYBRID, False
    try:
        want = K.conv_fwd(x, w, b, geom, resid=r, resid_up=True, relu_in=True)
    finally:
        K.X3_HYBRID = hybrid
    assert K.last_kernel().startswith('igemm'), K.last_kernel()
    with K.mma_dtype('f32x3'):
        got = K.conv_fwd(x, w, b, geom, resid=r, resid_up=True, relu_in=True)
        assert K.last_kernel().startswith('conv16x3h'), K.last_kernel()
    assert relerr(got, want) < 2e-6


# (N, C, H, K, k, stride): the forward convs of the gradient penalty's double backward (CIFAR ResNet critic, B = 64) + small shapes
MASK_CASES = [
    (64, 3, 32, 128, 3, 1),           # few -> many (fewch f2m)
    (64, 128, 32, 128, 3, 1),         # 32x32: halo-patch split-mode kernel
    (64, 128, 8, 128, 3, 1),          # 8x8: fp32 MFMA family (hybrid routing) / halo-patch small tiles
    (64, 128, 32, 128, 4, 2),         # stride 2 (slice kernel)
    (3, 32, 8, 32, 3, 1),             # partial tiles
    (2, 8, 5, 12, 3, 1),              # generic fp32 kernel (odd extents)
]


@pytest.mark.parametrize('mode', [None, 'f32x3', 'bf16'])
@pytest.mark.parametrize('case', MASK_CASES, ids=lambda c: 'N%d_C%d_H%d_K%d_k%d_s%d' % c)
def test_forward_conv_with_the_out_mask_epilogue_equals_conv_then_relu_mask_bitwise(K, case, mode):
    """ctgan_epilogue_ext.out_mask on every kernel family a forward conv can land on: the result kept where mask > 0 must be
    BIT-identical to the unmasked launch followed by ctgan_lrelu_bwd(., mask, 0) - same kernel, same arithmetic, one pass less."""
    N, C, H, Ko, k, st = case
    geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
    g = torch.Generator().manual_seed(21)
    x = cl(torch.randn(N, C, H, H, generator=g)) if C > 4 else torch.randn(N, C, H, H, generator=g).cuda()
    w, b = (torch.randn(k, k, C, Ko, generator=g) / 9).cuda(), torch.randn(Ko, generator=g).cuda()
    m = cl(torch.randn(N, Ko, geom.P, geom.Q, generator=g))
    ctx = K.mma_dtype(mode) if mode is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        plain = K.conv_fwd(x, w, b, geom, relu_in=C > 4)
        name = K.last_kernel()
        got = K.conv_fwd(x, w, b, geom, relu_in=C > 4, mask=m)
        assert K.last_kernel() == name, (K.last_kernel(), name)       # the mask does not change the routing
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    want = K.lrelu_bwd(plain, m, 0.0)
    assert torch.equal(got, want), name
    assert 0.3 < (got == 0).float().mean().item() < 0.7


def test_dropout_rng_mask_equals_dropout_then_relu_mask_bitwise(K):
    g = torch.Generator().manual_seed(5)
    ctr = torch.tensor([9], dtype=torch.int64, device='cuda')
    for shape in [(64, 128, 8, 8), (3, 5, 7, 3)]:
        x, ref = cl(torch.randn(*shape, generator=g)), cl(torch.randn(*shape, generator=g))
        y0 = K.dropout_rng(x, 0.5, 77, 3, ctr)
        y, ym = K.dropout_rng_mask(x, ref, 0.5, 77, 3, ctr)
        assert torch.equal(y, y0) and torch.equal(ym, K.lrelu_bwd(y0, ref, 0.0))
        none, ym2 = K.dropout_rng_mask(x, ref, 0.5, 77, 3, ctr, want_dropped=False)
        assert none is None and torch.equal(ym2, ym)


@pytest.mark.parametrize('ranged', [False, True])
def test_f32x3_halo_kernel_applies_the_epilogue_dropout_of_the_fp32_family(K, ranged):
    """tf.nn.dropout of the conv result inside the epilogue (one spec, or sample ranges with their own keep / stream as the shared
    forward launches of a critic step use them): the halo-patch kernel must zero exactly the elements the fp32 family zeroes (same
    Philox draws at the same physical offsets) and agree on the kept ones to fp32 rounding."""
    N, C, Hh, Ko = 384, 32, 8, 128                          # 8x8 images: whole-image tiles; 192 tiles
    geom = K.ConvGeom(C, Hh, Hh, Ko, 3, 3, 1, False)
    g = torch.Generator().manual_seed(13)
    x, w, b = cl(torch.randn(N, C, Hh, Hh, generator=g)), (torch.randn(3, 3, C, Ko, generator=g) / 17).cuda(), torch.randn(Ko, generator=g).cuda()
    ctr = torch.tensor([5], dtype=torch.int64, device='cuda')
    if ranged:
        drop = {'ranges': [(128, (0.5, 99, 3, ctr)), (192, None), (384, (0.8, 99, 4, ctr))]}
    else:
        drop = (0.5, 99, 3, ctr)
    hybrid, K.X3_HYBRID = K.X3_HYBRID, False
    try:
        want = K.conv_fwd(x, w, b, geom, relu=True, relu_in=True, drop=drop)
        assert K.last_kernel().startswith('igemm'), K.last_kernel()
    finally:
        K.X3_HYBRID = hybrid
    with K.mma_dtype('f32x3'):
        got = K.conv_fwd(x, w, b, geom, relu=True, relu_in=True, drop=drop)
        assert K.last_kernel().startswith('conv16x3h'), K.last_kernel()
    # kept / dropped pattern: identical wherever the undropped value is not itself ~0 (ReLU zeros are zeros on both sides anyway)
    assert torch.equal(got == 0, want == 0) or ((got == 0) != (want == 0)).float().mean().item() < 1e-5
    assert relerr(got, want) < 2e-6
    assert 0.2 < (want == 0).float().mean().item() < 0.9


def test_f32x3_wgrad_with_the_fused_bias_gradient(K):
    """ctgan_conv2d16_wgrad_bias: db = column sums of dy from the same launch (workgroups of the first tile row), equal to the separate
    column-sum pass to fp32 summation error, for one split and many, and dw unchanged by it (bit-identical to the launch without db)."""
    for N, C, H, Ko, k, st in [(8, 128, 16, 128, 3, 1), (128, 128, 32, 128, 4, 2), (2, 128, 8, 256, 3, 1)]:
        g = torch.Generator().manual_seed(N)
        geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
        x, gy = cl(torch.randn(N, C, H, H, generator=g)), cl(torch.randn(N, Ko, geom.P, geom.Q, generator=g))
        with K.mma_dtype('f32x3'):
            dw0 = K.conv_wgrad(x, gy, geom, relu_x=True)
            assert K.last_kernel().startswith(('wgrad16x3', 'reduce16')), K.last_kernel()
            dw1, db1 = K.conv_wgrad(x, gy, geom, with_bias=True, relu_x=True)
            out = (torch.full_like(dw0, 7.0), torch.full((Ko,), 7.0, device='cuda'))
            dw2, db2 = K.conv_wgrad(x, gy, geom, with_bias=True, relu_x=True, out=out)
        assert torch.equal(dw0, dw1) and torch.equal(dw1, dw2) and torch.equal(db1, db2) and dw2 is out[0]
        ref = gy.double().sum(dim=(0, 2, 3))
        assert relerr(db1, ref) < 2e-6, (N, relerr(db1, ref))


def test_grouped_split_mode_wgrad_equals_the_fp32_family(K):
    """ctgan_conv2d16_wgrad_group (hybrid fp32 mode): the queued weight gradients that fit the 128x128 split-mode tile - several filters,
    geometries, 1-3 segments each with their own relu / bias flags, a finished addend - from ONE grouped launch + one batched reduction
    agree with the fp32 MFMA family to fp32 rounding; a member outside the tile (C = 64) stays on the fp32 family's grouped launch in the
    same call; the result is deterministic."""
    g = torch.Generator().manual_seed(78)
    cases = [(128, 8, 128, 3, 1, (64, 64, 32)), (128, 8, 128, 3, 1, (96,)), (128, 16, 128, 4, 2, (48, 16)), (128, 16, 128, 3, 1, (64, 8)),
             (128, 32, 128, 4, 2, (32, 8)), (64, 8, 96, 1, 1, (5,)), (128, 8, 128, 1, 1, (192, 64)), (256, 8, 128, 3, 1, (7, 3)),
             (128, 4, 256, 3, 1, (16,)), (128, 8, 128, 3, 1, (33,))]
    groups, refs = [], []
    for ci, (C, H, Ko, k, st, Ns) in enumerate(cases):
        geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
        segs = []
        for i, n in enumerate(Ns):
            x = cl(torch.randn(n, C, H, H, generator=g)); gy = cl(torch.randn(n, Ko, geom.P, geom.Q, generator=g))
            segs.append((x, gy, (i + ci) % 2 == 0, i != 1 and ci % 3 != 2))
        has_b = any(sg[3] for sg in segs)
        dw = torch.empty(k, k, C, Ko, device='cuda'); db = torch.empty(Ko, device='cuda') if has_b else None
        dw_r = torch.empty_like(dw); db_r = torch.empty_like(db) if has_b else None
        K.conv_wgrad_multi(segs, geom, dw_r, db_r)
        add = (torch.randn(k, k, C, Ko, generator=g).cuda(), torch.randn(Ko, generator=g).cuda() if has_b else None) if ci % 4 == 1 else (None, None)
        if add[0] is not None:
            dw_r = dw_r + add[0]
            db_r = db_r + add[1] if has_b else None
        groups.append((segs, geom, dw, db, add[0], add[1])); refs.append((dw_r, db_r))
    assert K.X3_WGRAD_GROUP and K.X3_HYBRID
    K.conv_wgrad_group(groups)
    first = [(dw.clone(), None if db is None else db.clone()) for _, _, dw, db, _, _ in groups]
    for (segs, geom, dw, db, _, _), (dw_r, db_r) in zip(groups, refs):
        assert relerr(dw, dw_r) < 3e-6, (geom.C, geom.H, geom.R, relerr(dw, dw_r))
        if db is not None:
            assert relerr(db, db_r) < 3e-6
    K.conv_wgrad_group(groups)
    for (segs, geom, dw, db, _, _), (dw0, db0) in zip(groups, first):
        assert torch.equal(dw, dw0) and (db is None or torch.equal(db, db0))
    with K.mma_dtype('f32x3'):
        K.conv_wgrad_group(groups[:5])
        assert 'wgrad16x3_group' in K.last_kernel(), K.last_kernel()


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
def test_grouped_16bit_wgrad_equals_separate_16bit_launches(K, dt):
    """Mixed-precision modes: the small weight gradients of a step (kernels.grouped16_takes) ride ctgan_conv2d16_wgrad_group with
    mma = bf16 / f16 - same operand rounding, fp32 accumulation, another summation order than the stand-alone launches: equal to
    fp32 summation error."""
    g = torch.Generator().manual_seed(91)
    cases = [(128, 16, 256, 5, 2, (16,)), (256, 8, 512, 5, 2, (16, 16)), (128, 8, 128, 3, 1, (64, 32)), (256, 4, 256, 3, 1, (64,))]
    with K.mma_dtype(dt):
        groups, refs = [], []
        for ci, (C, H, Ko, k, st, Ns) in enumerate(cases):
            geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
            assert all(K.grouped16_takes(geom, n) for n in Ns)
            segs, dw_r, db_r = [], 0, 0
            for i, n in enumerate(Ns):
                x = cl(torch.randn(n, C, H, H, generator=g)); gy = cl(torch.randn(n, Ko, geom.P, geom.Q, generator=g))
                segs.append((x, gy, i % 2 == 0, True))
                w_, b_ = K.conv_wgrad(x, gy, geom, with_bias=True, relu_x=i % 2 == 0)
                assert K.last_kernel().startswith('wgrad16'), K.last_kernel()
                dw_r, db_r = dw_r + w_, db_r + b_
            dw = torch.empty(k, k, C, Ko, device='cuda'); db = torch.empty(Ko, device='cuda')
            groups.append((segs, geom, dw, db)); refs.append((dw_r, db_r))
        K.conv_wgrad_group(groups)
        assert K.last_kernel() == 'wgrad16_group<128x128>', K.last_kernel()
    for (segs, geom, dw, db), (dw_r, db_r) in zip(groups, refs):
        assert relerr(dw, dw_r) < 2e-5 and relerr(db, db_r) < 2e-5, (geom.C, geom.H, relerr(dw, dw_r))


def test_conv16_wgrad_runs_on_the_16bit_kernel_and_is_deterministic(K):
    N, C, H, Ko = 16, 128, 16, 256
    geom = K.ConvGeom(C, H, H, Ko, 5, 5, 2, False)
    g = torch.Generator().manual_seed(5)
    x, gy = cl(torch.randn(N, C, H, H, generator=g)), cl(torch.randn(N, Ko, 8, 8, generator=g))
    with K.mma_dtype('bf16'):
        a = K.conv_wgrad(x, gy, geom)
        assert K.last_kernel().startswith('wgrad16'), K.last_kernel()
        b = K.conv_wgrad(x, gy, geom)
    assert torch.equal(a, b)                                   # fixed-order split-K reduction


def test_packed_filter_cache_follows_the_registry_epoch(K):
    """Parameters are packed once per weight version (tflib.epoch); any other tensor is packed on every call."""
    import ctgan_amd.tflib as lib
    lib.delete_all_params(); lib.set_device(None)
    try:
        g = torch.Generator().manual_seed(1)
        w = lib.param('T.Filters', (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy())
        geom = K.ConvGeom(64, 8, 8, 64, 3, 3, 1, False)
        x = cl(torch.randn(2, 64, 8, 8, generator=g))
        with K.mma_dtype('bf16'):
            y0 = K.conv_fwd(x, w, None, geom)
            with torch.no_grad():
                w.mul_(2.0)                                    # values change WITHOUT an epoch bump: the cached image is used
            y1 = K.conv_fwd(x, w, None, geom)
            assert torch.equal(y0, y1)
            lib.bump_epoch()
            y2 = K.conv_fwd(x, w, None, geom)
            assert relerr(y2, 2 * y0) < 1e-6
            t = w.detach().clone()                             # not a registry parameter: packed per call
            y3 = K.conv_fwd(x, t, None, geom)
            t.mul_(0.5)
            y4 = K.conv_fwd(x, t, None, geom)
            assert relerr(y3, y2) < 1e-6 and relerr(y4, y0) < 1e-6
    finally:
        lib.delete_all_params()


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
def test_dcgan_cifar_step_in_16bit_mode_matches_oracle(K, dt):
    """BASELINE.json configs[1]: CT_gan_cifar.py DCGAN 32x32, BATCH 64, DIM 128, the 5x5 stride-2 convs and transposed convs
    on the 16-bit matrix cores (first / last layer: 3 channels, fp32 few-channel kernels).  One teacher-forced critic step
    and generator step against the fp64 oracle on injected randomness.

    Stated tolerance.  Loss terms: 2e-2 relative for bf16 (8-bit mantissa), 4e-3 for fp16 (measured: 3e-4 .. 1.3e-3 / bf16).
    Parameter gradients: what limits them is not the rounding of the products (~2^-9 per layer) but LeakyReLU SLOPE FLIPS - a
    pre-activation that the 16-bit forward moves across zero switches its slope between 1 and 0.2, an O(1) change of that
    element's gradient; with a fraction f of flipped elements the gradient's relative L2 error is ~sqrt(f) (f ~ 3e-3 for bf16
    => ~5 %).  Measured (gpurun_out/dcgan16_bf16.json): critic gradients 4.3-5.8e-2, generator gradients (critic data
    gradient + generator backward: twice the depth) 5-11.5e-2, cosine similarity with the fp64 gradient >= 0.993.  Bounds:
    relative L2 <= 0.15 and cosine >= 0.99 (bf16), <= 0.05 and >= 0.998 (fp16, 8x finer rounding).  The kernels themselves
    are checked to fp32 summation error on 16-bit-representable operands in test_conv16_fwd_dgrad_wgrad."""
    import ctgan_amd.gan_cifar as M
    import ctgan_amd.tflib as lib
    from ctgan_amd.dcgan_step import DCGANTrainer
    from oracle import nets as onets, steps as osteps, tflib_ref as oref
    B, dim = 64, 128
    ltol, gtol, ctol = (2e-2, 0.15, 0.99) if dt == 'bf16' else (4e-3, 0.05, 0.998)
    report = {}
    lib.delete_all_params(); lib.set_device(None); lib.set_seed(3)
    M.configure(DIM=dim, BATCH_SIZE=B)
    try:
        with torch.no_grad():
            M.Discriminator(M.Generator(2, noise=torch.zeros(2, 128, device='cuda')), u=[torch.ones(2, *s, device='cuda') for s in M.feat_shapes()])
        reg = oref.Registry(dtype=torch.float64)
        for n, p in lib._params.items():
            tr_ = n not in lib._non_trainable
            reg[n] = p.detach().cpu().double().requires_grad_(tr_)
            if not tr_:
                reg.non_trainable.add(n)
        g = torch.Generator().manual_seed(7)
        real_int = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        rnd = osteps.make_rnd_dcgan_d(B, M.feat_shapes(), g)
        G = lambda r, n, z: onets.cifar_generator(r, n, z, DIM=dim)
        D = lambda r, x, u: onets.cifar_discriminator(r, x, u, DIM=dim)
        real = 2. * ((real_int.double() / 255.) - .5)
        ref = osteps.dcgan_d_losses(reg, G, D, real, rnd)
        gref = osteps.grads_of(ref['cost'], reg, 'Discriminator')
        tr = DCGANTrainer(M, seed=1)
        dv = lambda o: [dv(t) for t in o] if isinstance(o, list) else o.float().cuda()
        with K.mma_dtype(dt):
            out = tr.d_step(real_int.cuda(), {k: dv(v) for k, v in rnd.items()})
        for k in ('cost', 'wgan_only', 'ct', 'gp'):
            a, b = out[k].item(), ref[k].item() * (M.cfg.LAMBDA if k == 'gp' else 1.0)     # the product's gp carries LAMBDA
            assert abs(a - b) <= ltol * max(1.0, abs(b)), (k, a, b)
        for k in ('cost', 'wgan_only', 'ct', 'gp'):
            report['d.' + k] = [out[k].item(), ref[k].item() * (M.cfg.LAMBDA if k == 'gp' else 1.0)]
        cos = lambda a, b: torch.nn.functional.cosine_similarity(a.detach().cpu().double().reshape(1, -1), b.detach().double().reshape(1, -1)).item()
        for n in gref:
            if gref[n].abs().max() < 1e-12:
                continue             # the critic's output bias cancels in every loss term: analytically zero gradient
            report['dgrad.' + n] = [rel_l2(out['grads'][n], gref[n]), cos(out['grads'][n], gref[n])]
        for n, v in report.items():
            if n.startswith('dgrad.'):
                assert v[0] <= gtol and v[1] >= ctol, (n, report)
        rg = osteps.make_rnd_dcgan_g(B, M.feat_shapes(), g)
        reg2 = oref.Registry(dtype=torch.float64)              # the generator step starts from the product's updated critic
        for n, p in lib._params.items():
            tr_ = n not in lib._non_trainable
            reg2[n] = p.detach().cpu().double().requires_grad_(tr_)
            if not tr_:
                reg2.non_trainable.add(n)
        refg = osteps.dcgan_g_losses(reg2, G, D, B, rg)
        ggref = osteps.grads_of(refg['cost'], reg2, 'Generator')
        with K.mma_dtype(dt):
            outg = tr.g_step({k: dv(v) for k, v in rg.items()})
        a, b = outg['cost'].item(), refg['cost'].item()
        assert abs(a - b) <= ltol * max(1.0, abs(b)), ('g cost', a, b)
        report['g.cost'] = [a, b]
        for n in ggref:
            if ggref[n].abs().max() < 1e-12:
                continue
            report['ggrad.' + n] = [rel_l2(outg['grads'][n], ggref[n]), cos(outg['grads'][n], ggref[n])]
        import json, os
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump(report, open('gpurun_out/dcgan16_%s.json' % dt, 'w'), indent=1)
        for n, v in report.items():
            if n.startswith('ggrad.'):
                assert v[0] <= gtol and v[1] >= ctol, (n, report)
    finally:
        lib.delete_all_params(); M.configure()


@pytest.mark.parametrize('mode', ['f32x3', 'bf16'])
def test_batched_filter_pack_equals_the_per_filter_pack_bitwise(K, mode):
    """ctgan_conv2d16_pack_batch (every packed image of a weight version in one launch; its forward images go through 32 x 32 LDS tiles
    since round 5: the image is the transpose of the HWIO filter) against ctgan_conv2d16_pack_filter (one launch per filter and
    operator, element by element): forward and data-gradient images, 3x3 / 4x4 stride-2 / 1x1 filters, channel counts with and
    without whole tiles - bit for bit, the fragment-order copy of the split mode included."""
    import ctgan_amd.tflib as lib
    lib.delete_all_params(); lib.set_device(None)
    try:
        g = torch.Generator().manual_seed(3)
        shapes = [('A', 3, 128, 128, 1, 8), ('B', 4, 128, 128, 2, 16), ('C', 1, 64, 64, 1, 8), ('D', 3, 96, 160, 1, 8), ('E', 3, 32, 32, 1, 8)]
        ws, ents = {}, {}
        with K.mma_dtype(mode):
            for name, k, C, Ko, st, H in shapes:
                w = lib.param(name + '.Filters', (torch.randn(k, k, C, Ko, generator=g) * 0.05).numpy())
                ws[name] = w
                geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
                x = cl(torch.randn(2, C, H, H, generator=g)); gy = cl(torch.randn(2, Ko, geom.P, geom.Q, generator=g))
                K.conv_fwd(x, w, None, geom); K.conv_dgrad(gy, w, geom, 2)          # lazily packed, one pack_filter launch per image
            torch.cuda.synchronize()
            single = {key: ent[0].clone() for key, ent in K._pack16.items()}
            assert len(single) >= 2 * len(shapes) - 2, len(single)
            with torch.no_grad():
                for w in ws.values():
                    w.mul_(-0.75)
            lib.bump_epoch()
            for key, ent in K._pack16.items():                                          # per-filter packs of the NEW values: the reference
                d = ent[3].desc(1, (0, 0, 0, 0), (0, 0, 0, 0))
                from ctgan_amd._lib import I64x4, check, lib as clib
                import ctypes
                d.xs = I64x4(4, 1, 4, 4); d.ys = I64x4(4, 1, 4, 4)
                ref = torch.empty_like(ent[0])
                check(clib.ctgan_conv2d16_pack_filter(ctypes.byref(d), ent[4], K._MMA_CODE[mode], ctypes.c_void_p(ent[2].data_ptr()),
                                                      ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'pack')
                ents[key] = ref
            K.prepare_packs()                                                           # ... and the batched launch
            torch.cuda.synchronize()
            for key, ent in K._pack16.items():
                assert not torch.equal(ent[0], single[key]), key
                assert torch.equal(ent[0], ents[key]), key
    finally:
        K.set_mma_dtype(None)
        lib.delete_all_params()


# (N, C, H, W, K, k, stride): the DCGAN critics' layers at the hand-scheduled step's 4B = 256 rows (128x64 tiles; 64x64 tiles + K split),
# the im2col'd first conv as a 1x1 (96-channel slices), a partial kout tile, and a small batch with an odd range boundary
ACT_CASES = [
    (256, 128, 16, 16, 256, 5, 2),
    (256, 256, 8, 8, 512, 5, 2),
    (64, 96, 16, 16, 128, 1, 1),
    (12, 64, 16, 16, 96, 3, 1),
    (20, 64, 16, 16, 128, 5, 2),
]


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('case', ACT_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_fused_lrelu_dropout_epilogue_equals_conv_then_its_own_launch_bitwise(K, case, dt):
    """ctgan_epilogue_ext.act (csrc/igemm16.hip conv16_act): dropout(LeakyReLU(conv + bias)) of TF/CT_gan_cifar.py:86-98 inside the slice
    kernels' epilogue - forward (ref = the result itself; one stream and two sample ranges) and, with ref = a forward result, on the data
    gradient (the pair's backward) - against the plain conv followed by ctgan_lrelu_dropout_rng(2): same bits, same draws."""
    from ctgan_amd.kernels import ConvGeom
    N, C, H, W, Kc, k, st = case
    K.set_mma_dtype(dt)
    try:
        gen = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
        g = ConvGeom(C, H, W, Kc, k, k, st)
        x = cl(torch.randn(N, C, H, W, generator=gen))
        w = (torch.randn(k, k, C, Kc, generator=gen) / (k * (C ** 0.5))).cuda()
        b = (0.1 * torch.randn(Kc, generator=gen)).cuda()
        ctr = torch.full((1,), 11, dtype=torch.int64, device='cuda')
        n1 = (3 * N) // 4
        one = (0.5, 1234, 5, ctr)
        two = {'ranges': [(n1, (0.5, 1234, 5, ctr)), (N, (0.5, 1234, 9, ctr))]}
        assert K.ACT_EPILOGUE
        for drop in (one, two):
            act = {'alpha': 0.2, 'ref': None, 'drop': drop}
            y = K.conv_fwd(x, w, b, g, act=act)
            fused_kernel = K.last_kernel()
            K.ACT_EPILOGUE = False
            try:
                y0 = K.conv_fwd(x, w, b, g, act=act)
            finally:
                K.ACT_EPILOGUE = True
            assert 'conv16' in fused_kernel, fused_kernel
            assert torch.equal(y, y0), (fused_kernel, (y != y0).float().mean().item())
            assert 0.3 < (y == 0).float().mean().item() < 0.7
            if drop is two:      # each range indexed from its own first element: the rows of the second range are a launch of their own on stream 9
                c = K.conv_fwd(x, w, b, g)
                assert torch.equal(y[n1:], K.lrelu_dropout_rng(c[n1:], c[n1:], 0.2, 0.5, 1234, 9, ctr))
            # data gradient of the same layer with the pair's backward: ref = a tensor of dx's shape
            gy = cl(torch.randn(N, Kc, g.P, g.Q, generator=gen))
            ref = cl(torch.randn(N, C, H, W, generator=gen))
            actb = {'alpha': 0.2, 'ref': ref, 'drop': drop}
            dx = K.conv_dgrad(gy, w, g, N, act=actb)
            K.ACT_EPILOGUE = False
            try:
                dx0 = K.conv_dgrad(gy, w, g, N, act=actb)
            finally:
                K.ACT_EPILOGUE = True
            assert torch.equal(dx, dx0), (dx != dx0).float().mean().item()
            plain = K.conv_dgrad(gy, w, g, N)
            kept = dx != 0
            slope = torch.where(ref > 0, torch.ones_like(ref), torch.full_like(ref, 0.2))
            assert torch.equal(dx[kept], (plain * slope * 2.0)[kept])
    finally:
        K.set_mma_dtype(None)
