"""The filter-column weight-gradient kernel (csrc/wgrad16c.hip) through the C-ABI's grouped call, DIRECTLY against the fp64 oracle:
dW = d<conv2d_same(relu?(x), w), dy>/dw by torch autograd in fp64 on `oracle.tf_ops.conv2d_same` (the restatement of tf.nn.conv2d under
TF/tflib/ops/conv2d.py:106-112; the gradient is what compute_gradients builds, TF/CT_gan_cifar_resnet.py:335-336) - not against another
HIP kernel.  The job table is the one the ResNet critic step queues at full width (DIM 128, B 64; tools/wgrad_group_bench.py prints it):
eight filters, two uses each (main pass rows + gradient-penalty rows), relu-on-load / bias flags as the step sets them."""
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


def oracle_wgrad(segs, geom):
    """fp64: sum over the uses of d<conv2d_same(relu?(x), w), dy>/dw, and the bias gradient of the uses that carry one."""
    dw, db = 0, 0
    for x, gy, relu_x, with_bias in segs:
        xd, gd = x.detach().cpu().double(), gy.detach().cpu().double()
        if relu_x:
            xd = xd.clamp_min(0)
        w = torch.zeros(geom.R, geom.S, geom.C, geom.K, dtype=torch.float64, requires_grad=True)
        (g,) = torch.autograd.grad(tf_ops.conv2d_same(xd, w, geom.stride), w, gd)
        dw = dw + g
        if with_bias:
            db = db + gd.sum(dim=(0, 2, 3))
    return dw, db


# (C, H, K, k, stride, rows per use, relu flags, bias flags) - the critic step's queue at DIM 128, B 64 (blocks 3-4: four 8x8 filters over
# 3B = 192 main-pass rows + B = 64 penalty rows; block 2: the folded 2x2 / 4x4 stride-2 filters and the 3x3 on 16x16 over 2B + B; block 1:
# the folded 4x4 stride-2 ConvMeanPool filter on 32x32)
D_STEP = [(128, 8, 128, 3, 1, (192, 64), (1, 0), (1, 0))] * 4 + [
    (128, 16, 128, 2, 2, (128, 64), (0, 0), (1, 0)),
    (128, 16, 128, 4, 2, (128, 64), (1, 0), (1, 0)),
    (128, 16, 128, 3, 1, (128, 64), (1, 0), (1, 0)),
    (128, 32, 128, 4, 2, (128, 64), (1, 0), (1, 0))]


def build(K, table, gen, scale_rows=1):
    groups, oracle = [], []
    for C, H, Ko, k, st, Ns, relus, biases in table:
        geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
        segs = []
        for n, r, b in zip(Ns, relus, biases):
            n = max(1, n // scale_rows)
            x = cl(torch.randn(n, C, H, H, generator=gen))
            gy = cl(torch.randn(n, Ko, geom.P, geom.Q, generator=gen))
            segs.append((x, gy, bool(r), bool(b)))
        has_b = any(sg[3] for sg in segs)
        dw = torch.full((k, k, C, Ko), 7.0, device='cuda')
        db = torch.full((Ko,), 7.0, device='cuda') if has_b else None
        groups.append((segs, geom, dw, db))
    return groups


def test_grouped_launch_on_the_critic_step_job_table_matches_the_fp64_oracle(K):
    assert K.X3_WGRAD_GROUP and K.X3_HYBRID and K.MMA_DTYPE is None
    gen = torch.Generator().manual_seed(404)
    groups = build(K, D_STEP, gen)
    K.conv_wgrad_group(groups)
    assert K.last_kernel().startswith(('wgrad16x3_group', 'reduce16')), K.last_kernel()
    first = [(dw.clone(), db.clone()) for _, _, dw, db in groups]
    for (segs, geom, dw, db) in groups:
        dw_r, db_r = oracle_wgrad(segs, geom)
        e = relerr(dw, dw_r)
        assert e <= 5e-6, ('dw', geom.C, geom.H, geom.R, geom.stride, e)
        eb = relerr(db, db_r)
        assert eb <= 5e-6, ('db', geom.H, geom.R, eb)
    # deterministic: fixed-order slab reduction, no atomics
    K.conv_wgrad_group(groups)
    for (_, _, dw, db), (dw0, db0) in zip(groups, first):
        assert torch.equal(dw, dw0) and torch.equal(db, db0)


def test_the_column_kernel_takes_the_critic_step_and_reports_its_symbol(K):
    """The grouped call must actually ride the filter-column kernel on these shapes (not silently the slice kernel)."""
    gen = torch.Generator().manual_seed(5)
    groups = build(K, D_STEP[4:7], gen, scale_rows=8)
    K.conv_wgrad_group(groups)
    kinds = K.debug_last_wgrad_group_kinds()
    assert kinds & 1, 'no problem of the call rode the column kernel'
    assert not (kinds & 2), 'a problem the column kernel takes fell back to the slice kernel'


@pytest.mark.parametrize('case', [
    # ragged / edge shapes: one row, odd row counts (chunks that end inside an image), three uses, 5x5 on 8x8 (two columns per s),
    # 1x1 (one tap per column), K = 256 (two kout blocks), C = 256 (two channel blocks), 32x32 stride 1 (one row per slice)
    (128, 8, 128, 3, 1, (1,), (1,), (1,)),
    (128, 8, 128, 3, 1, (33, 7, 2), (0, 1, 0), (1, 0, 1)),
    (128, 16, 128, 4, 2, (5, 3), (1, 0), (0, 0)),
    (128, 8, 128, 5, 1, (9,), (0,), (1,)),
    (128, 16, 128, 1, 1, (6,), (0,), (1,)),
    (128, 8, 256, 3, 1, (10,), (1,), (1,)),
    (256, 16, 128, 3, 1, (4,), (0,), (0,)),
    (128, 32, 128, 3, 1, (3, 2), (0, 1), (1, 1)),
    (128, 32, 128, 2, 2, (4,), (1,), (1,)),
    (128, 64, 128, 4, 2, (2,), (0,), (1,)),
])
def test_column_kernel_edge_shapes_match_the_fp64_oracle(K, case):
    gen = torch.Generator().manual_seed(sum(case[5]) + case[1] + case[3])
    # a second (plain 8x8) filter keeps the call on the grouped path when the case has one filter only
    table = [case, (128, 8, 128, 3, 1, (2,), (0,), (0,))]
    groups = build(K, table, gen)
    K.conv_wgrad_group(groups)
    for (segs, geom, dw, db) in groups:
        dw_r, db_r = oracle_wgrad(segs, geom)
        assert relerr(dw, dw_r) <= 5e-6, (case, relerr(dw, dw_r))
        if db is not None:
            assert relerr(db, db_r) <= 5e-6, (case, relerr(db, db_r))


def test_addend_and_zero_bias_row(K):
    """A finished addend aliasing dw / db (the in-place accumulation of the hybrid routing) and a filter whose bias buffer exists while
    only its second use contributes to it."""
    gen = torch.Generator().manual_seed(77)
    table = [(128, 8, 128, 3, 1, (16, 8), (1, 0), (0, 1)), (128, 16, 128, 4, 2, (8,), (0,), (1,))]
    groups = build(K, table, gen)
    segs, geom, dw, db = groups[0]
    add_w, add_b = torch.randn(dw.shape, generator=gen).cuda(), torch.randn(db.shape, generator=gen).cuda()
    dw.copy_(add_w); db.copy_(add_b)
    groups[0] = (segs, geom, dw, db, dw, db)
    K.conv_wgrad_group(groups)
    dw_r, db_r = oracle_wgrad(segs, geom)
    assert relerr(dw, dw_r + add_w.cpu().double()) <= 5e-6
    assert relerr(db, db_r + add_b.cpu().double()) <= 5e-6


def _q16(t, dt):
    return t.to(torch.bfloat16 if dt == 'bf16' else torch.float16).float()


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('case', [
    # (N, C, H, K, k, stride): 64-pixel rows (one-plane modes only: 64-pixel slices), the DCGAN 5x5 stride-2 layers (polyphase columns of
    # three and two taps), config[4]'s 3x3 stride-2 down convs (pad 0 / 1) and its 1024-channel 8x8 layers
    (6, 128, 64, 128, 3, 1), (16, 128, 16, 256, 5, 2), (8, 256, 16, 512, 5, 2), (5, 128, 32, 256, 3, 2), (3, 1024, 8, 1024, 3, 1),
    (7, 128, 32, 128, 4, 2),
])
def test_column_kernel_in_the_16bit_modes_is_exact_on_representable_operands(K, dt, case):
    """bf16 / fp16 modes (BASELINE configs[1] / [4]): one rounded plane per operand, 64-pixel slices.  With operands that are representable in
    the 16-bit format every product is exact in fp32, so the kernel must agree with the fp64 oracle to fp32 summation error - any
    indexing / ring / polyphase / transposition mistake shows at full size.  Through the single-problem entry (ctgan_conv2d16_wgrad_bias),
    which routes to the filter-column kernel."""
    N, C, H, Ko, k, st = case
    gen = torch.Generator().manual_seed(N * 7 + H)
    geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
    x = cl(_q16(torch.randn(N, C, H, H, generator=gen), dt))
    gy = cl(_q16(torch.randn(N, Ko, geom.P, geom.Q, generator=gen), dt))
    with K.mma_dtype(dt):
        dw, db = K.conv_wgrad(x, gy, geom, with_bias=True, relu_x=True)
        assert K.last_kernel() == 'wgrad16_group<col>', K.last_kernel()
        dw2 = K.conv_wgrad(x, gy, geom, relu_x=True)
    dw_r, db_r = oracle_wgrad([(x, gy, True, True)], geom)
    assert relerr(dw, dw_r) <= 2e-5, (case, dt, relerr(dw, dw_r))
    assert relerr(db, db_r) <= 2e-5
    assert torch.equal(dw, dw2)                       # the bias row changes nothing else; deterministic
