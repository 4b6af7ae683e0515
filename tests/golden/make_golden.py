#!/usr/bin/env python
"""Regenerates the golden fixtures in this directory from the CPU oracle (fp64).

The reference (Python-2 / TensorFlow-1.2.1) cannot be imported or run in the build container, so
these vectors are produced by our own restatement of its graph (oracle/) - they pin the oracle
against silent drift and give the GPU suite fixed inputs/outputs that do not depend on the oracle
being importable.  "Parity unpinned" by the reference itself: see oracle/__init__.py.

  python tests/golden/make_golden.py        # rewrites *.npz next to this file
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import nets, steps, tf_ops  # noqa: E402
from oracle import tflib_ref as ops  # noqa: E402

F64 = torch.float64


def npy(t):
    return t.detach().cpu().numpy()


def ops_fixture():
    g = torch.Generator().manual_seed(1)
    out = {}
    # Conv2D: (N,C,H,W,K,k,stride) incl. odd sizes / asymmetric SAME pads
    for i, (N, C, H, W, K, k, s) in enumerate([(2, 3, 8, 8, 8, 3, 1), (2, 8, 8, 8, 8, 1, 1), (2, 3, 8, 8, 8, 5, 2),
                                               (2, 4, 7, 7, 6, 5, 2), (1, 4, 6, 5, 4, 3, 2), (2, 32, 8, 8, 32, 3, 1)]):
        x = torch.randn(N, C, H, W, generator=g, dtype=F64).requires_grad_(True)
        w = (torch.randn(k, k, C, K, generator=g, dtype=F64) / np.sqrt(k * k * C)).requires_grad_(True)
        b = torch.randn(K, generator=g, dtype=F64)
        y = tf_ops.bias_add_nchw(tf_ops.conv2d_same(x, w, s), b)
        gy = torch.randn(y.shape, generator=g, dtype=F64)
        gx, gw = torch.autograd.grad(y, [x, w], gy)
        out.update({'conv%d_cfg' % i: np.array([N, C, H, W, K, k, s]), 'conv%d_x' % i: npy(x), 'conv%d_w' % i: npy(w),
                    'conv%d_b' % i: npy(b), 'conv%d_y' % i: npy(y), 'conv%d_gy' % i: npy(gy), 'conv%d_gx' % i: npy(gx),
                    'conv%d_gw' % i: npy(gw)})
    # Deconv2D (k=5, stride 2) on H in {4,7,8}
    for i, (N, Ci, Co, H) in enumerate([(2, 8, 4, 4), (1, 4, 4, 7), (2, 4, 3, 8)]):
        x = torch.randn(N, Ci, H, H, generator=g, dtype=F64)
        w = torch.randn(5, 5, Co, Ci, generator=g, dtype=F64) / np.sqrt(25 * Ci / 4)
        b = torch.randn(Co, generator=g, dtype=F64)
        y = tf_ops.bias_add_nchw(tf_ops.conv2d_transpose_same(x, w, 2), b)
        out.update({'deconv%d_x' % i: npy(x), 'deconv%d_w' % i: npy(w), 'deconv%d_b' % i: npy(b), 'deconv%d_y' % i: npy(y)})
    # elementwise
    x = torch.randn(3, 8, 4, 4, generator=g, dtype=F64)
    u = torch.rand(3, 8, 4, 4, generator=g, dtype=torch.float32).to(F64)
    out.update(ew_x=npy(x), ew_u=npy(u), ew_drop08=npy(tf_ops.dropout(x, 0.8, u)), ew_drop05=npy(tf_ops.dropout(x, 0.5, u)),
               ew_lrelu=npy(tf_ops.leaky_relu(x)), ew_pool=npy(tf_ops.mean_pool2(x)), ew_up=npy(tf_ops.upsample2(x)))
    # batch norm: fused, axes [0], conditional
    reg = ops.Registry(dtype=F64)
    xb = torch.randn(6, 8, 4, 4, generator=g, dtype=F64) * 2 + 1
    lab = torch.randint(0, 10, (6,), generator=g, dtype=torch.int32)
    reg['cbn.scale'] = torch.rand(10, 8, generator=g, dtype=F64) + .5
    reg['cbn.offset'] = torch.randn(10, 8, generator=g, dtype=F64)
    reg['bn.scale'] = torch.rand(8, generator=g, dtype=F64) + .5
    reg['bn.offset'] = torch.randn(8, generator=g, dtype=F64)
    reg['bn.moving_mean'] = torch.zeros(8, dtype=F64); reg['bn.moving_variance'] = torch.ones(8, dtype=F64)
    x2 = torch.randn(6, 40, generator=g, dtype=F64) + 2
    reg['bn0.scale'] = torch.rand(1, 40, generator=g, dtype=F64) + .5
    reg['bn0.offset'] = torch.randn(1, 40, generator=g, dtype=F64)
    out.update(bn_x=npy(xb), bn_labels=npy(lab), cbn_scale=npy(reg['cbn.scale']), cbn_offset=npy(reg['cbn.offset']),
               cbn_y=npy(ops.CondBatchnorm(reg, 'cbn', [0, 2, 3], xb, labels=lab, n_labels=10)),
               bn_scale=npy(reg['bn.scale']), bn_offset=npy(reg['bn.offset']), bn_y=npy(ops.Batchnorm(reg, 'bn', [0, 2, 3], xb)),
               bn0_x=npy(x2), bn0_scale=npy(reg['bn0.scale']), bn0_offset=npy(reg['bn0.offset']),
               bn0_y=npy(ops.Batchnorm(reg, 'bn0', [0], x2)))
    # loss heads
    d, d_ = torch.randn(8, generator=g, dtype=F64), torch.randn(8, generator=g, dtype=F64)
    f, f_ = torch.randn(8, 16, generator=g, dtype=F64), torch.randn(8, 16, generator=g, dtype=F64)
    gr = torch.randn(8, 48, generator=g, dtype=F64) * 0.2
    s = gr.norm(dim=1)
    out.update(ct_d=npy(d), ct_d_=npy(d_), ct_f=npy(f), ct_f_=npy(f_), ct_M0=npy(steps.ct_term(d, d_, f, f_, 2.0, 0.0)),
               ct_M05=npy(steps.ct_term(d, d_, f, f_, 2.0, 0.5)), gp_g=npy(gr), gp_val=npy(10 * ((s - 1) ** 2).mean()))
    # TF Adam, 3 steps with changing lr
    th = torch.randn(50, generator=g, dtype=F64); m = torch.zeros(50, dtype=F64); v = torch.zeros(50, dtype=F64)
    grads, thetas = [], [npy(th)]
    for t in range(1, 4):
        gg = torch.randn(50, generator=g, dtype=F64)
        th, m, v = tf_ops.tf_adam_step(th, gg, m, v, t, 2e-4 * (1 - t / 10.), 0.5, 0.9)
        grads.append(npy(gg)); thetas.append(npy(th))
    out.update(adam_g=np.stack(grads), adam_theta=np.stack(thetas))
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)


def resnet_fixture(dim=8, B=4, iters=2):
    """Reduced-width ResNet CT-WGAN: initial weights, every random draw, and the loss / gradient-norm
    trace of `iters` x (D step, G step) with teacher = the oracle itself (free running)."""
    reg = ops.Registry(dtype=F64, seed=11)
    cfg = nets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    lab0 = torch.zeros(2, dtype=torch.int32)
    nets.resnet_discriminator(reg, cfg, nets.resnet_generator(reg, cfg, 2, lab0, torch.zeros(2, 128, dtype=F64)), lab0, 1., 1., 1.)
    out = {'w.' + n: npy(t).astype(np.float32) for n, t in reg.items()}
    for n, t in reg.items():                       # fp32-representable initial weights on both sides
        with torch.no_grad():
            t.copy_(t.float().double())
    g = torch.Generator().manual_seed(21)
    optD = steps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    optG = steps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
    for it in range(iters):
        real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
        rnd = steps.make_rnd_resnet_d(B, dim, g)
        o = steps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=it, B=B)
        pre = 'it%d.' % it
        out[pre + 'real'] = npy(real); out[pre + 'labels'] = npy(labels)
        for k, v in rnd.items():
            if isinstance(v, list):
                for j, t in enumerate(v):
                    out[pre + 'd.%s.%d' % (k, j)] = npy(t).astype(np.float32)
            else:
                out[pre + 'd.' + k] = npy(v).astype(np.float32)
        for k in ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only', 'acc_real', 'acc_fake'):
            out[pre + 'd.out.' + k] = npy(o[k])
        out[pre + 'd.out.fake'] = npy(o['fake']).astype(np.float32)
        out[pre + 'd.out.slopes'] = npy(o['slopes'])
        out[pre + 'd.gradnorm'] = np.array([o['grads'][n].norm().item() for n in optD.names])
        rg = steps.make_rnd_resnet_g(B, dim, g)
        o = steps.resnet_g_step(reg, cfg, optG, rg, iteration=it + 1, B=B)
        for j in range(2):
            out[pre + 'g.z.%d' % j] = npy(rg['z'][j]).astype(np.float32)
            out[pre + 'g.label_u.%d' % j] = npy(rg['label_u'][j]).astype(np.float32)
            for i3 in range(3):
                out[pre + 'g.u.%d.%d' % (j, i3)] = npy(rg['u'][j][i3]).astype(np.float32)
        out[pre + 'g.out.cost'] = npy(o['cost'])
        out[pre + 'g.gradnorm'] = np.array([o['grads'][n].norm().item() for n in optG.names])
    out['d_names'] = np.array(optD.names); out['g_names'] = np.array(optG.names)
    out['cfg'] = np.array([dim, B, iters])
    np.savez_compressed(os.path.join(HERE, 'resnet_trace.npz'), **out)


def loop_trace_fixture(dim=32, B=8, iters=1000, seed=2024, init_seed=0, fname='resnet_loop_trace.npz', dtype=F64):
    """Free-running trace of the reference LOOP (TF/CT_gan_cifar_resnet.py:393-404: [G step if it > 0] + 5 x (batch, D
    step), LR decay) over `iters` iterations in fp64.  Seeds, not tensors: the initial weights are the registry's per-name
    init streams (`init_seed`; the product's lib.set_seed draws the same values), the 16 cycled synthetic batches are
    numpy default_rng(1234) exactly as bench.py builds them, and every random draw is a Philox stream of (seed, step)
    (oracle/philox.py, oracle/loop.py).  Stored: every loss term of every critic step and every generator cost."""
    from oracle import loop
    reg = ops.Registry(dtype=dtype, seed=init_seed)
    cfg = nets.ResnetCfg(DIM_G=dim, DIM_D=dim)
    lab0 = torch.zeros(2, dtype=torch.int32)
    nets.resnet_discriminator(reg, cfg, nets.resnet_generator(reg, cfg, 2, lab0, torch.zeros(2, 128, dtype=dtype)), lab0, 1., 1., 1.)
    nrng = np.random.default_rng(1234)
    batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)),
                torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32))) for _ in range(16)]
    cur = [0]

    def next_batch():
        cur[0] = (cur[0] + 1) % len(batches)
        return batches[cur[0]]
    d_recs, g_recs, _, _, pos = loop.resnet_train_loop(reg, cfg, next_batch, iters, B, seed, start_iteration=1, dtype=dtype)
    keys = ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only')
    np.savez_compressed(os.path.join(HERE, fname), cfg=np.array([dim, B, iters, seed, init_seed]), keys=np.array(keys),
                        d=np.array([[r[k] for k in keys] for r in d_recs]), g=np.array(g_recs), stream_pos=np.array(pos))


def _sample_index(name, numel, k):
    """Fixed pseudo-random positions of a parameter's gradient (by name: the GPU test draws the same ones)."""
    import zlib
    if numel <= k:
        return np.arange(numel)
    return np.sort(np.random.default_rng([7, zlib.crc32(name.encode())]).choice(numel, size=k, replace=False))


def dstep_b64_fixture(which, B=64, chunk=8, init_seed=9, data_seed=3, nsample=1024):
    """One critic step of config[4] (`which` = 'lsun128': LS/wgan_LSUN_Bedrooms128.py:137-296 at the reference widths, critic 47.7 M
    parameters) or config[1] ('cifar': TF/CT_gan_cifar.py:58-154, DIM 128) at the FULL batch B = 64 in fp64 (the GPU suite
    compared these configs with the oracle at B <= 16 only).  Seeds, not tensors: initial weights = the registry's per-name init streams
    (`init_seed`; lib.set_seed draws the same values - checked by weight checksums in the fixture), inputs and every random draw from
    torch.Generator(data_seed) exactly as tests/test_lsun128.py / test_gpu_dcgan_step.py build them.  Stored: the loss terms, every
    critic parameter gradient's norm and `nsample` fixed entries of it (all of it for small parameters).

    The fp64 graph of the 128x128 critic at B = 64 (three passes + the penalty's double backward) does not fit this container's
    memory, and it does not have to: every loss term is a batch MEAN of per-sample terms and the critic couples no samples (Layernorm is
    per sample; the generator's BatchNorm does, but its forward is evaluated once, for the whole batch, without a graph), so the step is
    evaluated on `chunk` rows at a time through the SAME oracle function (steps.dcgan_d_losses, with the generator replaced by the
    rows of the precomputed fake batch) and the chunk results are averaged - an identity in exact arithmetic."""
    import time
    g = torch.Generator().manual_seed(data_seed)
    reg = ops.Registry(dtype=F64, seed=init_seed)
    if which == 'lsun128':
        cfg = nets.Lsun128Cfg()
        h = B // 2       # two generator towers per batch, each with its own BN statistics (gan_lsun128.GEN_TOWERS)
        Gfull = lambda r, n, zz: torch.cat([nets.lsun128_generator(r, cfg, h, zz[:h]), nets.lsun128_generator(r, cfg, h, zz[h:])])   # noqa: E731
        D = lambda r, xx, uu: nets.lsun128_discriminator(r, cfg, xx, 0.8, 0.5, 0.5, uu)                                            # noqa: E731
        out_dim = cfg.OUTPUT_DIM
        feat = [(cfg.DIM_D_8, 8, 8)] * 3
        with torch.no_grad():
            D(reg, Gfull(reg, 4, torch.zeros(4, 128, dtype=F64)), [torch.ones(4, *s, dtype=F64) for s in feat])
    else:
        Gfull = lambda r, n, zz: nets.cifar_generator(r, n, zz, DIM=128)        # noqa: E731
        D = lambda r, xx, uu: nets.cifar_discriminator(r, xx, uu, DIM=128)      # noqa: E731
        out_dim = 3072
        feat = [(128, 16, 16), (256, 8, 8), (512, 4, 4)]
        with torch.no_grad():
            D(reg, Gfull(reg, 2, torch.zeros(2, 128, dtype=F64)), [torch.full((2,) + s, 0.9, dtype=F64) for s in feat])
    for n, t in reg.items():                       # fp32-representable weights on both sides
        with torch.no_grad():
            t.copy_(t.float().double())
    real_in = torch.randint(0, 256, (B, out_dim), generator=g, dtype=torch.int32)
    rnd = steps.make_rnd_dcgan_d(B, feat, g)
    real = 2 * ((real_in.double() / 255.) - .5)
    t0 = time.time()
    with torch.no_grad():
        fake = Gfull(reg, B, rnd['z'])
    print('generator forward %.1f s' % (time.time() - t0), flush=True)
    names = [n for n, _ in reg.trainable_with_name('Discriminator')]
    acc = {n: torch.zeros_like(reg[n]) for n in names}
    terms = {k: 0.0 for k in ('cost', 'wgan_only', 'ct', 'gp')}
    slopes = []
    nch = B // chunk
    assert nch * chunk == B
    for c in range(nch):
        rows = slice(c * chunk, (c + 1) * chunk)
        sub = {k: ([t[rows] for t in v] if isinstance(v, list) else v[rows]) for k, v in rnd.items()}
        o = steps.dcgan_d_losses(reg, lambda r, n, zz: fake[rows], D, real[rows], sub)
        gr = steps.grads_of(o['cost'], reg, 'Discriminator')
        for n in names:
            if n in gr:
                acc[n] += gr[n].detach() / nch
        for k in terms:
            terms[k] += o[k].item() / nch
        slopes.append(o['slopes'].detach())
        del o, gr
        print('chunk %d/%d  %.1f s  cost so far %.6f' % (c + 1, nch, time.time() - t0, terms['cost'] * nch / (c + 1)), flush=True)
    out = {'cfg': np.array([B, chunk, init_seed, data_seed, nsample]), 'names': np.array(names), 'slopes': npy(torch.cat(slopes)),
           'fake_abs_sum': np.array(fake.abs().sum().item()),
           'theta_abs_sum': np.array(sum(reg[n].detach().abs().sum().item() for n in names))}
    for k, v in terms.items():
        out['loss.' + k] = np.array(v)
    for n in names:
        flat = acc[n].reshape(-1)
        idx = _sample_index(n, flat.numel(), nsample)
        out['norm.' + n] = np.array(flat.norm().item())
        out['vals.' + n] = npy(flat[torch.from_numpy(idx)]).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, '%s_dstep_%d.npz' % (which, B)), **out)


def gstep_b64_fixture(which, B=64, chunk=8, init_seed=9, data_seed=5, nsample=1024, check_direct=False):
    """One GENERATOR step of config[4] ('lsun128') / config[1] ('cifar') at the full batch B = 64 in fp64: gen_cost = -mean(D(G(z)))
    (TF/CT_gan_cifar.py:124; LS/wgan_LSUN_Bedrooms128.py:246-258 with its two generator towers) and its gradient w.r.t. every generator
    parameter (the B = 64 fixtures pinned the critic step only).  Same conventions as dstep_b64_fixture: seeds, not
    tensors; loss, per-parameter gradient norms and `nsample` fixed entries.

    Memory: the generator's graph at B = 64 is kept whole (its BatchNorm couples the samples of a tower); the critic couples no samples
    and no critic parameter gradient is needed, so d cost / d x is evaluated `chunk` rows at a time on detached rows of the fake batch and
    pushed through the generator's graph once - the chain rule, an identity in exact arithmetic (check_direct: asserted against the
    one-graph evaluation, which fits for config[1])."""
    import time
    g = torch.Generator().manual_seed(data_seed)
    reg = ops.Registry(dtype=F64, seed=init_seed)
    if which == 'lsun128':
        cfg = nets.Lsun128Cfg()
        h = B // 2
        Gfull = lambda r, n, zz: torch.cat([nets.lsun128_generator(r, cfg, h, zz[:h]), nets.lsun128_generator(r, cfg, h, zz[h:])])   # noqa: E731
        D = lambda r, xx, uu: nets.lsun128_discriminator(r, cfg, xx, 0.8, 0.5, 0.5, uu)                                            # noqa: E731
        feat = [(cfg.DIM_D_8, 8, 8)] * 3
        with torch.no_grad():
            D(reg, torch.cat([nets.lsun128_generator(reg, cfg, 2, torch.zeros(2, 128, dtype=F64))] * 2), [torch.ones(4, *s, dtype=F64) for s in feat])
    else:
        Gfull = lambda r, n, zz: nets.cifar_generator(r, n, zz, DIM=128)        # noqa: E731
        D = lambda r, xx, uu: nets.cifar_discriminator(r, xx, uu, DIM=128)      # noqa: E731
        feat = [(128, 16, 16), (256, 8, 8), (512, 4, 4)]
        with torch.no_grad():
            D(reg, Gfull(reg, 2, torch.zeros(2, 128, dtype=F64)), [torch.full((2,) + s, 0.9, dtype=F64) for s in feat])
    for n, t in reg.items():                       # fp32-representable weights on both sides
        with torch.no_grad():
            t.copy_(t.float().double())
    rnd = steps.make_rnd_dcgan_g(B, feat, g)
    names = [n for n, _ in reg.trainable_with_name('Generator')]
    params = [reg[n] for n in names]
    t0 = time.time()
    x = Gfull(reg, B, rnd['z'])
    print('generator forward (with graph) %.1f s' % (time.time() - t0), flush=True)
    gx = torch.zeros_like(x)
    cost = 0.0
    nch = B // chunk
    assert nch * chunk == B
    for c in range(nch):
        rows = slice(c * chunk, (c + 1) * chunk)
        xs = x[rows].detach().requires_grad_(True)
        d, _ = D(reg, xs, [u[rows] for u in rnd['u_fake']])
        cc = -d.sum() / B
        (gxs,) = torch.autograd.grad(cc, xs)
        gx[rows] = gxs
        cost += cc.item()
        del d, cc, gxs, xs
        print('chunk %d/%d  %.1f s' % (c + 1, nch, time.time() - t0), flush=True)
    grads = torch.autograd.grad(x, params, grad_outputs=gx, allow_unused=True)
    print('generator backward %.1f s  cost %.9f' % (time.time() - t0, cost), flush=True)
    if check_direct:
        o = steps.dcgan_g_losses(reg, Gfull, D, B, rnd)
        gd = steps.grads_of(o['cost'], reg, 'Generator')
        assert abs(o['cost'].item() - cost) <= 1e-12 * max(1.0, abs(cost)), (o['cost'].item(), cost)
        for n, ga in zip(names, grads):
            if ga is not None:
                assert (ga - gd[n]).abs().max().item() <= 1e-11 * max(gd[n].abs().max().item(), 1e-30), n
        print('chunked == one-graph evaluation (1e-11)', flush=True)
    out = {'cfg': np.array([B, chunk, init_seed, data_seed, nsample]), 'names': np.array([n for n, ga in zip(names, grads) if ga is not None]),
           'loss.cost': np.array(cost), 'fake_abs_sum': np.array(x.detach().abs().sum().item()),
           'theta_abs_sum': np.array(sum(reg[n].detach().abs().sum().item() for n in names))}
    for n, ga in zip(names, grads):
        if ga is None:
            continue
        flat = ga.detach().reshape(-1)
        idx = _sample_index(n, flat.numel(), nsample)
        out['norm.' + n] = np.array(flat.norm().item())
        out['vals.' + n] = npy(flat[torch.from_numpy(idx)]).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, '%s_gstep_%d.npz' % (which, B)), **out)


if __name__ == '__main__':
    torch.set_num_threads(4)
    which = sys.argv[1:] or ['ops', 'resnet', 'loop']
    if 'ops' in which:
        ops_fixture()
    if 'resnet' in which:
        resnet_fixture()
    if 'loop' in which:
        loop_trace_fixture()
    if 'loop32' in which:       # the oracle's own fp32 twin on the same streams: how far ANY fp32 evaluation drifts from the fp64 trace
        loop_trace_fixture(fname='resnet_loop_trace_f32twin.npz', dtype=torch.float32)
    if 'loopfull' in which:     # the loop at FULL width (DIM 128, B 64), 2 iterations: what tests/test_gpu_graph_loop.py replays on the device
        torch.set_num_threads(8)
        loop_trace_fixture(dim=128, B=64, iters=2, fname='resnet_loop_128_64.npz')
        loop_trace_fixture(dim=128, B=64, iters=2, fname='resnet_loop_128_64_f32twin.npz', dtype=torch.float32)
    if 'lsun64' in which:       # config[4] at B = 64 (chunked, ~minutes of host time): what tests/test_lsun128.py compares the fp16 step with
        torch.set_num_threads(6)
        dstep_b64_fixture('lsun128')
    if 'cifar64' in which:      # config[1] at B = 64
        torch.set_num_threads(6)
        dstep_b64_fixture('cifar')
    if 'cifar64g' in which:     # config[1]'s generator step at B = 64 (the chunked evaluation asserted against the one-graph evaluation)
        torch.set_num_threads(6)
        gstep_b64_fixture('cifar', check_direct=True)
    if 'lsun64g' in which:      # config[4]'s generator step at B = 64
        torch.set_num_threads(6)
        gstep_b64_fixture('lsun128')
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
