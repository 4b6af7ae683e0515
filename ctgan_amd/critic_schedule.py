"""Hand-scheduled backward of the ResNet critic step (TF/CT_gan_cifar_resnet.py:169-186, 277-305, 335-336): ONE data-gradient chain over
the rows of the two dropout passes AND the rows of the gradient-penalty pass.

`tf.gradients(Discriminator(x_hat), [x_hat])` (:284) and the data gradients of `disc_cost` w.r.t. the activations of the two dropout
passes (:335-336) are the same linear maps - conv^T with the same filters, the ReLU / dropout masks of each row's own forward - applied
to different rows, and the loss gradients of the dropout passes do not depend on the penalty's value.  Under autograd they are two
calls (the penalty's first backward runs inside the loss construction, with create_graph), hence two chains of launches: 192 / 128 rows
for the dropout passes and a 64-row chain for the penalty, whose 8x8 and 16x16 layers do not fill the chip.  Here the step is scheduled
by hand (no autograd tape):

  phase A  forward, as gan_cifar_resnet.Trainer.d_losses runs it: blocks 1-2 once on [real ; fake ; x_hat], blocks 3-4 once on
           [real, fake, real' | x_hat | real, fake (clean)] with per-range dropout; loss heads of the dropout passes (no penalty yet);
  phase B  ONE backward chain: seeds = (gradient of the loss heads ; dD/dz of the penalty rows), every data gradient on 4B rows in
           blocks 3-4 and 3B rows in blocks 1-2; weight gradients queued for the rows of the dropout passes only; the chain ends in
           dD/dx_hat on the B penalty rows -> slopes -> gp;
  phase C  the penalty's double backward on the x_hat rows only (the critic is piecewise linear: ReLU / dropout masks are constants):
           d gp / d(dD/dx_hat) pushed forward through conv(., W) per layer, its weight gradients (that cotangent (x) the penalty rows'
           data gradients of phase B) queued next to the first ones - every filter still gets ONE grouped launch entry with two segments.

Every tensor between the nodes has exactly the rows that carry information (no zero-padded 192-row cotangents).  Each launch calls the
same C-ABI entry points, with the same epilogue fusions, as the autograd path (functional.ConvFn / ConvDgradFn); that path stays - it
serves the parity mode (injected random draws), the Layernorm critic and widths outside the few-channel kernels - and is what
tests/test_host_logic_resnet.py / test_gpu_resnet_step.py compare this schedule with.
"""
import os as _os

import torch

from . import functional as F
from . import kernels as K
from . import tflib as lib
from .kernels import ConvGeom

# A/B switch: the hand-scheduled critic step (default) / the autograd path
MERGED_BWD = _os.environ.get('CTGAN_MERGED_BWD', '1') != '0'


def usable(R, rnd, rng, real_int, fake):
    """Is the hand-scheduled step the same computation as Trainer.d_losses + autograd here?  Only the fully fused default path
    (every switch it builds on at its default), in-kernel Philox draws, a piecewise-linear critic whose first layers run on the
    direct few-channel kernels."""
    cfg = R.cfg
    D = cfg.DIM_D
    return bool(MERGED_BWD and rnd is None and rng is not None and fake is not None and R.PREP_FUSION and R.TRUNK_SHARE and R.TAIL_SHARE
                and R.HEADS_FOLD and R.FUSE_RELU and R.DROP_FUSION and R.HEAD_FUSION and F.FORK_FUSION and F.RESAMPLE_FUSION
                and F.MASK_IN_DGRAD_EPILOGUE and F.PREMASK_FUSION and F.DEFER_WGRADS and not cfg.NORMALIZATION_D
                and R._heads_fusable(rnd, rng) and D % 32 == 0 and cfg.OUTPUT_DIM == 3072
                and K.fewch_handles(ConvGeom(3, 32, 32, D, 3, 3, 1)) and K.fewch_handles(ConvGeom(3, 16, 16, D, 1, 1, 1))
                and real_int.is_contiguous() and fake.is_contiguous())


def _dgrad(gy, w, g, N, mask=None, resid=None, drop=None, out_strides=None):
    """conv^T(gy, w) [kept where mask > 0] [+ resid] [x dropout mask]: what ConvDgradFn.forward launches."""
    return K.conv_dgrad(gy, w, g, N, out_strides=out_strides, bias=None, wt=F._repacked(w, g, drop is None), mask=mask, resid=resid, drop=drop)


def _chain_ok(x, D, main_specs, gp_specs):
    """May blocks 3-4 of the backward passes run as one launch per chain (kernels.conv_chain8x8)?  8 x 8 x 128 images in a split-mode routing, and
    both dropout sites drawn in the kernels with one keep / seed / counter for the two row ranges (always so on the default path)."""
    if not (hasattr(K, 'chain8x8_usable') and K.chain8x8_usable(x, D, 8, 8)):
        return False
    specs = list(main_specs[:2]) + list(gp_specs[:2])
    if any(sp is None or len(sp) != 4 for sp in specs):
        return False
    return all(main_specs[i][0] == gp_specs[i][0] and main_specs[i][1] == gp_specs[i][1] and main_specs[i][3] is gp_specs[i][3] for i in range(2)) \
        and main_specs[0][1] == main_specs[1][1] and main_specs[0][3] is main_specs[1][3]


class _Grads:
    """Collects the step's parameter gradients by name.  A filter's weight gradient is requested once per use (F._wgrad: queued per
    filter inside deferred_wgrads; the FIRST request of a filter returns the buffer the flush fills, later ones return None)."""

    def __init__(self):
        self.by_name = {}
        self.keys = {}          # conv name -> the queue key of its filter (functional._wgrad)

    def wgrad(self, name, x, gy, w, g, relu_x, with_bias, spread=None):
        self.keys[name] = (w.data_ptr(), (g.C, g.H, g.W, g.K, g.R, g.S, g.stride))
        gw, gb = F._wgrad(x, gy, w, g, relu_x, with_bias)
        if gw is not None:
            if spread is not None:
                # w is the cached spread filter of parameter `name`: its gradient is folded back after the flush (FilterSpreadFn.backward)
                R_, S_ = gw.shape[0] - 1, gw.shape[1] - 1
                out = torch.empty((R_, S_, gw.shape[2], gw.shape[3]), dtype=torch.float32, device=gw.device)
                # queued use: gw is the filter's ONE result buffer, filled by the flush - fold it after that, once.  A use launched at
                # once (16-bit modes' per-filter policy, f32x3, shapes the column kernel rejects) returns a FINISHED gradient per use:
                # fold it now - a deferred fold per use would have _put sum buffers nothing has written yet
                if F._DEFER['on'] and self.keys[name] in F._DEFER['groups']:
                    F._DEFER['post'].append(('fold', gw, spread, False, out))
                else:
                    K.filter_fold(gw.contiguous(), spread, False, out=out)
                gw = out
            self._put(name + '.Filters', gw)
        if gb is not None:
            self._put(name + '.Biases', gb)

    def _put(self, key, val):
        """A queued filter hands its buffer over once; a filter whose uses are launched at once (the 16-bit modes' per-filter policy for
        large layers, functional._wgrad: never mixed with queued uses) returns one finished gradient per use - summed here, as autograd would."""
        if key in self.by_name:
            K.axpby(self.by_name[key], val, 1.0, 1.0, out=self.by_name[key])
        else:
            self.by_name[key] = val


EARLY_CONVS = ('Discriminator.1.Conv1', 'Discriminator.1.Conv2', 'Discriminator.1.Shortcut', 'Discriminator.2.Conv1', 'Discriminator.2.Conv2',
               'Discriminator.2.Shortcut')


def critic_step(tr, R, real_int, labels, fake, early=None):
    """One critic step's losses AND parameter gradients -> (out, grads): `out` as Trainer.d_losses returns it, `grads` aligned with
    tr.d_params.  Must run inside torch.no_grad() and functional.deferred_wgrads() (the caller flushes the queue by leaving the latter).

    early (batch-sharded steps, SURVEY 5.7 / north star: "all-reduce overlapped with backward"): callable(dict name -> gradient).  The
    weight gradients of blocks 1-2 - a contiguous prefix of the flat bucket, 45 % of its bytes - are complete as soon as the
    penalty's double backward has left block 2, while six more launches of blocks 3-4 are still to come: with `early` they are flushed
    there (functional.flush_partial: a grouped launch of their own) and handed over, so that their all-reduce can run under the rest."""
    cfg = R.cfg
    B, D = cfg.BATCH_SIZE, cfg.DIM_D
    rng = tr.rng
    P = lib.param
    use_ac = cfg.CONDITIONAL and cfg.ACGAN
    assert not torch.is_grad_enabled()

    # ------------------------------------------------------------------ phase A: forward (Trainer.d_losses' own launches)
    rf, interp, both = K.critic_prep(real_int, fake, rng.seed, rng._sid(), rng._sid(), rng.ctr, 0.0, 1. / 128, 256.0)
    with F.tape_record() as trunk:
        R.DiscriminatorTrunk(both)
    y1, _c12, pool_x, h1, a2, _c22, h2 = trunk
    tail = R.shared_tail_forward(h2, B, rng, with_clean=use_ac)
    tin, a31, b3, a41, y = tail[0]
    gp_specs, main_specs, ranges = tail[1], tail[2], tail[3]
    assert ranges['main'] == (0, 3 * B) and ranges['gp'] == (3 * B, 4 * B)
    T = 4 * B                                   # rows of the merged backward in blocks 3-4: [real, fake, real' | x_hat]
    w_out, b_out = P('Discriminator.Output.W'), P('Discriminator.Output.b')
    w_ac = P('Discriminator.ACGANOutput.W') if use_ac else None
    b_ac = P('Discriminator.ACGANOutput.b') if use_ac else None
    y_clean = y[ranges['clean'][0]:ranges['clean'][1]] if use_ac else None
    # loss heads of the dropout passes; the penalty joins out5[0] / out5[4] when its value exists (phase C's first launch)
    out5, f_all, d_all, _a, ct_i, probs, acc = K.tail_critic_heads_fwd(y[:3 * B], B, w_out, b_out, w_ac, b_ac, labels, None, cfg.LAMBDA_2, cfg.Factor_M,
                                                                        cfg.ACGAN_SCALE if use_ac else 0.0, y_clean=y_clean, clean_relu=False)

    # ------------------------------------------------------------------ phase B: one backward chain over [dropout-pass rows ; penalty rows]
    G = _Grads()
    one = tr.cost_seed(out5[0]).reshape(1)
    g3 = ConvGeom(D, 8, 8, D, 3, 3, 1)
    gy = K.empty_cl(T, D, 8, 8, y.device)
    # both seeds in one launch: the loss heads' gradient on the dropout-pass rows, dD/dz = (y > 0) w_out / hw / keep on the penalty rows
    _, gw_out, gb_out, gw_ac, gb_ac = K.tail_heads_bwd(y[:3 * B], d_all, f_all, probs, labels, ct_i, one, B, cfg.LAMBDA_2, cfg.Factor_M,
                                                       cfg.ACGAN_SCALE if use_ac else 0.0, 1.0 / 0.5, w_out, w_ac, out=gy[:3 * B],
                                                       y_gp=y[3 * B:T], out_gp=gy[3 * B:T])

    def rdrop(i):         # the dropout in front of block 3 (i = 0) / block 4 (i = 1): each range with the mask of its own forward draw
        return {'ranges': [(3 * B, main_specs[i]), (T, gp_specs[i])]}

    W42, W41 = P('Discriminator.4.Conv2.Filters'), P('Discriminator.4.Conv1.Filters')
    W32, W31 = P('Discriminator.3.Conv2.Filters'), P('Discriminator.3.Conv1.Filters')
    m = 3 * B                                   # rows whose data gradients are loss gradients (weight gradients come from these)
    chain = _chain_ok(gy, D, main_specs, gp_specs)
    if chain:
        # blocks 4 and 3 of the backward in ONE launch, one image per workgroup (kernels.conv_chain8x8, csrc/chain8x8.hip): the four data
        # gradients with their mask / residual / ranged-dropout epilogues; every intermediate goes to HBM once (the weight gradients read them)
        g_a41, g_z3, g_a31, g_tin = K.conv_chain8x8(gy, [
            {'save': 1},                                                                                      # slot 1 = gy
            {'w': W42, 'op': 1, 'mask': a41[:T], 'out': True},
            {'w': W41, 'op': 1, 'mask': b3[:T], 'resid': 1, 'drop': 1, 'save': 2, 'out': True},               # g_z3 (slot 2)
            {'w': W32, 'op': 1, 'mask': a31[:T], 'out': True},
            {'w': W31, 'op': 1, 'mask': tin[:T], 'resid': 2, 'drop': 2, 'out': True}],
            drops=[(main_specs[1][0], main_specs[1][2], gp_specs[1][2], 3 * B), (main_specs[0][0], main_specs[0][2], gp_specs[0][2], 3 * B)],
            seed=main_specs[0][1], ctr=main_specs[0][3])
    else:
        g_a41 = _dgrad(gy, W42, g3, T, mask=a41[:T])
        g_z3 = _dgrad(g_a41, W41, g3, T, mask=b3[:T], resid=gy, drop=rdrop(1))
        g_a31 = _dgrad(g_z3, W32, g3, T, mask=a31[:T])
        g_tin = _dgrad(g_a31, W31, g3, T, mask=tin[:T], resid=g_z3, drop=rdrop(0))
    G.wgrad('Discriminator.4.Conv2', a41[:m], gy[:m], W42, g3, True, True)
    G.wgrad('Discriminator.4.Conv1', b3[:m], g_a41[:m], W41, g3, True, True)
    G.wgrad('Discriminator.3.Conv2', a31[:m], g_z3[:m], W32, g3, True, True)
    G.wgrad('Discriminator.3.Conv1', tin[:m], g_a31[:m], W31, g3, True, True)
    # rows [real, fake | real' | x_hat] -> rows [real, fake, x_hat] of the trunk (pass 2 shares the trunk rows of the real half)
    g_h2 = K.rows_cat_bwd(g_tin, 2 * B, B, B)

    m = 2 * B
    W21 = P('Discriminator.2.Conv1.Filters')
    w22 = F._cached_filter(P('Discriminator.2.Conv2.Filters'), K.FILTER_SPREAD, (0, 0), 0.25)
    wsc2 = F._cached_filter(P('Discriminator.2.Shortcut.Filters'), K.FILTER_SPREAD, (0, 0), 0.25)
    w12 = F._cached_filter(P('Discriminator.1.Conv2.Filters'), K.FILTER_SPREAD, (0, 0), 0.25)
    W11, Wsc1 = P('Discriminator.1.Conv1.Filters'), P('Discriminator.1.Shortcut.Filters')
    g21 = ConvGeom(D, 16, 16, D, 3, 3, 1)
    g22 = ConvGeom(D, 16, 16, D, 4, 4, 2)
    gsc2 = ConvGeom(D, 16, 16, D, 2, 2, 2)
    g12 = ConvGeom(D, 32, 32, D, 4, 4, 2)
    g11 = ConvGeom(3, 32, 32, D, 3, 3, 1)
    gsc1 = ConvGeom(3, 16, 16, D, 1, 1, 1)
    N3 = 3 * B
    gx_sc2 = _dgrad(g_h2, wsc2, gsc2, N3)
    G.wgrad('Discriminator.2.Shortcut', h1[:m], g_h2[:m], wsc2, gsc2, False, True, spread=0.25)
    g_a2 = _dgrad(g_h2, w22, g22, N3, mask=a2)
    G.wgrad('Discriminator.2.Conv2', a2[:m], g_h2[:m], w22, g22, True, True, spread=0.25)
    g_h1 = _dgrad(g_a2, W21, g21, N3, mask=h1, resid=gx_sc2)
    G.wgrad('Discriminator.2.Conv1', h1[:m], g_a2[:m], W21, g21, True, True)
    g_y1 = _dgrad(g_h1, w12, g12, N3, mask=y1)
    G.wgrad('Discriminator.1.Conv2', y1[:m], g_h1[:m], w12, g12, True, True, spread=0.25)
    G.wgrad('Discriminator.1.Shortcut', pool_x[:m], g_h1[:m], Wsc1, gsc1, False, True)
    x4 = both.reshape(N3, 3, 32, 32)
    G.wgrad('Discriminator.1.Conv1', x4[:m], g_y1[:m], W11, g11, False, True)
    # every (x, dy) pair of the dropout-pass rows exists: their weight gradients go to a side stream now, under the penalty's double backward
    # (phase C: a dependent chain of 64-row launches that leaves most of the chip idle); phase C's own segments join them at the final flush
    F.flush_async()
    # the chain's end on the penalty rows: dD/dx_hat through the first conv and through the pooled shortcut (:146-153)
    gx = _dgrad(g_y1[m:], W11, g11, B, out_strides=(3072, 1024, 32, 1))
    slopes = K.gp_finish(gx, _dgrad(g_h1[m:], Wsc1, gsc1, B), 0.25)      # += the shortcut's gradient through the 2x2 mean pool; per-sample norms
    grads_x = gx.reshape(B, cfg.OUTPUT_DIM)

    # ------------------------------------------------------------------ phase C: the penalty's double backward, x_hat rows only
    # cotangent of dD/dx_hat (and the penalty's value into the sums that contain it)
    ggx, gp = K.gp_bwd_mean(grads_x, slopes, one, cfg.GP_LAMBDA, out5)
    ggx4 = ggx.reshape(B, 3, 32, 32)
    # block 1 (pushed forward): through the first conv - its result is the cotangent of the masked g_y1, the mask rides the epilogue
    u_y1 = K.conv_fwd(ggx4, W11, None, g11, mask=y1[m:])
    G.wgrad('Discriminator.1.Conv1', ggx4, g_y1[m:], W11, g11, False, False)
    p_x = K.pool2(ggx4, 0.25)
    G.wgrad('Discriminator.1.Shortcut', p_x, g_h1[m:], Wsc1, gsc1, False, False)
    t = K.conv_fwd(u_y1, w12, None, g12)
    G.wgrad('Discriminator.1.Conv2', u_y1, g_h1[m:], w12, g12, False, False, spread=0.25)
    u_h1 = K.conv_fwd(p_x, Wsc1, None, gsc1, resid=t)                 # cotangent of g_h1
    # block 2
    u_h1m = K.lrelu_bwd(u_h1, h1[m:], 0.0)
    u_a2 = K.conv_fwd(u_h1m, W21, None, g21, mask=a2[m:])
    G.wgrad('Discriminator.2.Conv1', u_h1m, g_a2[m:], W21, g21, False, False)
    t = K.conv_fwd(u_a2, w22, None, g22)
    G.wgrad('Discriminator.2.Conv2', u_a2, g_h2[m:], w22, g22, False, False, spread=0.25)
    u_h2 = K.conv_fwd(u_h1, wsc2, None, gsc2, resid=t)                # cotangent of g_h2 = of the penalty rows of g_tin
    G.wgrad('Discriminator.2.Shortcut', u_h1, g_h2[m:], wsc2, gsc2, False, False, spread=0.25)
    if early is not None and F._DEFER['on']:
        F.flush_partial([G.keys[c] for c in EARLY_CONVS])
        early({n: G.by_name[n] for c in EARLY_CONVS for n in (c + '.Filters', c + '.Biases')})
    # block 3: dropout mask, then the ReLU mask (constants of the second pass); the dropped-only tensor goes on through the shortcut
    s1, s2 = gp_specs[0], gp_specs[1]
    if chain and K.chain8x8_usable(u_h2, D, 8, 8):
        # blocks 3 and 4 of the double backward in ONE launch: u = (u_h2 x dropout) masked; conv, mask; conv + residual, x dropout, masked; conv,
        # mask; conv + residual - the residuals (the dropped-only tensors r3 / r4) never leave the kernel
        u, u_a31, u4, u_a41, u_gz = K.conv_chain8x8(u_h2, [
            {'drop': 1, 'save': 1, 'post_mask': tin[3 * B:T], 'out': True},
            {'w': W31, 'op': 0, 'mask': a31[3 * B:T], 'out': True},
            {'w': W32, 'op': 0, 'resid': 1, 'drop': 2, 'save': 2, 'post_mask': b3[3 * B:T], 'out': True},
            {'w': W41, 'op': 0, 'mask': a41[3 * B:T], 'out': True},
            {'w': W42, 'op': 0, 'resid': 2, 'out': True}],
            drops=[(s1[0], s1[2], s1[2], 0), (s2[0], s2[2], s2[2], 0)], seed=s1[1], ctr=s1[3])
    else:
        r3, u = K.dropout_rng_mask(u_h2, tin[3 * B:T], s1[0], s1[1], s1[2], s1[3], want_dropped=True)
        u_a31 = K.conv_fwd(u, W31, None, g3, mask=a31[3 * B:T])
        u_z3 = K.conv_fwd(u_a31, W32, None, g3, resid=r3)
        # block 4
        r4, u4 = K.dropout_rng_mask(u_z3, b3[3 * B:T], s2[0], s2[1], s2[2], s2[3], want_dropped=True)
        u_a41 = K.conv_fwd(u4, W41, None, g3, mask=a41[3 * B:T])
        u_gz = K.conv_fwd(u_a41, W42, None, g3, resid=r4)
    G.wgrad('Discriminator.3.Conv1', u, g_a31[3 * B:T], W31, g3, False, False)
    G.wgrad('Discriminator.3.Conv2', u_a31, g_z3[3 * B:T], W32, g3, False, False)
    G.wgrad('Discriminator.4.Conv1', u4, g_a41[3 * B:T], W41, g3, False, False)
    G.wgrad('Discriminator.4.Conv2', u_a41, gy[3 * B:T], W42, g3, False, False)
    # the seed dD/dz = (y > 0) w_out / hw / keep depends on w_out
    K.gp_head_wgrad(u_gz, y[3 * B:T], 1.0 / 0.5, w_out, add_to=gw_out)

    by = G.by_name
    by['Discriminator.Output.W'], by['Discriminator.Output.b'] = gw_out, gb_out
    if use_ac:
        by['Discriminator.ACGANOutput.W'], by['Discriminator.ACGANOutput.b'] = gw_ac, gb_ac
    grads = [by.get(n) for n, _ in tr.d_named]
    out = {'cost': out5[0], 'wgan': out5[4], 'acgan': out5[3] if use_ac else None, 'wgan_only': out5[1], 'ct': out5[2], 'gp': gp, 'slopes': slopes,
           'fake': fake, 'real': rf[:B], 'd_real': d_all[:B], 'd_fake': d_all[B:2 * B], 'gp_grads': grads_x}
    if use_ac:
        out['acc_real'], out['acc_fake'] = acc[0], acc[1]
    return out, grads
