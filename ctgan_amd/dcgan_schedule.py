"""Hand-scheduled critic step of the two DCGAN scripts (MODE 'wgan-CT': TF/CT_gan_cifar.py:81-154, TF/CT_gan_mnist.py:89-179) - the
schedule of critic_schedule.py (round 5) on the three-conv LeakyReLU + dropout critic.

The autograd form (dcgan_step.DCGANTrainer.d_losses) evaluates the critic on the 3B rows [real (masks A) ; fake (masks C) ; real (masks B)]
and, separately, on the B rows of x_hat with create_graph - a 64-row forward, a 64-row backward and a 64-row double backward of three
convs each: 45 of the 52 launches of config[1]'s smallest conv tile (`conv16<64x64,k64>`, 21 % of its iteration).  The critic is piecewise
linear and couples no samples, so here

  phase A  ONE forward over [real, fake, real | x_hat] = 4B rows; the dropout of each activation draws the main rows' stream on the
           first 3B rows and the penalty pass's stream on the last B (the same draws as the two separate evaluations);
  phase B  ONE backward chain over the 4B rows: seeds = (gradient of the fused loss heads through the Linear layer ; W_out for the
           penalty rows - dD/dfeatures), weight gradients from the first 3B rows, dD/dx_hat from the last B;
  phase C  the penalty's double backward on the B rows: the cotangent of dD/dx_hat pushed forward through the three convs with the
           LeakyReLU / dropout factors as constants, its weight gradients queued as second segments of the same filters.

With a loss scale S (fp16 mode) the main rows' seed carries S, the penalty rows' first backward stays unscaled (its slopes enter the loss
value) and S enters phase C through the seed of gp, exactly as in the autograd form.
"""
import os as _os

import torch

from . import functional as F
from . import kernels as K
from . import tflib as lib
from .critic_schedule import _Grads
from .kernels import ConvGeom

# A/B switch: the hand-scheduled critic step of the DCGAN scripts (default) / the autograd path
MERGED_BWD = _os.environ.get('CTGAN_DCGAN_MERGED_BWD', '1') != '0'


def usable(tr, rnd, fake, real_in):
    m = tr.mod
    spec = getattr(m, 'SCHEDULED_CRITIC', None)
    return bool(MERGED_BWD and spec is not None and rnd is None and fake is not None and tr.piecewise and F.LRELU_DROP_FUSION and F.DEFER_WGRADS
                and m.cfg.DIM % 32 == 0 and fake.is_contiguous() and real_in.is_contiguous())


def _dgrad(gy, w, g, N, out_strides=None):
    return K.conv_dgrad(gy, w, g, N, out_strides=out_strides, bias=None, wt=F._repacked(w, g))


def critic_step(tr, real_in, fake):
    """-> (out, grads aligned with tr.d_params); inside torch.no_grad() and functional.deferred_wgrads()."""
    m, cfg = tr.mod, tr.mod.cfg
    c0, H = m.SCHEDULED_CRITIC['channels'], m.SCHEDULED_CRITIC['size']
    B, D = cfg.BATCH_SIZE, cfg.DIM
    rng = tr.rng
    P = lib.param
    assert not torch.is_grad_enabled()
    dev = fake.device
    T, M3 = 4 * B, 3 * B
    alpha_l, keep = 0.2, 0.5

    # ------------------------------------------------------------------ phase A
    real = m.real_prep(real_in)
    alpha = rng.uniform(B, 1)
    interp = K.interpolate(real, fake, alpha)
    x4 = torch.cat([real, fake, real, interp], 0)                  # rows: real (masks A), fake (masks C), real (masks B) | x_hat
    main_specs = [F.drop_spec(rng, keep) for _ in range(3)]        # call-site order of the autograd form: the 3B-row pass first ...
    gp_specs = [F.drop_spec(rng, keep) for _ in range(3)]          # ... then the penalty pass
    img = x4.reshape(T, c0, H, H)
    W1, b1 = P('Discriminator.1.Filters'), P('Discriminator.1.Biases')
    W2, b2 = P('Discriminator.2.Filters'), P('Discriminator.2.Biases')
    W3, b3 = P('Discriminator.3.Filters'), P('Discriminator.3.Biases')
    Wo, bo = P('Discriminator.Output.W'), P('Discriminator.Output.b')
    g1 = ConvGeom(c0, H, H, D, 5, 5, 2)
    g2 = ConvGeom(D, g1.P, g1.Q, 2 * D, 5, 5, 2)
    g3 = ConvGeom(2 * D, g2.P, g2.Q, 4 * D, 5, 5, 2)
    few = K.fewch_handles(g1)
    if few:
        w1k, g1k = W1, g1
        x1 = img
    else:
        # few input channels outside the direct kernels: patches once, then a 1x1 conv on the GEMM kernels (functional.conv2d)
        cpad = -(-(25 * c0) // 32) * 32
        w1k = F.GemmFilterFn.apply(W1, 'cols', cpad).view(1, 1, cpad, D)
        g1k = ConvGeom(cpad, g1.P, g1.Q, D, 1, 1, 1)
        x1 = K.im2col(img, g1, cpad)

    # The LeakyReLU + dropout pair after every conv rides the conv's epilogue where the 16-bit slice kernels run it (kernels._conv_act;
    # its own launch otherwise - the same draws either way):
    def act(i, ref=None):
        """dropout(LeakyReLU(.)) on all 4B rows: the main rows on their stream, the penalty rows on theirs (each indexed from its own first
        row).  ref = the forward result: the pair's backward, g * slope(y) * mask / keep (functional.LReluDropBwdFn)"""
        return {'alpha': alpha_l, 'ref': ref, 'drop': {'ranges': [(M3, main_specs[i]), (T, gp_specs[i])]}}

    def act_bwd(g, y, i):
        ms, gs = main_specs[i], gp_specs[i]
        return K.lrelu_dropout_rng2(g, y, M3, alpha_l, ms[0], ms[1], ms[2], gs[2], ms[3])

    def act_gp(y, i):
        """the same diagonal factor applied to a cotangent on the penalty rows only (the double backward: the map is its own adjoint)"""
        return {'alpha': alpha_l, 'ref': y[M3:T], 'drop': gp_specs[i]}

    a1 = K.conv_fwd(x1, w1k, b1, g1k, act=act(0))
    a2 = K.conv_fwd(a1, W2, b2, g2, act=act(1))
    a3 = K.conv_fwd(a2, W3, b3, g3, act=act(2))
    nf = 4 * D * g3.P * g3.Q
    # The reference flattens NCHW (reshape [-1, 4*4*4*DIM]); here the features stay in the convs' channels-last order and W_out is permuted
    # to it instead (8 K elements against three passes over the activations): d = <f, W> and the consistency term's ||f - f'||^2 do not
    # depend on the order of the features.
    f = a3.permute(0, 2, 3, 1).reshape(T, nf)                      # a view: a3 is dense channels-last
    Wp = Wo.reshape(4 * D, g3.P, g3.Q).permute(1, 2, 0).reshape(nf, 1).contiguous()
    gL = ConvGeom(nf, 1, 1, 1, 1, 1, 1)
    Wv = Wp.view(1, 1, nf, 1)
    f_main = f[:M3]
    d = K.conv_fwd(f_main.reshape(M3, nf, 1, 1), Wv, bo, gL).reshape(M3)
    out5, ct_i, _ = K.critic_heads_fwd(d, f_main, None, None, B, cfg.LAMBDA_2, cfg.Factor_M, 0.0, None)

    # ------------------------------------------------------------------ phase B
    G = _Grads()
    seed = tr.cost_seed().reshape(1)
    gd, gf, _ = K.critic_heads_bwd(d, f_main, None, None, ct_i, seed, B, cfg.LAMBDA_2, cfg.Factor_M, 0.0)
    gd4 = gd.reshape(M3, 1, 1, 1)
    gWo, gbo = K.conv_wgrad(f_main.reshape(M3, nf, 1, 1), gd4, gL, with_bias=True)
    g_a3 = K.empty_cl(T, 4 * D, g3.P, g3.Q, dev)
    g_f = g_a3.permute(0, 2, 3, 1).reshape(T, nf)                                            # the same storage in feature order
    K.axpby(K.conv_dgrad(gd4, Wv, gL, M3).reshape(M3, nf), gf, 1.0, 1.0, out=g_f[:M3])     # through the Linear layer + the CT term's direct part
    g_f[M3:].copy_(Wp.reshape(1, nf).expand(B, nf))                                          # penalty rows: dD/dfeatures = W_out
    g_c3 = act_bwd(g_a3, a3, 2)
    G.wgrad('Discriminator.3', a2[:M3], g_c3[:M3], W3, g3, False, True)
    g_c2 = K.conv_dgrad(g_c3, W3, g3, T, wt=F._repacked(W3, g3), act=act(1, a2))
    G.wgrad('Discriminator.2', a1[:M3], g_c2[:M3], W2, g2, False, True)
    g_c1 = K.conv_dgrad(g_c2, W2, g2, T, wt=F._repacked(W2, g2), act=act(0, a1))
    # the first conv's two uses (dropout-pass rows; below, the penalty's double backward): queued like every filter in the fp32 mode and on
    # the direct few-channel kernels; in the 16-bit modes its im2col'd GEMM filter (96 input columns) is outside the grouped 16-bit launch and
    # each use would be a weight gradient + reduction of its own - there the two uses go to ONE multi-segment launch of the fp32 family
    multi1 = (not few) and K.MMA_DTYPE is not None
    if not multi1:
        G.wgrad('Discriminator.1', x1[:M3], g_c1[:M3], w1k, g1k, False, True)
    nchw = (c0 * H * H, H * H, H, 1)
    if few:
        gx = _dgrad(g_c1[M3:], W1, g1, B, out_strides=nchw)
    else:
        gx = K.col2im(K.to_channels_last(_dgrad(g_c1[M3:], w1k, g1k, B)), g1, B, nchw)
    grads_x = gx.reshape(B, cfg.OUTPUT_DIM)
    _, slopes = K.gp_fwd(grads_x, float(cfg.LAMBDA), True)

    # ------------------------------------------------------------------ phase C
    ggx, gp = K.gp_bwd_mean(grads_x, slopes, seed, float(cfg.LAMBDA), out5)
    ggx4 = ggx.reshape(B, c0, H, H)
    u_x1 = ggx4 if few else K.im2col(ggx4, g1, g1k.C)
    u_a1 = K.conv_fwd(u_x1, w1k, None, g1k, act=act_gp(a1, 0))
    if multi1:
        dw1 = torch.empty(w1k.shape, dtype=torch.float32, device=dev)
        db1 = torch.empty(D, dtype=torch.float32, device=dev)
        try:
            K.conv_wgrad_multi([(x1[:M3], g_c1[:M3], False, True), (u_x1, g_c1[M3:], False, False)], g1k, dw1, db1)
            G._put('Discriminator.1.Filters', dw1); G._put('Discriminator.1.Biases', db1)
        except NotImplementedError:
            G.wgrad('Discriminator.1', x1[:M3], g_c1[:M3], w1k, g1k, False, True)
            G.wgrad('Discriminator.1', u_x1, g_c1[M3:], w1k, g1k, False, False)
    else:
        G.wgrad('Discriminator.1', u_x1, g_c1[M3:], w1k, g1k, False, False)
    u_a2 = K.conv_fwd(u_a1, W2, None, g2, act=act_gp(a2, 1))
    G.wgrad('Discriminator.2', u_a1, g_c2[M3:], W2, g2, False, False)
    u_a3 = K.conv_fwd(u_a2, W3, None, g3, act=act_gp(a3, 2))
    G.wgrad('Discriminator.3', u_a2, g_c3[M3:], W3, g3, False, False)
    u_f = u_a3.permute(0, 2, 3, 1).reshape(B, nf)
    # the penalty rows' seed was W_out itself: its cotangent sums over the rows
    K.axpby(gWo.reshape(-1), K.colsum_channels(u_f.reshape(B, nf, 1, 1)), 1.0, 1.0, out=gWo.reshape(-1))
    gWo = gWo.reshape(g3.P, g3.Q, 4 * D).permute(2, 0, 1)                                    # back to the parameter's (c, h, w) order

    by = G.by_name
    if not few:        # the gradient of the padded GEMM filter maps back onto the parameter by a view (functional.GemmFilterFn.backward)
        gw2 = by['Discriminator.1.Filters']
        by['Discriminator.1.Filters'] = gw2.reshape(-1, D)[:25 * c0].reshape(5, 5, c0, D)
    by['Discriminator.Output.W'], by['Discriminator.Output.b'] = gWo.reshape(Wo.shape), gbo.reshape(bo.shape)
    grads = [by.get(n) for n, _ in tr.d_named]
    out = {'cost': out5[0], 'wgan_only': out5[1], 'ct': out5[2], 'gp': gp, 'fake': fake, 'slopes': slopes, 'gp_grads': grads_x}
    return out, grads
