"""CT-WGAN ResNet for CIFAR-10 on MI355X: the hot path of TF/CT_gan_cifar_resnet.py.

Same call surface as the reference script - `Generator(n_samples, labels, noise=None)`,
`Discriminator(inputs, labels, kp1, kp2, kp3)`, the UPPERCASE hyper-parameters - built on the
drop-in `tflib.ops` (HIP kernels).  The loss graph of `:190-338` is restated imperatively in
`Trainer.d_step` / `Trainer.g_step`; the loop of `:393-434` in `Trainer.train_iteration`.

Exact restructurings relative to the reference graph (same outputs, less work):
  * the two generator towers of one step are one batch with two BatchNorm statistic groups;
  * the critic trunk (blocks 1-2, before the first dropout) is evaluated once per input and
    shared by the two dropout passes and the clean accuracy pass;
  * pass 2 is evaluated on the real half only (its fake half reaches no loss term);
  * 1x1 shortcut convs commute with mean-pool / nearest-upsample and run on the small side;
  * residual adds ride the conv epilogue; UpsampleConv never materialises the upsampled tensor.
"""
import contextlib
import functools

import torch

from . import functional as F
from . import kernels as K
from . import tflib as lib
from .optim import FlatAdam
from .rng import DeviceRNG
from .tflib.ops import batchnorm as _bn
from .tflib.ops import cond_batchnorm as _cbn
from .tflib.ops import conv2d as _conv2d
from .tflib.ops import layernorm as _ln
from .tflib.ops import linear as _linear


class Config:
    """UPPERCASE globals of TF/CT_gan_cifar_resnet.py:33-56."""
    LAMBDA_2 = 2.0
    Factor_M = 0.0
    BATCH_SIZE = 64
    GEN_BS_MULTIPLE = 2
    ITERS = 100000
    DIM_G = 128
    DIM_D = 128
    NORMALIZATION_G = True
    NORMALIZATION_D = False
    OUTPUT_DIM = 3072
    LR = 2e-4
    DECAY = True
    N_CRITIC = 5
    CONDITIONAL = True
    ACGAN = True
    ACGAN_SCALE = 1.
    ACGAN_SCALE_G = 0.1
    GP_LAMBDA = 10.0          # the literal 10.0 of :286

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(Config, k):
                raise AttributeError('unknown hyper-parameter %s' % k)
            setattr(self, k, v)


cfg = Config()
import os as _os
# A/B switch: fold the critic's ReLUs into the conv gathers / dgrad epilogues (relu(x) never materialised).
# Measured on MI355X (tools/fuse_bench.py): the forward gather and the wgrad relu-on-load are free (the relu(x)
# tensor is never written or re-read); the mask in the dgrad EPILOGUE is not (it lengthens the un-overlapped tail of
# an MFMA kernel by as much as the stand-alone mask kernel costs), so the backward keeps the stand-alone kernel
# (functional.MASK_IN_DGRAD_EPILOGUE = False).
FUSE_RELU = _os.environ.get('CTGAN_FUSE_RELU', '1') != '0'


def configure(**kw):
    global cfg
    cfg = Config(**kw)
    return cfg


def nonlinearity(x):
    return F.relu(x)


def _norm_relu(name, inputs, labels=None, groups=1):
    """Normalize followed by nonlinearity (:134-135,137-138).  Returns (tensor, relu_pending): with a norm the
    ReLU is fused into the BN kernel; without one (the critic) it is deferred into the next conv's input
    gather (`relu_in`), so relu(x) is never written to memory."""
    if ('Generator' in name) and cfg.NORMALIZATION_G:
        return Normalize(name, inputs, labels=labels, groups=groups, relu=True), False
    if ('Discriminator' in name) and cfg.NORMALIZATION_D:
        return F.relu(Normalize(name, inputs, labels=labels)), False         # Layernorm critic: not piecewise linear
    if not FUSE_RELU:
        return F.relu(inputs), False
    return inputs, True


def Normalize(name, inputs, labels=None, groups=1, relu=False):
    """TF/CT_gan_cifar_resnet.py:70-87 (+ build-only `groups`, fused `relu`)."""
    if not cfg.CONDITIONAL:
        labels = None
    if cfg.CONDITIONAL and cfg.ACGAN and ('Discriminator' in name):
        labels = None
    if ('Discriminator' in name) and cfg.NORMALIZATION_D:
        return _ln.Layernorm(name, [1, 2, 3], inputs)        # :76-77 (the main-tree op takes no labels, layernorm.py:6)
    elif ('Generator' in name) and cfg.NORMALIZATION_G:
        if labels is not None:
            return _cbn.Batchnorm(name, [0, 2, 3], inputs, labels=labels, n_labels=10, groups=groups, relu=relu)
        return _bn.Batchnorm(name, [0, 2, 3], inputs, fused=True, groups=groups, relu=relu)
    return F.relu(inputs) if relu else inputs


def ConvMeanPool(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True, resid=None, relu_in=False):
    """:89-92.  A 1x1 conv commutes with the mean pool: pool first (4x fewer MACs)."""
    if filter_size == 1:
        assert not relu_in
        if F.RESAMPLE_FUSION and input_dim % 32 == 0 and output_dim % 32 == 0:
            # pool + 1x1 conv = one 2x2 stride-2 conv (no pooled intermediate, no upsample kernel in the backward)
            return _conv2d.Conv2D(name, input_dim, output_dim, 1, inputs, he_init=he_init, biases=biases, pool=True, resid=resid)
        return _conv2d.Conv2D(name, input_dim, output_dim, 1, F.mean_pool2(inputs), he_init=he_init, biases=biases,
                              resid=resid)
    # conv + mean pool = one stride-2 conv with the spread (k+1)x(k+1) filter (functional.conv2d_mean_pool)
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=he_init, biases=biases, relu_in=relu_in,
                          pool=True, resid=resid)


def MeanPoolConv(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True, resid=None):
    """:94-98"""
    out = F.mean_pool2(inputs)
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, out, he_init=he_init, biases=biases, resid=resid)


def UpsampleConv(name, input_dim, output_dim, filter_size, inputs, he_init=True, biases=True, resid=None, relu_in=False):
    """:100-107.  upsample + conv = one stride-2 transposed conv with the spread (k+1)x(k+1) filter
    (functional.upsample_conv2d; falls back to the x_up input gather for unaligned channel counts); a 1x1
    conv commutes with the upsample and runs on the small side."""
    if filter_size == 1:
        out = _conv2d.Conv2D(name, input_dim, output_dim, 1, inputs, he_init=he_init, biases=biases)
        out = F.upsample2(out)
        return out if resid is None else F.add(out, resid)
    if relu_in:
        inputs = F.relu(inputs)
    return _conv2d.Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=he_init, biases=biases,
                          x_up=True, resid=resid)


def ResidualBlock(name, input_dim, output_dim, filter_size, inputs, resample=None, no_dropout=False, labels=None,
                  groups=1, in_drop=None, out_epi=None):
    """:109-141  (resample: None, 'down', or 'up')"""
    if resample not in (None, 'down', 'up'):
        raise Exception('invalid resample value')
    out, r1 = _norm_relu(name + '.N1', inputs, labels=labels, groups=groups)
    if resample == 'down':
        if r1:      # critic: `inputs` feeds Conv1 and the shortcut; fork=True returns it back for the shortcut
            out, inputs = _conv2d.Conv2D(name + '.Conv1', input_dim, input_dim, filter_size, out, relu_in=True, fork=True)
        else:
            out = _conv2d.Conv2D(name + '.Conv1', input_dim, input_dim, filter_size, out, relu_in=r1)
        out, r2 = _norm_relu(name + '.N2', out, labels=labels, groups=groups)
        out = ConvMeanPool(name + '.Conv2', input_dim, output_dim, filter_size, out, relu_in=r2)
        # shortcut = ConvMeanPool 1x1 (he_init=False); the residual add rides its epilogue
        return ConvMeanPool(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True, resid=out)
    if resample == 'up':
        if RESID_UP_FUSION and output_dim % 32 == 0:
            # 1x1 shortcut conv on the small side; its nearest-2x upsample happens where Conv2's epilogue reads it
            shortcut = _conv2d.Conv2D(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
            epi = {'resid_up': True}
        else:
            shortcut = UpsampleConv(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
            epi = None
        out = UpsampleConv(name + '.Conv1', input_dim, output_dim, filter_size, out, relu_in=r1)
        fused = _fused_norm_conv(name, out, output_dim, filter_size, shortcut, epi is not None, labels, groups)
        if fused is not None:
            return fused
        out, r2 = _norm_relu(name + '.N2', out, labels=labels, groups=groups)
        return _conv2d.Conv2D(name + '.Conv2', output_dim, output_dim, filter_size, out, resid=shortcut, relu_in=r2, epi=epi)
    # resample None.  in_drop: `inputs` is the result of that dropout - its mask is applied to the block's input gradient
    # in Conv1's dgrad epilogue; out_epi: dropout (/ ReLU) applied to the block's result in Conv2's epilogue.
    if r1:
        out, inputs = _conv2d.Conv2D(name + '.Conv1', input_dim, output_dim, filter_size, out, relu_in=True, fork=True,
                                     epi={'in_drop': in_drop} if in_drop is not None else None)
    else:
        assert in_drop is None
    if output_dim == input_dim:
        shortcut = inputs
    else:
        shortcut = _conv2d.Conv2D(name + '.Shortcut', input_dim, output_dim, 1, inputs, he_init=False, biases=True)
    if not r1:
        out = _conv2d.Conv2D(name + '.Conv1', input_dim, output_dim, filter_size, out, relu_in=r1)
    out, r2 = _norm_relu(name + '.N2', out, labels=labels, groups=groups)
    return _conv2d.Conv2D(name + '.Conv2', output_dim, output_dim, filter_size, out, resid=shortcut, relu_in=r2, epi=out_epi)


def OptimizedResBlockDisc1(inputs):
    """:143-153"""
    D = cfg.DIM_D
    out = _conv2d.Conv2D('Discriminator.1.Conv1', 3, D, 3, inputs)
    if FUSE_RELU:
        out = ConvMeanPool('Discriminator.1.Conv2', D, D, 3, out, relu_in=True)    # nonlinearity folded into Conv2's gather
    else:
        out = ConvMeanPool('Discriminator.1.Conv2', D, D, 3, nonlinearity(out))
    return MeanPoolConv('Discriminator.1.Shortcut', 3, D, 1, inputs, he_init=False, biases=True, resid=out)


def Generator(n_samples, labels, noise=None, groups=1, rng=None):
    """:155-167.  `groups` > 1 evaluates that many reference towers (separate BN statistics) at once."""
    G = cfg.DIM_G
    if noise is None:
        noise = rng.normal(n_samples, 128)
    out = _linear.Linear('Generator.Input', 128, 4 * 4 * G, noise)
    out = F.to_channels_last(out.reshape(-1, G, 4, 4))
    out = ResidualBlock('Generator.1', G, G, 3, out, resample='up', labels=labels, groups=groups)
    out = ResidualBlock('Generator.2', G, G, 3, out, resample='up', labels=labels, groups=groups)
    out = ResidualBlock('Generator.3', G, G, 3, out, resample='up', labels=labels, groups=groups)
    fused = _fused_output_stage(out, G, groups)
    if fused is not None:
        return fused.reshape(-1, cfg.OUTPUT_DIM)
    out = Normalize('Generator.OutputN', out, groups=groups, relu=True)
    out = _conv2d.Conv2D('Generator.Output', G, 3, 3, out, he_init=False, out_nchw=True)
    out = F.tanh(out)
    return out.reshape(-1, cfg.OUTPUT_DIM)


# A/B switch: under no_grad (the fake batches of the critic steps) the output stage tanh(Conv2D(relu(Batchnorm(h)))) (:164-166) as the
# moments + ONE conv launch - the batch norm applied while the many -> few kernel stages its input, tanh in its epilogue
OUTPUT_STAGE_FUSION = _os.environ.get('CTGAN_OUTPUT_STAGE_FUSION', '1') != '0'


def _fused_norm_conv(name, x, dim, filter_size, shortcut, resid_up, labels, groups):
    """Conv2(relu(N2(x))) + shortcut of an 'up' block (:134-141) under no_grad: the moments, then ONE conv launch with the (conditional) batch norm
    and the ReLU applied while the halo patch is staged (kernels.conv_fwd_bn_in); None where that form does not apply."""
    if (not OUTPUT_STAGE_FUSION or torch.is_grad_enabled() or not cfg.NORMALIZATION_G or 'Generator' not in name or not x.is_cuda
            or (not resid_up and shortcut is not None and tuple(shortcut.shape) != tuple(x.shape))):
        return None
    names = (name + '.N2.scale', name + '.N2.offset', name + '.Conv2.Filters', name + '.Conv2.Biases')
    if any(nm not in lib._params for nm in names):
        return None                   # first call: the operators create their parameters
    scale, offset, w, b = (lib.param(nm) for nm in names)
    lab = labels if (cfg.CONDITIONAL and labels is not None and scale.dim() == 2 and scale.shape[0] > 1) else None
    if lab is None and scale.numel() != dim:
        return None
    g = K.ConvGeom(dim, x.shape[2], x.shape[3], dim, filter_size, filter_size, 1)
    if not K.conv_fwd_bn_in_supported(x, g, labels=lab, resid=shortcut):
        return None                   # (asked before the moments: the fallback computes its own)
    try:
        mean, rstd = K.bn_stats(x, groups)
        return K.conv_fwd_bn_in(x, w, b, g, mean, rstd, scale, offset, groups, relu_in=True, labels=lab, resid=shortcut, resid_up=resid_up)
    except NotImplementedError:
        return None


def _fused_output_stage(h, G, groups):
    if not OUTPUT_STAGE_FUSION or torch.is_grad_enabled() or not cfg.NORMALIZATION_G or not h.is_cuda:
        return None
    names = ('Generator.OutputN.scale', 'Generator.OutputN.offset', 'Generator.Output.Filters', 'Generator.Output.Biases')
    if any(nm not in lib._params for nm in names):
        return None                   # first call: the operators create their parameters
    scale, offset, w, b = (lib.param(nm) for nm in names)
    g = K.ConvGeom(G, h.shape[2], h.shape[3], 3, 3, 3, 1)
    if not K.conv_fwd_bn_in_supported(h, g, tanh=True):
        return None
    try:
        mean, rstd = K.bn_stats(h, groups)
        N = h.shape[0]
        return K.conv_fwd_bn_in(h, w, b, g, mean, rstd, scale, offset, groups, relu_in=True, tanh=True,
                                out_strides=(3 * g.P * g.Q, g.P * g.Q, g.Q, 1))
    except NotImplementedError:
        return None


def DiscriminatorTrunk(inputs):
    """Blocks 1-2 of the critic: everything before the first dropout (:170-172)."""
    D = cfg.DIM_D
    out = inputs.reshape(-1, 3, 32, 32)
    out = OptimizedResBlockDisc1(out)
    return ResidualBlock('Discriminator.2', D, D, 3, out, resample='down')


def _tail_fusable(kp1, kp2, kp3, u, rng):
    return (DROP_FUSION and FUSE_RELU and F.FORK_FUSION and not cfg.NORMALIZATION_D and u is None and rng is not None
            and max(kp1, kp2, kp3) < 1.0)


def DiscriminatorTailBody(h, kp1, kp2, kp3, u=None, rng=None, mask_done=False, cat_extra=0, specs=None):
    """dropout -> block 3 -> dropout -> block 4 -> dropout -> relu (:173-179).  mask_done (fused path only): the consumer
    of the result returns the gradient w.r.t. the last conv's result, relu/dropout mask included (F.critic_tail_heads,
    F.gp_head_grad)."""
    D = cfg.DIM_D
    if _tail_fusable(kp1, kp2, kp3, u, rng):
        # dropout -> block 3 -> dropout -> block 4 -> dropout -> relu with the masks inside the conv kernels: forward in the
        # epilogue of the conv that produces the tensor, backward in the dgrad epilogue of the conv that consumed it
        s1, s2, s3 = specs if specs is not None else (F.drop_spec(rng, kp1), F.drop_spec(rng, kp2), F.drop_spec(rng, kp3))
        if cat_extra:       # the tail's input is [h ; h[:cat_extra]] (pass 2 on the real half): concat + dropout in one launch
            out = F.rows_cat_dropout(h, cat_extra, s1)
        else:
            out = F.dropout(h, kp1, spec=s1, bwd_fused=True)
        out = ResidualBlock('Discriminator.3', D, D, 3, out, resample=None, in_drop=s1,
                            out_epi={'out_drop': s2, 'out_drop_bwd_fused': True})
        return ResidualBlock('Discriminator.4', D, D, 3, out, resample=None, in_drop=s2,
                             out_epi={'out_drop': s3, 'out_relu': True, 'mask_done': mask_done})    # = relu(dropout(.)): both are >= 0 scalings
    assert not mask_done and not cat_extra

    def drop(i, x, kp):
        if kp == 1.0:
            return x
        return F.dropout(x, kp, u[i]) if u is not None else F.dropout(x, kp, rng=rng)

    out = drop(0, h, kp1)
    out = ResidualBlock('Discriminator.3', D, D, 3, out, resample=None)
    out = drop(1, out, kp2)
    out = ResidualBlock('Discriminator.4', D, D, 3, out, resample=None)
    out = drop(2, out, kp3)
    return nonlinearity(out)


def DiscriminatorTail(h, kp1, kp2, kp3, u=None, rng=None, heads=('wgan', 'acgan')):
    """dropout -> block 3 -> dropout -> block 4 -> dropout -> relu -> mean -> heads (:173-186).
    `heads`: which of the two linear heads the caller consumes (the other one is not launched)."""
    D = cfg.DIM_D
    out = DiscriminatorTailBody(h, kp1, kp2, kp3, u=u, rng=rng)
    output2 = F.spatial_mean(out)
    output_wgan = _linear.Linear('Discriminator.Output', D, 1, output2).reshape(-1) if 'wgan' in heads else None
    if cfg.CONDITIONAL and cfg.ACGAN and 'acgan' in heads:
        output_acgan = _linear.Linear('Discriminator.ACGANOutput', D, 10, output2)
        return output_wgan, output2, output_acgan
    return output_wgan, output2, None


def Discriminator(inputs, labels, kp1, kp2, kp3, u=None, rng=None, heads=('wgan', 'acgan')):
    """:169-186 - returns (D [n], D_ [n,DIM_D], acgan logits [n,10] or None).
    `u` = the three dropout uniforms [n,DIM_D,8,8] (explicit draws); else drawn from `rng`."""
    return DiscriminatorTail(DiscriminatorTrunk(inputs), kp1, kp2, kp3, u=u, rng=rng, heads=heads)


def build_params(device=None):
    """Instantiate every parameter once (the reference does this while building its graph)."""
    if device is not None:
        lib.set_device(device)
    dev = lib._dev()
    lab = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.no_grad():
        x = Generator(2, lab, noise=torch.zeros(2, 128, device=dev))
        Discriminator(x, lab, 1.0, 1.0, 1.0)


def _cat_rows(a, b):
    return torch.cat([a, b], dim=0)


# A/B switch: the tail's dropouts (and the final ReLU) inside the neighbouring conv kernels (see DiscriminatorTail)
DROP_FUSION = _os.environ.get('CTGAN_DROP_FUSION', '1') != '0'
# A/B switch: the generator's upsampled 1x1 shortcut is read at low resolution by the epilogue of the block's last conv
RESID_UP_FUSION = _os.environ.get('CTGAN_RESID_UP', '1') != '0'


# A/B switch: the critic's output head (mean, both Linear layers) fused around the loss heads (F.critic_tail_heads) and the
# gradient-penalty branch started from dD/dz of the last block directly (F.gp_head_grad)
HEAD_FUSION = _os.environ.get('CTGAN_HEAD_FUSION', '1') != '0'


# A/B switch: blocks 1-2 of the critic run once on [real ; fake ; x_hat]; the dropout passes and the gradient-penalty pass
# build their autograd graphs on row ranges of that one forward (functional.tape_record / tape_replay)
TRUNK_SHARE = _os.environ.get('CTGAN_TRUNK_SHARE', '1') != '0'
# A/B switch: blocks 3-4 of every pass of a critic step (dropout passes, clean pass, GP pass) in one set of forward launches
# with per-row-range dropout (shared_tail_forward); needs TRUNK_SHARE
TAIL_SHARE = _os.environ.get('CTGAN_TAIL_SHARE', '1') != '0'
# the penalty's batch mean and the clean pass's class head + accuracies inside the two launches of the fused loss heads (experiment switch)
HEADS_FOLD = _os.environ.get('CTGAN_HEADS_FOLD', '1') != '0'
# A/B switch: dequantisation, interpolation and the [real ; fake] concat of a critic step in one launch
PREP_FUSION = _os.environ.get('CTGAN_PREP_FUSION', '1') != '0'


# Batch-sharded steps: the critic's gradient bucket in two parts, the first one all-reduced under the rest of the backward (Trainer.split_flush)
SPLIT_FLUSH = _os.environ.get('CTGAN_SPLIT_FLUSH', '0') == '1'


# Draw the fake batches of all N_CRITIC critic steps of an iteration in one generator forward (Trainer.generate_fakes)
BATCH_FAKES = _os.environ.get('CTGAN_BATCH_FAKES', '1') != '0'


def _heads_fusable(rnd, rng):
    return (HEAD_FUSION and rnd is None and _critic_piecewise_linear() and _tail_fusable(0.8, 0.5, 0.5, None, rng)
            and cfg.DIM_D % 4 == 0 and cfg.DIM_D <= 1024)


def shared_tail_forward(h_all, B, rng, with_clean):
    """Blocks 3-4 of the critic for every pass of a critic step in ONE set of forward launches (F.tape_record): rows of the
    result are [real, fake, real' (dropout passes, specs `main`) | x_hat (gradient-penalty pass, specs `gp`) | real, fake (clean
    pass, no dropout)], h_all = the shared trunk output [real ; fake ; x_hat].  Each range's dropout masks are those of its own
    tensor (ctgan_epilogue_ext row ranges), so the passes replay rows of this forward and keep their separate backward graphs -
    or (critic_schedule.py) share ONE backward over the leading 4B rows: the rows that carry a gradient come first.
    -> (tape, gp_specs, main_specs, ranges = {'main': (r0, r1), 'clean': ..., 'gp': ...})"""
    D = cfg.DIM_D
    kps = (0.8, 0.5, 0.5)
    gp_specs = tuple(F.drop_spec(rng, kp) for kp in kps)            # call-site order of the unshared step: GP pass first
    main_specs = tuple(F.drop_spec(rng, kp) for kp in kps)
    n_main, n_clean = 3 * B, (2 * B if with_clean else 0)
    r_gp = n_main
    r_clean = n_main + B
    segs = [(0, 2 * B, kps[0], main_specs[0][2], 0), (0, B, kps[0], main_specs[0][2], 0)]
    segs.append((2 * B, B, kps[0], gp_specs[0][2], r_gp))
    if with_clean:
        segs.append((0, 2 * B, 1.0, 0, r_clean))

    def ranged(i):
        rs = [(n_main, main_specs[i]), (r_gp + B, gp_specs[i])]
        if with_clean:
            rs.append((r_clean + n_clean, None))
        return {'ranges': rs}

    with torch.no_grad(), F.tape_record() as tape:
        tin = K.rows_gather_dropout(h_all, segs, rng.seed, rng.ctr)
        tape.append(tin)
        out = ResidualBlock('Discriminator.3', D, D, 3, tin, resample=None, out_epi={'out_drop': ranged(1)})
        ResidualBlock('Discriminator.4', D, D, 3, out, resample=None, out_epi={'out_drop': ranged(2), 'out_relu': True})
    return tape, gp_specs, main_specs, {'main': (0, n_main), 'clean': (r_clean, r_clean + n_clean), 'gp': (r_gp, r_gp + B)}


def gradient_penalty_branch(interp, labels, rng, rnd=None, trunk_tape=None, tail_tape=None, specs=None, defer_mean=False):
    """GP = lambda * mean((||dD(x_hat)/dx_hat||_2 - 1)^2) with its own dropout masks (:277-286): critic forward on x_hat,
    data gradient back to x_hat under create_graph.  -> (gp, slopes, dD/dx_hat).  interp must require grad.
    defer_mean: gp is returned as a slot that the caller's F.critic_tail_heads(..., gp, slopes=slopes, gp_lambda=...) fills."""
    fuse_heads = _heads_fusable(rnd, rng)
    with F.weight_grads(not _critic_piecewise_linear()):
        if fuse_heads:
            if trunk_tape is not None:       # the trunk's forward launches were shared with the dropout passes (F.tape_record)
                with F.tape_replay(*trunk_tape):
                    h_gp = DiscriminatorTrunk(interp)
            else:
                h_gp = DiscriminatorTrunk(interp)
            if tail_tape is not None:
                with F.tape_replay(*tail_tape):
                    y_gp = DiscriminatorTailBody(h_gp, 0.8, 0.5, 0.5, rng=rng, mask_done=True, specs=specs)
            else:
                y_gp = DiscriminatorTailBody(h_gp, 0.8, 0.5, 0.5, rng=rng, mask_done=True)
        else:
            u_gp = rnd['u_gp'] if rnd is not None else None
            d_gp = Discriminator(interp, labels, 0.8, 0.5, 0.5, u=u_gp, rng=rng, heads=('wgan',))[0]
    if fuse_heads:
        # D(x_hat) itself is never used: start the backward at the last block with dD/dz (one launch)
        gz = F.gp_head_grad(y_gp, lib.param('Discriminator.Output.W'), 1.0 / 0.5)
        (grads,) = torch.autograd.grad(y_gp, interp, grad_outputs=gz, create_graph=True)
    else:
        ones = torch.ones_like(d_gp)
        (grads,) = torch.autograd.grad(d_gp, interp, grad_outputs=ones, create_graph=True)
    gp, slopes = F.gradient_penalty(grads, cfg.GP_LAMBDA, defer_mean=defer_mean)
    return gp, slopes, grads


class Trainer:
    """Owns the optimizers, the random streams and the D/G step (the session of the reference)."""

    def __init__(self, seed=2024, rank=0, world_size=1, allreduce=None):
        self.dev = lib._dev()
        self.rank, self.world = rank, world_size
        self.allreduce = allreduce            # callable(flat_tensor) -> None, sums across ranks (ddp.py)
        self.rng = DeviceRNG(seed, rank, self.dev)
        self.d_named = lib.named_params_with_name('Discriminator.', trainable_only=True)
        self.g_named = lib.named_params_with_name('Generator', trainable_only=True)
        self._opt_state = torch.zeros(8, dtype=torch.float32, device=self.dev)      # both optimizers' {lr, b1^t, b2^t, -}: set_lr()
        self.d_opt = FlatAdam(self.d_named, 0.0, 0.9, state=self._opt_state[0:4])
        self.g_opt = FlatAdam(self.g_named, 0.0, 0.9, state=self._opt_state[4:8])
        self.d_params = [p for _, p in self.d_named]
        self.g_params = [p for _, p in self.g_named]
        self._one = None
        # Batch-sharded steps: hand the critic's gradient bucket to the all-reduce in two parts - blocks 1-2 (a prefix of the flat bucket)
        # as soon as the hand-scheduled step has completed them, under the rest of the penalty's double backward (north star: "all-reduce
        # overlapped with backward").  A switch, off by default: a second grouped weight-gradient launch and a second collective against
        # ~0.15 ms of overlap on a 4 MB bucket; bench.py --split-flush reports the exposed time of both forms.
        self.split_flush = SPLIT_FLUSH
        names = [n for n, _ in self.d_named]
        self._n_early = 0
        while self._n_early < len(names) and names[self._n_early].startswith(('Discriminator.1.', 'Discriminator.2.')):
            self._n_early += 1

    def cost_seed(self, cost):
        """grad_outputs of a step's backward: a cached tensor of ones (autograd otherwise launches a fill kernel per step)."""
        if self._one is None or self._one.device != cost.device:
            self._one = torch.ones((), dtype=torch.float32, device=cost.device)
        return self._one.reshape(cost.shape)

    # ------------------------------------------------------------------ losses
    def d_losses(self, real_int, labels, rnd=None, fake=None):
        """Critic loss graph :194-305.  `rnd` (parity mode) injects every random draw; see
        oracle/steps.make_rnd_resnet_d for the keys and shapes."""
        B = cfg.BATCH_SIZE
        rng = self.rng
        tape = None
        with torch.no_grad():
            if fake is None:      # `fake`: samples drawn earlier from the SAME generator weights (generate_fakes)
                z = torch.cat(rnd['z'], 0) if rnd is not None else None
                fake = Generator(B, labels, noise=z, groups=2, rng=rng)
            if PREP_FUSION and rnd is None and cfg.OUTPUT_DIM % 4 == 0 and real_int.is_contiguous() and fake.is_contiguous():
                # dequantised reals, x_hat and the [real ; fake] batch in one launch (same Philox call sites as below)
                rf, interp, both = K.critic_prep(real_int, fake, rng.seed, rng._sid(), rng._sid(), rng.ctr, 0.0, 1. / 128, 256.0)
                real = rf[:B]
                if TRUNK_SHARE and not cfg.NORMALIZATION_D and _heads_fusable(rnd, rng):
                    # blocks 1-2 once for [real ; fake ; x_hat]: the two passes below replay rows of it (F.tape_record)
                    with F.tape_record() as tape:
                        DiscriminatorTrunk(both)
            else:
                deq = rnd['dequant'] if rnd is not None else rng.uniform(B, cfg.OUTPUT_DIM, lo=0.0, hi=1. / 128)
                real = K.real_prep(real_int, deq, 256.0)
                alpha = rnd['alpha'] if rnd is not None else rng.uniform(B, 1)
                interp = K.interpolate(real, fake, alpha)
                rf = _cat_rows(real, fake)

        # gradient penalty :277-286, issued FIRST.  The critic is piecewise linear (no normalisation in D), so the penalty reaches the
        # weights only through the backward ops: skip the forward's own wgrads.  (The branch on a side stream - parallel hipGraph
        # branches, so that the penalty's 64-row double backward overlaps the main pass's backward - was measured twice: -1 % in round 1,
        # -2.5 % in round 3 (16.47 vs 16.06 ms on one box): removed.)
        interp.requires_grad_(True)
        fuse_heads = _heads_fusable(rnd, rng)
        use_ac = cfg.CONDITIONAL and cfg.ACGAN
        tail = None
        if TAIL_SHARE and tape is not None and fuse_heads:
            tail = shared_tail_forward(tape[-1], B, rng, with_clean=use_ac)
        gp, slopes, grads = gradient_penalty_branch(interp, labels, rng, rnd, trunk_tape=(tape, 2 * B, 3 * B) if tape is not None else None,
                                                        tail_tape=(tail[0],) + tail[3]['gp'] if tail is not None else None,
                                                        specs=tail[1] if tail is not None else None, defer_mean=fuse_heads and HEADS_FOLD)

        # dropout passes 1 and 2 share the trunk; pass 2 is needed on the real half only
        if tape is not None:
            with F.tape_replay(tape, 0, 2 * B):
                h = DiscriminatorTrunk(rf)
        else:
            h = DiscriminatorTrunk(rf)
        if rnd is not None:
            u = [_cat_rows(a, b[:B]) for a, b in zip(rnd['u_pass1'], rnd['u_pass2'])]
        else:
            u = None
        out = {}
        use_ac = cfg.CONDITIONAL and cfg.ACGAN
        y_clean = None
        if fuse_heads:
            # mean + both Linear heads + every loss head: two launches forward, one backward (gradient w.r.t. the last conv's
            # result and the head weights)
            if tail is not None:
                with F.tape_replay(tail[0], *tail[3]['main']):
                    y = DiscriminatorTailBody(h, 0.8, 0.5, 0.5, rng=rng, mask_done=True, cat_extra=B, specs=tail[2])
            else:
                y = DiscriminatorTailBody(h, 0.8, 0.5, 0.5, rng=rng, mask_done=True, cat_extra=B)
            P = lib.param
            # the penalty's batch mean (slopes -> gp) and, with the shared tail, the clean pass's class head + accuracies (:249-266)
            # ride the two launches of the heads
            y_clean = None
            if HEADS_FOLD and use_ac and tail is not None:
                c0, c1 = tail[3]['clean']
                y_clean = tail[0][-1][c0:c1].detach()
            cost, wgan, ct, acgan, disc_wgan, d_all, acc = F.critic_tail_heads(
                y, P('Discriminator.Output.W'), P('Discriminator.Output.b'),
                P('Discriminator.ACGANOutput.W') if use_ac else None, P('Discriminator.ACGANOutput.b') if use_ac else None,
                labels, B, cfg.LAMBDA_2, cfg.Factor_M, cfg.ACGAN_SCALE if use_ac else 0.0, 1.0 / 0.5, gp,
                slopes if HEADS_FOLD else None, cfg.GP_LAMBDA, y_clean, False)
        else:
            tail_in = _cat_rows(h, h[:B])
            d_all, f_all, a_all = DiscriminatorTail(tail_in, 0.8, 0.5, 0.5, u=u, rng=rng)
            # every loss head of the two dropout passes in one kernel (fwd) / one kernel (bwd): wgan :244, CT :288-291, ACGAN :246-248
            cost, wgan, ct, acgan, disc_wgan = F.critic_heads(d_all, f_all, a_all if use_ac else None, labels, B, cfg.LAMBDA_2,
                                                              cfg.Factor_M, cfg.ACGAN_SCALE if use_ac else 0.0, gp)
        if use_ac and fuse_heads and y_clean is not None:
            out['acc_real'], out['acc_fake'] = acc[0], acc[1]
        elif use_ac:
            with torch.no_grad():                                            # clean pass: accuracies only :228,249-266
                if tail is not None:       # rows of the shared tail forward (ReLU already applied, no dropout in this range)
                    c0, c1 = tail[3]['clean']
                    _, _, a_clean = K.tail_heads_fwd(tail[0][-1][c0:c1], None, None, lib.param('Discriminator.ACGANOutput.W'),
                                                     lib.param('Discriminator.ACGANOutput.b'), relu=False)
                elif fuse_heads:       # relu + mean + Linear in one launch
                    D = cfg.DIM_D
                    yc = ResidualBlock('Discriminator.4', D, D, 3, ResidualBlock('Discriminator.3', D, D, 3, h.detach(), resample=None),
                                       resample=None)
                    if not yc.permute(0, 2, 3, 1).is_contiguous():
                        yc = F.to_channels_last(yc)
                    _, _, a_clean = K.tail_heads_fwd(yc, None, None, lib.param('Discriminator.ACGANOutput.W'),
                                                     lib.param('Discriminator.ACGANOutput.b'), relu=True)
                else:
                    _, _, a_clean = DiscriminatorTail(h.detach(), 1.0, 1.0, 1.0, heads=('acgan',))
                acc = K.accuracy2(a_clean.contiguous(), labels, B)
            out['acc_real'], out['acc_fake'] = acc[0], acc[1]
        else:
            acgan = None
        out.update(cost=cost, wgan=disc_wgan, acgan=acgan, wgan_only=wgan, ct=ct, gp=gp, slopes=slopes, fake=fake,
                   real=real, d_real=d_all[:B], d_fake=d_all[B:2 * B], gp_grads=grads)
        return out

    def early_reduce(self, gpart):
        """`early` hook of critic_schedule.critic_step: the finished gradients of blocks 1-2 -> the prefix of the flat bucket -> their
        all-reduce, asynchronous on the side stream (ddp.FlatAllReduce).  Returns the number of parameters handed over."""
        opt, n = self.d_opt, self._n_early
        part = opt.gather_grads([gpart.get(name) for name in opt.names[:n]] + [None] * (len(opt.names) - n), 0, n)
        self.allreduce(part)
        return n

    def d_grads(self, real_int, labels, rnd=None, fake=None, early=None):
        """Losses and parameter gradients of one critic step -> (out, grads aligned with self.d_params): compute_gradients(disc_cost) of
        :335-336.  Default: the hand-scheduled step (critic_schedule.py - one backward chain over the rows of the dropout passes and of the
        gradient-penalty pass); the autograd path (d_losses + torch.autograd.grad) wherever that schedule does not apply - injected draws
        (parity mode), a Layernorm critic, widths outside the few-channel kernels, any fusion switch off."""
        from . import critic_schedule as CS
        if fake is None and rnd is None:
            with torch.no_grad():      # (the draw d_losses would make first: same Philox call sites either way)
                fake = Generator(cfg.BATCH_SIZE, labels, groups=2, rng=self.rng)
        if CS.usable(_this_module(), rnd, self.rng, real_int, fake):
            with torch.no_grad(), F.deferred_wgrads():
                return CS.critic_step(self, _this_module(), real_int, labels, fake, early=early)
        out = self.d_losses(real_int, labels, rnd, fake=fake)
        with F.deferred_wgrads():
            grads = torch.autograd.grad(out['cost'], self.d_params, grad_outputs=self.cost_seed(out['cost']), allow_unused=True)
        return out, grads

    def g_losses(self, rnd=None):
        """Generator loss graph :314-330: two towers of GEN_BS_MULTIPLE*B/2 samples."""
        n = cfg.GEN_BS_MULTIPLE * cfg.BATCH_SIZE
        rng = self.rng
        if rnd is not None:
            fake_labels = torch.cat([(lu * 10).to(torch.int32) for lu in rnd['label_u']], 0)
            z = torch.cat(rnd['z'], 0)
            u = [torch.cat([rnd['u'][0][i], rnd['u'][1][i]], 0) for i in range(3)]
        else:
            fake_labels = rng.labels(n, 10)
            z, u = None, None
        x = Generator(n, fake_labels, noise=z, groups=2, rng=rng)
        use_ac = cfg.CONDITIONAL and cfg.ACGAN
        with F.weight_grads(False):                      # only dD/dx is needed from the critic
            if _heads_fusable(rnd, rng):
                # mean + both Linear heads + the loss: two launches forward, one backward (F.gen_tail_heads)
                y = DiscriminatorTailBody(DiscriminatorTrunk(x), 0.8, 0.5, 0.5, rng=rng, mask_done=True)
                cost, _ = F.gen_tail_heads(y, lib.param('Discriminator.Output.W'), lib.param('Discriminator.Output.b'),
                                           lib.param('Discriminator.ACGANOutput.W') if use_ac else None,
                                           lib.param('Discriminator.ACGANOutput.b') if use_ac else None, fake_labels,
                                           cfg.ACGAN_SCALE_G if use_ac else 0.0, 1.0 / 0.5)
                return {'cost': cost, 'samples': x}
            d, _, a = Discriminator(x, fake_labels, 0.8, 0.5, 0.5, u=u, rng=rng)
        cost = F.mean_diff(d, n, 0, -1.0, 0.0)
        if use_ac:
            ce, _ = F.softmax_cross_entropy(a, fake_labels)
            cost = cost + cfg.ACGAN_SCALE_G * ce
        return {'cost': cost, 'samples': x}

    # ------------------------------------------------------------------ steps
    def lr(self, iteration):
        decay = max(0., 1. - float(iteration) / cfg.ITERS) if cfg.DECAY else 1.
        return cfg.LR * decay

    def set_lr(self, lr):
        """The (common, :333-338) learning rate of both optimizers in ONE fill of their shared state allocation."""
        lr = float(lr)
        if self.d_opt._lr_last == lr and self.g_opt._lr_last == lr:
            return
        self._opt_state[0::4].fill_(lr)
        self.d_opt._lr_last = self.g_opt._lr_last = lr

    def generate_fakes(self, labels_all):
        """The fake batches of the next len(labels_all)/B critic steps in ONE generator forward.  The generator does
        not change between the N_CRITIC critic updates of an iteration (:393-404), so drawing their fake batches
        together is the same computation as drawing them one per step - each critic step's batch keeps its own two
        BN statistic groups (the reference's two towers, :207-213) - at 5x the rows per kernel launch."""
        B = cfg.BATCH_SIZE
        n = labels_all.shape[0]
        assert n % B == 0
        F.prepare_filters()
        self.rng.begin_step()
        with torch.no_grad():
            fake = Generator(n, labels_all, groups=2 * (n // B), rng=self.rng)
        self.rng.end_step()
        return fake.reshape(n // B, B, cfg.OUTPUT_DIM)

    def d_step(self, real_int, labels, rnd=None, iteration=0, set_lr=True, fake=None):
        """session.run([..., disc_train_op]) :402"""
        F.prepare_filters()
        self.rng.begin_step()
        handed = [0]
        early = None
        if self.split_flush and self.world > 1 and self.allreduce is not None and self._n_early:
            def early(gpart):
                handed[0] = self.early_reduce(gpart)
        out, grads = self.d_grads(real_int, labels, rnd, fake=fake, early=early)
        self._apply(self.d_opt, grads, iteration, set_lr, lo=handed[0])
        out['grads'] = dict(zip([n for n, _ in self.d_named], grads))
        return out

    def g_step(self, rnd=None, iteration=0, set_lr=True):
        """session.run([gen_train_op]) :397"""
        F.prepare_filters()
        self.rng.begin_step()
        out = self.g_losses(rnd)
        with F.deferred_wgrads():
            grads = torch.autograd.grad(out['cost'], self.g_params, grad_outputs=self.cost_seed(out['cost']), allow_unused=True)
        self._apply(self.g_opt, grads, iteration, set_lr)
        out['grads'] = dict(zip([n for n, _ in self.g_named], grads))
        return out

    def _apply(self, opt, grads, iteration, set_lr, lo=0):
        """lo: the first `lo` parameters' gradients are already in the bucket and on their way through the all-reduce (early_reduce)."""
        if set_lr:
            opt.set_lr(self.lr(iteration))
        if self.allreduce is None or self.world <= 1:
            # single rank: gradient bucket + Adam in one launch, the end of the step (beta powers, Philox step counter) in another
            opt.update(grads, 1.0 / self.world, rng=self.rng)
            return
        flat = opt.gather_grads(grads, lo)
        self.reduce_and_update(opt, flat)

    def reduce_and_update(self, opt, flat, between=None, end_rng=True):
        """All-reduce the flat gradient bucket (asynchronous on the side stream, ddp.FlatAllReduce), run `between()` - work
        that does not depend on the update, e.g. staging the next step's inputs - while it is in flight, then Adam with
        the 1/world average folded in.  The update launch also ends the step (FlatAdam.step(rng=...)) unless the caller's captured
        graph already advanced the Philox counter (end_rng=False)."""
        ar = self.allreduce if (self.allreduce is not None and self.world > 1) else None
        if ar is not None:
            ar(flat)
        if between is not None:
            between()
        if ar is not None and hasattr(ar, 'wait'):
            ar.wait()
        opt.step(grad_scale=1.0 / self.world, rng=self.rng if end_rng else None)

    def train_iteration(self, iteration, next_batch):
        """One pass of the loop body :393-404: [G step if it>0] then N_CRITIC x (batch, D step)."""
        if iteration > 0:
            self.g_step(iteration=iteration)
        out = None
        if not BATCH_FAKES:
            for _ in range(cfg.N_CRITIC):
                data, labels = next_batch()
                out = self.d_step(data, labels, iteration=iteration)
            return out
        batches = [next_batch() for _ in range(cfg.N_CRITIC)]
        fakes = self.generate_fakes(torch.cat([lab for _, lab in batches], 0))
        for i, (data, labels) in enumerate(batches):
            out = self.d_step(data, labels, iteration=iteration, fake=fakes[i])
        return out

    def generate_samples(self, noise, labels):
        """fixed_noise_samples / generate_image :341-348: int pixels = ((s+1)*255/2) truncated."""
        with torch.no_grad():
            s = Generator(noise.shape[0], labels, noise=noise)
        return s, ((s + 1.) * (255. / 2)).to(torch.int32)


def train(data_dir, n_examples=50000, iters=None, out_dir='.', seed=2024, use_graphs=True, sample_every=100,
          checkpoint_every=1000, resume=None, log=print):
    """The module-level training loop of the reference (TF/CT_gan_cifar_resnet.py:350-434) minus the Inception
    score (needs the 2015 Inception graph + network): CIFAR-10 generator factories, `time` /
    `cost` / `wgan` / `acgan` / `acc_real` / `acc_fake` series (train_log.Series), fixed-noise sample grids every
    `sample_every` iterations (:341-348, :429), checkpoints every `checkpoint_every` (build-only)."""
    import os
    import time

    from . import checkpoint
    from .engine import GraphedTrainer
    from .tflib import cifar10, save_images
    from .train_log import Series
    iters = cfg.ITERS if iters is None else iters
    build_params()
    trainer = Trainer(seed=seed)
    start = checkpoint.load(resume, trainer) if resume else 0
    eng = GraphedTrainer(trainer, use_graphs=use_graphs)
    train_gen, dev_gen = cifar10.load(cfg.BATCH_SIZE, data_dir, n_examples)
    feed = cifar10.prefetch_to_device(cifar10.inf_train_gen(train_gen), trainer.dev)
    fixed_noise = trainer.rng.normal(100, 128)
    fixed_labels = torch.arange(10, dtype=torch.int32, device=trainer.dev).repeat(10)
    series = Series(os.path.join(out_dir, 'log.jsonl'), echo=log)
    for iteration in range(start, iters):
        t0 = time.time()
        out = eng.train_iteration(iteration, lambda: next(feed))
        series.add('cost', out['cost'].item())
        if out.get('acgan') is not None:
            for k in ('wgan', 'acgan', 'acc_real', 'acc_fake'):
                series.add(k, out[k].item())
        series.add('time', time.time() - t0)
        if iteration % sample_every == sample_every - 1:
            _, px = trainer.generate_samples(fixed_noise, fixed_labels)
            save_images.save_images(px.reshape(100, 3, 32, 32).cpu().numpy(), os.path.join(out_dir, 'samples_%d.png' % iteration))
        if checkpoint_every and iteration % checkpoint_every == checkpoint_every - 1:
            checkpoint.save(os.path.join(out_dir, 'checkpoint.pt'), trainer, iteration + 1)
        if iteration < 500 or iteration % 1000 == 999:
            series.flush()
        series.tick()
    return trainer


def _critic_piecewise_linear():
    return not cfg.NORMALIZATION_D


def _this_module():
    import sys
    return sys.modules[__name__]
