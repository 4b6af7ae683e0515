"""Differentiable operators built on the HIP kernels (autograd wiring only - no arithmetic here).

The gradient penalty needs the gradient of a gradient (`tf.gradients` inside the loss,
TF/CT_gan_cifar_resnet.py:284-286), so every op on the critic path has a backward that is itself
made of differentiable ops: conv / conv-dgrad / conv-wgrad form a closed family, the mask ops
(ReLU, dropout), pool/upsample and spatial mean/broadcast are linear maps whose adjoints are
again members of the set.

`weight_grads(False)` marks a forward whose parameter gradients are not wanted (the critic inside
the generator step; the critic pass that only feeds the gradient penalty, where the critic is
piecewise linear so the penalty reaches the weights through the backward ops alone).
"""
import contextlib

import torch
from torch.autograd import Function

from . import kernels as K
from .kernels import ConvGeom

_WEIGHT_GRADS = True
# ReLU backward of conv(relu(x)): True = mask read inside the dgrad epilogue (16-B loads issued ahead of the
# stores), False = stand-alone mask kernel on the data gradient.  A/B switch CTGAN_MASK_EPI.
import os as _os
MASK_IN_DGRAD_EPILOGUE = _os.environ.get('CTGAN_MASK_EPI', '1') != '0'
# Double backward of the gradient penalty: a data gradient g_a = mask(a) * conv^T(g_y) multiplies whatever arrives for g_a by mask(a)
# again.  When g_a's consumer is itself a data-gradient node (the block's first conv), the conv that node launches in the double
# backward takes mask(a) in its epilogue and the separate mask pass (one launch per block per step) disappears.  Experiment switch.
PREMASK_FUSION = _os.environ.get('CTGAN_PREMASK', '1') != '0'
# Mixed-precision modes: queue the small weight gradients for the grouped 16-bit launch (kernels.grouped16_takes).  The size rule is
# applied PER FILTER, by its first use of the step: a filter must not get one queued and one immediate result - two gradient tensors for
# one parameter, which autograd sums the moment the second arrives, before the flush has written the queued one (_wgrad keeps the
# first use's choice for every later use).  CTGAN_DEFER_16BIT=0: every weight gradient of those modes at once, as in round 3.
DEFER_16BIT = _os.environ.get('CTGAN_DEFER_16BIT', '1') == '1'


class _PreMask:
    """Link between the node that produced a masked data gradient and the node that consumes it.  The token travels ON the tensor
    (attribute `_ctgan_premask` of the producer's result; `_ctgan_fused_for` of the consumer's) - not in a table keyed by device address:
    a tensor that passed through anything else (a view, an accumulation by the autograd engine) is another Python object without the
    attribute and simply does not fuse, whatever the allocator did with the addresses in between.  `fused_done` = the consumer's conv
    already applied this node's mask to what it hands back."""
    __slots__ = ('mask', 'fused_done')


@contextlib.contextmanager
def weight_grads(enabled):
    """Forward passes inside this context record (enabled) or skip (not enabled) parameter grads."""
    global _WEIGHT_GRADS
    old = _WEIGHT_GRADS
    _WEIGHT_GRADS = enabled
    try:
        yield
    finally:
        _WEIGHT_GRADS = old


# ---- deferred weight gradients ------------------------------------------------------------------------------
# Inside `deferred_wgrads()` (wrapped around the first-order torch.autograd.grad of a step) a conv's weight gradient is
# not launched where autograd asks for it: the (x, dy) pair is queued under its filter, the FIRST request of a filter
# returns the (still empty) result buffer, later requests of the same filter return None, and on exit every filter
# gets ONE multi-segment launch (K.conv_wgrad_multi) that sums its uses - the dropout passes and the gradient-penalty
# double backward - instead of one wgrad + reduction per use and an autograd `add` per extra use.
_DEFER = {'on': False, 'groups': None, 'post': None, 'imm': None, 'join': None}      # imm: filters whose first use of the step was launched at once
# A/B switch: the hand-scheduled critic step launches the weight gradients of the dropout-pass rows on a SIDE stream as soon as its backward
# chain has produced them, under the (latency-bound, 64-row) launches of the penalty's double backward (flush_async)
WGRAD_OVERLAP = _os.environ.get('CTGAN_WGRAD_OVERLAP', '0') == '1'
_SIDE = {}
DEFER_WGRADS = _os.environ.get('CTGAN_DEFER_WGRADS', '1') != '0'
# A/B switch: the few-channel weight gradients (first critic conv / shortcut) are queued too: both uses of a filter in one launch
FEWCH_DEFER = _os.environ.get('CTGAN_FEWCH_DEFER', '1') != '0'
# A/B switch: all queued weight gradients in one launch per tile configuration + one launch for their reductions
WGRAD_GROUPED = _os.environ.get('CTGAN_WGRAD_GROUPED', '1') != '0'


class _WgradGroup:
    # inplace / inplace_b: dw / db already HOLD a finished addend - the result of the first use the split-mode kernel took at request
    # time (_wgrad), written straight into the filter's buffers; the flush then accumulates the queued segments onto it inside the
    # batched split-K reduction (no axpby / copy launches).  pre: (gw, gb) results of further such uses (rare), added at the flush.
    # Either way autograd sees ONE buffer per filter, complete when the deferred_wgrads() block exits.
    __slots__ = ('g', 'segs', 'dw', 'db', 'pre', 'inplace', 'inplace_b')


def _new_group(g, device):
    grp = _WgradGroup()
    grp.g, grp.segs, grp.db, grp.pre, grp.inplace, grp.inplace_b = g, [], None, [], False, False
    grp.dw = torch.empty((g.R, g.S, g.C, g.K), dtype=torch.float32, device=device)
    return grp


@contextlib.contextmanager
def deferred_wgrads():
    if not DEFER_WGRADS or _DEFER['on']:
        yield
        return
    _DEFER.update(on=True, groups={}, post=[], imm=set(), join=None)
    try:
        yield
    finally:
        groups, post, join = _DEFER['groups'], _DEFER['post'], _DEFER['join']
        _DEFER.update(on=False, groups=None, post=None, imm=None, join=None)
        if join is not None and join[0] is not None:       # the early launch on the side stream (flush_async) wrote the buffers the flush below accumulates onto
            torch.cuda.current_stream().wait_event(join[0])
        _flush_groups(list(groups.values()))
        join = None                             # (the early launch's operands were kept alive until here)
        folds = [e[1:] for e in post if isinstance(e, tuple) and e[0] == 'fold']
        if folds:
            K.filter_fold_batch(folds)               # every spread-filter gradient of the step folded in one launch
        for fn in post:
            if not isinstance(fn, tuple):
                fn()


def flush_partial(keys):
    """Inside deferred_wgrads(): flush NOW the queued weight gradients of the filters `keys` (the (filter address, geometry) keys of
    _wgrad) - one grouped launch for them - and fold the spread-filter gradients that belong to them; the rest of the queue stays for
    the flush at the end of the block.  For a step that hands part of its gradient bucket to the all-reduce early
    (critic_schedule.critic_step(early=...): blocks 1-2 of the critic are complete while the penalty's double backward is still in blocks
    3-4).  Every use of those filters must have been queued already: a later request would open a second result buffer."""
    assert _DEFER['on']
    groups = _DEFER['groups']
    part = [groups.pop(k) for k in keys if k in groups]
    if not part:
        return
    if _DEFER['join'] is not None and _DEFER['join'][0] is not None:
        torch.cuda.current_stream().wait_event(_DEFER['join'][0])         # (flush_async's launch wrote what this one accumulates onto)
    _flush_groups(part)
    bufs = {id(g.dw) for g in part}
    post, keep = _DEFER['post'], []
    folds = []
    for e in post:
        if isinstance(e, tuple) and e[0] == 'fold' and id(e[1]) in bufs:
            folds.append(e[1:])
        else:
            keep.append(e)
    post[:] = keep
    if folds:
        K.filter_fold_batch(folds)
    _DEFER['imm'].update(keys)          # a stray later use of one of these filters is launched at once instead of re-opening its queue


def flush_async():
    """Inside deferred_wgrads(): launch NOW, on a side stream, every weight gradient queued so far (the filters of the grouped MFMA launch;
    the few-channel filters stay queued), and keep their queues open: segments requested later are accumulated onto these results by the
    flush at the end of the block (the in-place form of the grouped launch, as for a first use the split-mode kernel took at request time).

    For a step whose remaining launches do not fill the chip: the hand-scheduled critic step calls it when its backward chain has produced
    the (x, dy) pairs of the dropout-pass rows - 3B / 2B rows per filter, ~0.24 ms of matrix work - and the penalty's double backward, a
    dependent chain of ~20 launches on the B x_hat rows (0.3 ms, none of them wider than 256 workgroups), is still to come: the two
    run side by side instead of one after the other.  Under hipGraph capture the side stream is forked into the capture (event wait) and
    joined by the final flush.  No-op outside CUDA / with the switch off."""
    if not (_DEFER['on'] and WGRAD_OVERLAP) or _DEFER['join'] is not None:
        return False
    grps = [g for g in _DEFER['groups'].values() if g.segs and not g.pre and not K.fewch_handles(g.g)]
    if len(grps) < 2:
        return False
    dev = grps[0].segs[0][0].device
    if dev.type != 'cuda':
        _flush_groups(grps)                      # (the CPU stand-ins of the test suite: same two-part flush, no streams)
        done = None
    else:
        side = _SIDE.get(dev.index)
        if side is None:
            side = _SIDE[dev.index] = torch.cuda.Stream(device=dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            _flush_groups(grps)
            done.record(side)
    keep = []
    for g in grps:
        keep.append(g.segs)                      # operands stay referenced until the join: the allocator must not hand their memory to a
        g.inplace_b = g.inplace_b or _seg_bias(g)        # launch of the main stream while the side stream still reads it
        g.segs = []
        g.inplace = True
    _DEFER['join'] = (done, keep)
    return True


def _seg_bias(grp):
    return any(sg[3] for sg in grp.segs)


def _flush_groups(grps):
    """All queued weight gradients: one grouped launch per tile configuration + one reduction launch when every group
    fits the pipelined kernel (K.conv_wgrad_group), else group by group."""
    _all = list(grps)
    grps = [g for g in grps if g.segs]
    simple = [g for g in grps if len(g.segs) <= K.WGRAD_MAX_SEGS and not K.fewch_handles(g.g)]
    if WGRAD_GROUPED and len(simple) > 1:
        for g in simple:                                   # a queued bias buffer nothing contributes to
            if g.db is not None and not _seg_bias(g) and not g.inplace_b:
                g.db.zero_()
        done = []
        try:
            for i in range(0, len(simple), K.WGRAD_GROUP_LIMIT):
                part = simple[i:i + K.WGRAD_GROUP_LIMIT]
                K.conv_wgrad_group([(g.segs, g.g, g.dw, g.db if _seg_bias(g) else None,
                                     g.dw if g.inplace else None, g.db if (g.inplace_b and _seg_bias(g)) else None)
                                    for g in part])
                done.extend(part)
        except NotImplementedError:
            # K.conv_wgrad_group validates every member before its first launch: the call that raised launched nothing.  Calls that
            # completed (earlier parts of a queue longer than WGRAD_GROUP_LIMIT) stay done - re-running them would add an in-place
            # addend twice.
            pass
        grps = [g for g in grps if not any(g is d for d in done)]
    for grp in grps:
        _flush_group(grp)
    for grp in _all:
        dw_set = bool(grp.segs) or grp.inplace
        db_set = _seg_bias(grp) or grp.inplace_b
        for gw, gb in grp.pre:                              # further results the split-mode kernel produced at request time
            if dw_set:
                K.axpby(grp.dw, gw, 1.0, 1.0, out=grp.dw)
            else:
                grp.dw.copy_(gw)
            dw_set = True
            if gb is not None and grp.db is not None:
                if db_set:
                    K.axpby(grp.db, gb, 1.0, 1.0, out=grp.db)
                else:
                    grp.db.copy_(gb)
                db_set = True
        if grp.db is not None and not db_set:
            grp.db.zero_()


def _flush_group(grp):
    segs = grp.segs
    acc_w, acc_b = grp.inplace, grp.inplace_b               # the buffers already hold an addend: accumulate instead of overwriting

    def put(dw2, db2):
        nonlocal acc_w, acc_b
        if acc_w:
            K.axpby(grp.dw, dw2, 1.0, 1.0, out=grp.dw)
        elif dw2 is not grp.dw:
            grp.dw.copy_(dw2)
        acc_w = True
        if db2 is not None:
            if acc_b:
                K.axpby(grp.db, db2, 1.0, 1.0, out=grp.db)
            elif db2 is not grp.db:
                grp.db.copy_(db2)
            acc_b = True
    try:
        per = 2 if K.fewch_handles(grp.g) else K.WGRAD_MAX_SEGS
        for i in range(0, len(segs), per):
            chunk = segs[i:i + per]
            cb = any(sg[3] for sg in chunk)
            direct = not acc_w and not (cb and acc_b)
            dw2 = grp.dw if direct else torch.empty_like(grp.dw)
            db2 = (grp.db if direct else torch.empty_like(grp.db)) if cb else None
            K.conv_wgrad_multi(chunk, grp.g, dw2, db2)
            put(dw2, db2)
    except NotImplementedError:                                 # shape outside the pipelined kernel: one call per use
        for x, gy, relu_x, with_bias in segs:
            r = K.conv_wgrad(x, gy, grp.g, with_bias=with_bias, relu_x=relu_x)
            gw, gb = r if with_bias else (r, None)
            put(gw, gb)
    if grp.db is not None and not acc_b and not grp.pre:
        grp.db.zero_()


def _like_first(t, ref):
    """t with the per-sample strides of `ref` (same C,H,W; the batch sizes may differ)."""
    if t.stride()[1:] == ref.stride()[1:]:
        return t
    assert t.shape[1:] == ref.shape[1:]
    per_sample = t.shape[1] * t.shape[2] * t.shape[3]
    return K.copy4d(t, torch.empty_strided(t.shape, (per_sample,) + tuple(ref.stride()[1:]), dtype=t.dtype, device=t.device))


def _wgrad(x, gy, w, g, relu_x, with_bias):
    """Weight (and bias) gradient of one use of filter `w`: launched now, or queued (see deferred_wgrads).
    Returns (gw, gb); either may be None when another request of the same filter already owns the result."""
    stable = isinstance(w, torch.nn.Parameter) or w.data_ptr() in _SPREAD_BUFS or w.data_ptr() in _QUEUEABLE_GEMM     # same identity in every pass of the step
    fewch = K.fewch_handles(g)                    # few-channel convs: the direct kernel sums two uses in one launch
    gk = (g.C, g.H, g.W, g.K, g.R, g.S, g.stride)
    key = (w.data_ptr(), gk)       # (w: a registry parameter or a cached derived filter - persistent tensors, their addresses are stable)
    can = (_DEFER['on'] and stable and not torch.is_grad_enabled() and x.is_cuda == gy.is_cuda and not g.x_up
           and ((g.C % 32 == 0 and g.K % 4 == 0) or (fewch and FEWCH_DEFER)) and not (g.R == 1 and g.H == 1 and g.W == 1))
    if can and key in _DEFER['imm'] and key not in _DEFER['groups']:
        can = False                # launched at once earlier in this step (or flushed early, flush_partial): never re-open its queue
    if can and K.MMA_DTYPE is not None:
        # mixed-precision modes: the small problems the grouped 16-bit launch takes (kernels.grouped16_takes), everything else at once on
        # its own tile - decided by the filter's FIRST use of the step and kept for its later uses
        if key in _DEFER['groups']:
            can = x.is_cuda and K.grouped16_member(g)
        else:
            can = DEFER_16BIT and x.is_cuda and K.grouped16_takes(g, x.shape[0], x.stride())
            if not can:
                _DEFER['imm'].add(key)
    defer = can
    if not defer:
        if with_bias:
            return K.conv_wgrad(x, gy, g, with_bias=True, relu_x=relu_x)
        return K.conv_wgrad(x, gy, g, relu_x=relu_x), None
    if (with_bias or K.MMA_DTYPE is not None) and not gy.permute(0, 2, 3, 1).is_contiguous() and not fewch:
        gy = K.to_channels_last(gy)          # (the grouped 16-bit launch reads dy dense channels-last)
    grp = _DEFER['groups'].get(key)
    if K.wgrad_prefers_x3(g, x.shape[0], x.device):
        # a large layer: the split-mode kernel is the faster fp32 path (kernels.X3_HYBRID) - launched now.  The filter keeps ONE result
        # buffer: the first such use writes straight into it (bias gradient fused into the same launch), the queued segments of the
        # filter are accumulated onto it by the flush; further uses of this kind are added at the flush.
        gw = gb = None
        if grp is None:
            grp = _new_group(g, x.device)
            gw = grp.dw
            _DEFER['groups'][key] = grp
        if with_bias and grp.db is None:
            grp.db = gb = torch.empty(g.K, dtype=torch.float32, device=x.device)
        if not grp.inplace and not grp.pre:
            K.conv_wgrad(x, gy, g, with_bias=with_bias, relu_x=relu_x, out=(grp.dw, grp.db if with_bias else None))
            grp.inplace, grp.inplace_b = True, bool(with_bias)
        else:
            r = K.conv_wgrad(x, gy, g, with_bias=with_bias, relu_x=relu_x)
            grp.pre.append(r if with_bias else (r, None))
        return gw, gb
    if grp is not None and grp.segs:
        # A later use whose operands have another memory layout is repacked into the first use's layout.  It must NOT open a
        # second queue: that would hand autograd a second, still unfilled, buffer for the same parameter, and the engine
        # sums the two the moment the second arrives - before the flush has written either.
        x, gy = _like_first(x, grp.segs[0][0]), _like_first(gy, grp.segs[0][1])
    gw = gb = None
    if grp is None:
        grp = _new_group(g, x.device)
        gw = grp.dw
        _DEFER['groups'][key] = grp
    if with_bias and grp.db is None:
        grp.db = gb = torch.empty(g.K, dtype=torch.float32, device=x.device)
    grp.segs.append((x, gy, bool(relu_x), bool(with_bias)))
    return gw, gb


# ---- forward tape: one set of forward launches for several row ranges of a batch ---------------------------------
# The samples of a critic batch are independent (no normalisation across the batch), so passes over different inputs that
# use the same weights - the dropout passes on [real ; fake] and the gradient-penalty pass on x_hat - can share their
# forward launches: run the layer sequence ONCE on the concatenated rows under `tape_record()`, then build each pass's own
# autograd graph by running the same layer sequence under `tape_replay(tape, r0, r1)`: every conv / pool forward then
# returns rows [r0, r1) of the recorded result instead of launching.  The backward passes stay separate (they have
# different needs: weight gradients only vs. a differentiable data gradient).
_TAPE = {'mode': None, 'items': None, 'pos': 0, 'rows': None}


def _taped(launch):
    mode = _TAPE['mode']
    if mode is None:
        return launch()
    if mode == 'record':
        y = launch()
        _TAPE['items'].append(y)
        return y
    y = _TAPE['items'][_TAPE['pos']]
    _TAPE['pos'] += 1
    r0, r1 = _TAPE['rows']
    return y[r0:r1]


@contextlib.contextmanager
def tape_record():
    assert _TAPE['mode'] is None
    items = []
    _TAPE.update(mode='record', items=items, pos=0, rows=None)
    try:
        yield items
    finally:
        _TAPE.update(mode=None, items=None)


@contextlib.contextmanager
def tape_replay(items, r0, r1):
    assert _TAPE['mode'] is None
    _TAPE.update(mode='replay', items=items, pos=0, rows=(int(r0), int(r1)))
    try:
        yield
        assert _TAPE['pos'] == len(items), 'replayed layer sequence differs from the recorded one'
    finally:
        _TAPE.update(mode=None, items=None)


# --------------------------------------------------------------------------------- conv family
class ConvFn(Function):
    """y = conv(x, w) [+ b] [+ resid]; relu_in: y = conv(relu(x), w) ... without materialising relu(x)
    (SAME conv described by geometry g)"""

    @staticmethod
    def forward(ctx, x, w, b, resid, g, out_strides, relu_in=False, fork=False, epi=None):
        # epi (critic tail, dropout fused into the conv kernels): dict with
        #   out_drop / in_drop = (keep, seed, site, ctr) dropout specs, out_relu, out_drop_bwd_fused
        epi = epi or {}
        ctx.out_drop, ctx.in_drop = epi.get('out_drop'), epi.get('in_drop')
        ctx.out_relu = bool(epi.get('out_relu'))
        ctx.out_drop_bwd_fused = bool(epi.get('out_drop_bwd_fused'))
        ctx.mask_done = bool(epi.get('mask_done'))      # the consumer returns the gradient w.r.t. the conv result itself
        ctx.resid_up = bool(epi.get('resid_up')) and resid is not None      # resid is the low-resolution shortcut
        ctx.out_mask = epi.get('out_mask')         # constant tensor: y kept where out_mask > 0 (no resid / relu / dropout with it)
        ctx.g = g
        ctx.N = x.shape[0]
        ctx.x_strides = x.stride()
        ctx.want_w = _WEIGHT_GRADS
        ctx.has_b = b is not None
        ctx.has_resid = resid is not None
        ctx.relu_in = bool(relu_in)
        ctx.fork = bool(fork)
        if fork:
            ctx.set_materialize_grads(False)       # an unused shortcut branch must not cost a zero-filled add
        if ctx.out_mask is not None:
            y = K.conv_fwd(x, w, b, g, out_strides=out_strides, relu_in=relu_in, mask=ctx.out_mask)
        else:
            y = _taped(lambda: K.conv_fwd(x, w, b, g, resid=resid, relu=ctx.out_relu, out_strides=out_strides, relu_in=relu_in,
                                          drop=ctx.out_drop, resid_up=ctx.resid_up))
        if ctx.out_relu:
            ctx.save_for_backward(x, w, y)         # y > 0  <=>  pre-activation > 0 and the element survived the dropout
        else:
            ctx.save_for_backward(x, w)
        if fork:
            # second output = x itself (for the block's shortcut branch): x then has ONE consumer in the autograd
            # graph and the shortcut's gradient arrives here, where it rides the dgrad epilogue as `resid`
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, gy, g_fork=None):
        if ctx.out_relu:
            x, w, y = ctx.saved_tensors
        else:
            x, w = ctx.saved_tensors
        g = ctx.g
        if gy is None:                                   # fork only: y itself was not used
            return g_fork, None, None, None, None, None, None, None, None
        if ctx.out_mask is not None:
            gy = _relu_mask(gy, ctx.out_mask, 0.0)
        # gradient w.r.t. the conv result z, given the gradient w.r.t. y = dropout(relu(z))
        if ctx.mask_done:
            pass
        elif ctx.out_relu:
            gy = _relu_mask(gy, y, 0.0, (1.0 / ctx.out_drop[0]) if ctx.out_drop is not None else 1.0)
        elif ctx.out_drop is not None and not ctx.out_drop_bwd_fused:
            gy = DropoutRngFn.apply(gy, ctx.out_drop[0], ctx.out_drop[1], ctx.out_drop[2], ctx.out_drop[3], _cl_strides(gy.shape))
        gx = gw = gb = gr = None
        gr_alias = None
        mask = x.detach() if ctx.relu_in else None      # ReLU backward rides the dgrad epilogue (a constant: no graph edge)
        need_w = ctx.needs_input_grad[1] and ctx.want_w
        need_b = ctx.has_b and ctx.needs_input_grad[2] and ctx.want_w
        # (the weight gradient on a side stream next to the data gradient was measured: -7 %, cross-stream joins cost more than the idle
        # they fill - removed)
        if need_w and not torch.is_grad_enabled():
            gw, gb = _wgrad(x, gy, w, g, ctx.relu_in, need_b)          # first-order pass: may be queued (deferred_wgrads)
        elif need_w and need_b:
            gw, gb = ConvWgradBiasFn.apply(x, gy, g, ctx.relu_in)       # bias gradient rides the wgrad kernel
        elif need_w:
            gw = ConvWgradFn.apply(x, gy, g, ctx.relu_in)
        elif need_b:
            gb = ChannelSumFn.apply(gy)
        if ctx.needs_input_grad[0]:
            if g.x_up:
                gfull = ConvDgradFn.apply(gy, w, None, _no_up(g), ctx.N, None, None)
                gx = Pool2Fn.apply(gfull, 1.0)
                if mask is not None:
                    gx = _relu_mask(gx, x, 0.0)
            else:
                keep = ctx.x_strides if _is_plain_nchw(x) else None
                if mask is not None and not MASK_IN_DGRAD_EPILOGUE:
                    gx = _relu_mask(ConvDgradFn.apply(gy, w, None, g, ctx.N, keep, None), x, 0.0)
                    if g_fork is not None:
                        gx = add(gx, g_fork)
                    if ctx.in_drop is not None:
                        gx = DropoutRngFn.apply(gx, ctx.in_drop[0], ctx.in_drop[1], ctx.in_drop[2], ctx.in_drop[3], _cl_strides(gx.shape))
                elif ctx.has_resid and ctx.needs_input_grad[3] and torch.is_grad_enabled() and FORK_FUSION and not ctx.resid_up:
                    gx, gr_alias = ConvDgradFn.apply(gy, w, None, g, ctx.N, keep, mask, g_fork, ctx.in_drop, True)
                else:
                    gx = ConvDgradFn.apply(gy, w, None, g, ctx.N, keep, mask, g_fork, ctx.in_drop)
                g_fork = None
            if g_fork is not None:
                gx = add(gx, g_fork)
        elif g_fork is not None:
            gx = g_fork
        if ctx.has_resid and ctx.needs_input_grad[3]:
            gr = gr_alias if gr_alias is not None else (Pool2Fn.apply(gy, 1.0) if ctx.resid_up else gy)
        return gx, gw, gb, gr, None, None, None, None, None


class ConvDgradFn(Function):
    """gx = conv^T(gy, w) [+ b] [kept where mask > 0]   (with b: Deconv2D's forward; with mask: the data
    gradient of conv(relu(x)) w.r.t. x, mask = x)"""

    @staticmethod
    def forward(ctx, gy, w, b, g, N, out_strides, mask=None, resid=None, drop=None, fork=False):
        # fork: also return gy itself (the gradient that passes straight through the conv's residual input), so that gy has
        # ONE consumer in the first-backward graph and, in the double backward, the gradient arriving through that branch
        # is added in the epilogue of the conv this node launches (resid) instead of by an autograd `add`.
        ctx.fork = bool(fork)
        if fork:
            ctx.set_materialize_grads(False)
        ctx.g = g
        ctx.want_w = _WEIGHT_GRADS
        ctx.has_b = b is not None
        ctx.has_mask = mask is not None
        ctx.has_resid = resid is not None
        ctx.drop = drop                                # dropout mask the result is multiplied with (after mask and resid)
        if mask is not None:
            ctx.save_for_backward(gy, w, mask)
        else:
            ctx.save_for_backward(gy, w)
        ctx.pre = ctx.own = None
        if PREMASK_FUSION and not fork:                 # is gy a masked data gradient whose only processing there is the mask?
            ctx.pre = getattr(gy, '_ctgan_premask', None)
        dx = K.conv_dgrad(gy, w, g, N, out_strides=out_strides, bias=b, wt=_repacked(w, g, drop is None), mask=mask, resid=resid, drop=drop)
        if PREMASK_FUSION and mask is not None and resid is None and drop is None and b is None:
            tok = _PreMask()
            tok.mask, tok.fused_done = mask, False      # (the token does not reference dx: no cycle dx -> grad_fn -> ctx -> token -> dx)
            dx._ctgan_premask = ctx.own = tok
        return (dx, gy.view_as(gy)) if fork else dx

    @staticmethod
    def backward(ctx, ggx, gg_fork=None):
        if ggx is None:                                 # fork only: dx itself was not used
            return gg_fork, None, None, None, None, None, None, None, None, None
        need_res = ctx.has_resid and ctx.needs_input_grad[7]
        masked = False
        if (isinstance(ctx.drop, tuple) and ctx.has_mask and PREMASK_FUSION and not torch.is_grad_enabled()
                and tuple(ggx.stride()) == tuple(_cl_strides(ggx.shape)) and ctx.saved_tensors[2].stride() == ggx.stride()):
            # dropout and ReLU mask (both constants of the second pass) in one launch; the dropped-only tensor is the residual's gradient
            g_res, ggx = K.dropout_rng_mask(ggx, ctx.saved_tensors[2], ctx.drop[0], ctx.drop[1], ctx.drop[2], ctx.drop[3], want_dropped=need_res)
            masked = True
        else:
            if ctx.drop is not None:                    # the dropout mask is a constant of the second pass
                ggx = DropoutRngFn.apply(ggx, ctx.drop[0], ctx.drop[1], ctx.drop[2], ctx.drop[3], _cl_strides(ggx.shape))
            g_res = ggx if need_res else None           # added after the mask
        if masked:
            gy, w, mask = ctx.saved_tensors
        elif ctx.has_mask:
            gy, w, mask = ctx.saved_tensors
            fused = ctx.own is not None and getattr(ggx, '_ctgan_fused_for', None) is ctx.own and ctx.own.fused_done
            if ctx.own is not None:
                ctx.own.fused_done = False
            if not fused:
                ggx = _relu_mask(ggx, mask, 0.0)      # the mask is a constant of the second pass
            # (else: ggx IS the result of the consumer's conv, whose epilogue applied this node's mask)
        else:
            gy, w = ctx.saved_tensors
        g = ctx.g
        g_gy = g_w = g_b = None
        if ctx.needs_input_grad[0] and ctx.pre is not None and gg_fork is None:
            # gy = mask(a) * (...) was produced by a data-gradient node that masks what arrives for it: apply that mask here
            g_gy = ConvFn.apply(ggx, w, None, None, g, None, False, False, {'out_mask': ctx.pre.mask})
            g_gy._ctgan_fused_for = ctx.pre
            ctx.pre.fused_done = True
        elif ctx.needs_input_grad[0]:
            g_gy = ConvFn.apply(ggx, w, None, gg_fork, g, None, False)        # + the fork branch's gradient, in the epilogue
        elif gg_fork is not None:
            g_gy = gg_fork
        if ctx.needs_input_grad[1] and ctx.want_w:
            if torch.is_grad_enabled():
                g_w = ConvWgradFn.apply(ggx, gy, g, False)
            else:
                g_w, _ = _wgrad(ggx, gy, w, g, False, False)
        if ctx.has_b and ctx.needs_input_grad[2] and ctx.want_w:
            g_b = ChannelSumFn.apply(ggx)
        return g_gy, g_w, g_b, None, None, None, None, g_res, None, None


# ---- derived-filter cache ---------------------------------------------------------------------------------
# Registry parameters are used by several kernels per step in layouts other than HWIO: the rotated / phase-major
# filters of the data gradients (dropout passes, GP backward and its double backward) and the spread filters of the
# resampled convs.  Each derived layout lives in a persistent buffer that is rebuilt once per weight version
# (`lib.epoch()`): lazily on first use, or - `prepare_filters()` - all of them in one or two launches at the start
# of a step.  A consumer on another stream waits for the producer's event.
class _FilterEntry:
    __slots__ = ('src', 'buf', 'kind', 'pad', 'scale', 'epoch', 'ev', 'st', 'group', 'pre', 'pre_scale')

    def job(self):
        return (self.src, self.buf, self.kind, self.pad[0], self.pad[1], self.scale, self.pre, self.pre_scale)


_FCACHE = {}          # (src data_ptr, kind, R, S, C, K, pad_t, pad_l, scale) -> _FilterEntry
_SPREAD_BUFS = {}     # data_ptr of a cached spread buffer -> its entry (a spread filter's dgrad layouts are cached too)
_hooked = [False]


def clear_filter_cache():
    K._STABLE_PTRS.difference_update(_SPREAD_BUFS.keys())      # (a freed buffer's address may come back as an unrelated tensor)
    for ptr in _SPREAD_BUFS:
        K._STABLE_GROUP.pop(ptr, None)
    _FCACHE.clear()
    _SPREAD_BUFS.clear()


def _mark_built(entries):
    from . import tflib as lib
    st = ev = None
    if entries and entries[0].buf.is_cuda:
        st = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(st)
    for e in entries:
        e.epoch, e.ev, e.st = lib.epoch(e.group), ev, st


def _cached_filter(src, kind, pad=(0, 0), scale=1.0):
    """Derived layout `kind` of `src` (a registry Parameter, or a cached spread filter); None if not cacheable."""
    from . import tflib as lib
    parent = None
    if not isinstance(src, torch.nn.Parameter):
        parent = _SPREAD_BUFS.get(src.data_ptr())
        if parent is None or parent.epoch != lib.epoch(parent.group):
            return None
    key = (src.data_ptr(), kind) + tuple(src.shape) + tuple(pad) + (scale,)
    e = _FCACHE.get(key)
    if e is None:
        if not _hooked[0]:
            lib.on_delete_all_params(clear_filter_cache)
            _hooked[0] = True
        e = _FilterEntry()
        # a layout OF a cached spread filter is built straight from the parameter (job.pre): no job depends on another one
        e.src = parent.src if parent is not None else src
        e.pre, e.pre_scale = (parent.kind, parent.scale) if parent is not None else (0, 1.0)
        e.buf = torch.empty(K.filter_job_shape(kind, *src.shape), dtype=torch.float32, device=src.device)
        e.kind, e.pad, e.scale, e.epoch, e.ev, e.st = kind, tuple(pad), scale, -1, None, None
        e.group = parent.group if parent is not None else lib.group_of(src)     # network whose updates invalidate it
        _FCACHE[key] = e
        if kind in (K.FILTER_SPREAD, K.FILTER_SPREAD_FLIP):
            _SPREAD_BUFS[e.buf.data_ptr()] = e
            K._STABLE_PTRS.add(e.buf.data_ptr())       # the 16-bit family may cache its packed image per registry epoch
            K._STABLE_GROUP[e.buf.data_ptr()] = e.group
    if e.epoch != lib.epoch(e.group):
        K.filter_batch([e.job()])
        _mark_built([e])
    elif e.ev is not None and torch.cuda.current_stream() != e.st:
        torch.cuda.current_stream().wait_event(e.ev)
    return e.buf


def prepare_filters():
    """Rebuild every known derived filter for the current weight version now, on the current stream, in one launch (the
    data-gradient layouts of spread filters are computed from the parameter, not from the spread buffer)."""
    from . import tflib as lib
    todo = [e for e in _FCACHE.values() if e.epoch != lib.epoch(e.group)]
    if todo:
        K.filter_batch([e.job() for e in todo])
        _mark_built(todo)
    for key, e in _GEMM_FILTERS.items():      # before the packs: a packed image is built from the buffer's CURRENT contents
        _refresh_gemm_filter(key, e)
    if (todo or K._pack16) and (K.MMA_DTYPE is not None or K.X3_HYBRID):       # no 16-bit / split-mode launch can be routed otherwise
        K.prepare_packs()                 # the 16-bit / split-mode images of the parameters and of the filters just rebuilt: one launch


# A/B switch: the graphed critic step rebuilds its derived / packed filters on a side stream, under the step's first launches
PREP_ASYNC = _os.environ.get('CTGAN_PREP_ASYNC', '0') == '1'      # measured: 12.16 -> 12.40 ms per iteration (cross-queue edges in the replayed graph cost more than the 35 us they hide)
_PREP_SIDE = {}


def prepare_filters_async():
    """prepare_filters() on a SIDE stream forked from the current one; returns the event the caller must make its stream wait for before
    the step ends (every consumer of a derived filter / packed image waits for it by itself: _cached_filter, kernels._packed16).

    A critic step opens with ~35 us of filter work (the spread / rotated layouts and the split-mode fragment images of the weights its Adam
    step just changed: filter_batch + pack_batch, 256 workgroups each) while its first launches - input staging and the first conv, a
    few-channel kernel that reads the fp32 parameter itself - do not need any of it: the two run side by side.  The side stream starts
    behind everything already on the caller's stream (the previous step's readers of the old images are done)."""
    if not PREP_ASYNC or not torch.cuda.is_available() or _GEMM_FILTERS:      # (the GEMM-route buffers are refreshed by plain copies: no event of their own)
        prepare_filters()
        return None
    cur = torch.cuda.current_stream()
    dev = cur.device_index if hasattr(cur, 'device_index') else torch.cuda.current_device()
    side = _PREP_SIDE.get(dev)
    if side is None:
        side = _PREP_SIDE[dev] = torch.cuda.Stream()
    start = torch.cuda.Event()
    start.record(cur)
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        side.wait_event(start)
        prepare_filters()
        done.record(side)
    return done


def _repacked(w, g, plain=True):
    """The fp32 family's data-gradient layout of w (cached per weight version), or None where conv_dgrad does not read one.  plain: no
    epilogue dropout (the launches the 16-bit family takes in the bf16 / fp16 modes - from its own packed image)."""
    if not K.dgrad_wants_repack(g) or (plain and K.dgrad_runs_16bit(g)):
        return None
    kind = K.dgrad_filter_kind(g)
    return _cached_filter(w, kind, (g.pad_t, g.pad_l) if kind == K.FILTER_PHASES else (0, 0))


def prepare_dgrad_filters(params):
    """Kept for callers that fork side streams: make sure the derived filters exist on the current stream."""
    prepare_filters()


def _wgrad_backward(ctx, ggw, ggb):
    x, gy = ctx.saved_tensors
    g = ctx.g
    g_x = g_gy = None
    ggw = ggw.contiguous()
    mask = x.detach() if ctx.relu_x else None
    if ctx.needs_input_grad[0]:
        if g.x_up:
            g_x = Pool2Fn.apply(ConvDgradFn.apply(gy, ggw, None, _no_up(g), ctx.N, None, None), 1.0)
            if mask is not None:
                g_x = _relu_mask(g_x, x, 0.0)
        elif mask is not None and not MASK_IN_DGRAD_EPILOGUE:
            g_x = _relu_mask(ConvDgradFn.apply(gy, ggw, None, g, ctx.N, None, None), x, 0.0)
        else:
            g_x = ConvDgradFn.apply(gy, ggw, None, g, ctx.N, None, mask)
    if ctx.needs_input_grad[1]:
        g_gy = ConvFn.apply(x, ggw, ggb, None, g, None, ctx.relu_x)
    return g_x, g_gy


class ConvWgradFn(Function):
    """gw = sum_pixels x (x) gy     (relu_x: x -> relu(x) on the fly)"""

    @staticmethod
    def forward(ctx, x, gy, g, relu_x=False):
        ctx.g, ctx.N, ctx.relu_x = g, x.shape[0], bool(relu_x)
        ctx.save_for_backward(x, gy)
        return K.conv_wgrad(x, gy, g, relu_x=relu_x)

    @staticmethod
    def backward(ctx, ggw):
        g_x, g_gy = _wgrad_backward(ctx, ggw, None)
        return g_x, g_gy, None, None


class ConvWgradBiasFn(Function):
    """(gw, gb) = (sum_pixels x (x) gy, sum_pixels gy) in one launch."""

    @staticmethod
    def forward(ctx, x, gy, g, relu_x=False):
        ctx.g, ctx.N, ctx.relu_x = g, x.shape[0], bool(relu_x)
        ctx.save_for_backward(x, gy)
        return K.conv_wgrad(x, gy, g, with_bias=True, relu_x=relu_x)

    @staticmethod
    def backward(ctx, ggw, ggb):
        g_x, g_gy = _wgrad_backward(ctx, ggw, ggb)
        return g_x, g_gy, None, None


class ChannelSumFn(Function):
    """[N,K,P,Q] -> [K] (bias gradient); its adjoint broadcasts (only reached by 3rd-order use)."""

    @staticmethod
    def forward(ctx, gy):
        ctx.shape = gy.shape
        return K.colsum_channels(gy)

    @staticmethod
    def backward(ctx, g):
        return g.view(1, -1, 1, 1).expand(ctx.shape)


def _cl_strides(shape):
    """Strides of a dense channels-last tensor of logical shape [N,C,H,W] (what the conv kernels write)."""
    N, C, H, W = shape
    return (H * W * C, 1, W * C, C)


def _no_up(g):
    return ConvGeom(g.C, g.H, g.W, g.K, g.R, g.S, g.stride, False)


def _is_plain_nchw(x):
    return x.dim() == 4 and x.is_contiguous() and not x.permute(0, 2, 3, 1).is_contiguous()


class Im2colFn(Function):
    """Patch expansion of a few-channel image (linear; adjoint = Col2imFn)."""

    @staticmethod
    def forward(ctx, x, g, cpad):
        ctx.g, ctx.N = g, x.shape[0]
        ctx.x_strides = x.stride() if _is_plain_nchw(x) else None
        return K.im2col(x, g, cpad)

    @staticmethod
    def backward(ctx, gcols):
        return Col2imFn.apply(gcols, ctx.g, ctx.N, ctx.x_strides), None, None


class Col2imFn(Function):
    @staticmethod
    def forward(ctx, cols, g, N, out_strides):
        ctx.g, ctx.cpad = g, cols.shape[1]
        return K.col2im(K.to_channels_last(cols), g, N, out_strides)

    @staticmethod
    def backward(ctx, gx):
        return Im2colFn.apply(gx, ctx.g, ctx.cpad), None, None, None


# The GEMM filters of the few-channel routes - the first critic conv of the DCGAN scripts as im2col + 1x1 conv (TF/CT_gan_cifar.py:84), the
# image-producing Deconv2D as 1x1 conv + col2im (:76) - are reshapes of a registry parameter, zero-padded to a multiple of 32 rows /
# columns.  Built per CALL (reshape, cat with a fresh zero block, contiguous) they cost three launches per use plus a 16-bit pack per use
# (a temporary is not cacheable): 40-50 launches per iteration of config[1].  Kept in a buffer per (parameter, layout) instead, refreshed by
# one copy when the parameter's network was updated; the buffer's address is stable, so its packed image is cached per weight version and
# its weight gradients can be queued per filter.
_GEMM_FILTERS = {}        # (param data_ptr, kind, pad) -> [buffer, epoch, group, param]
_QUEUEABLE_GEMM = set()   # addresses of the 'cols' buffers: their weight gradient maps back to the parameter's by a VIEW, so it may be queued


def _clear_gemm_filters():
    K._STABLE_PTRS.difference_update(e[0].data_ptr() for e in _GEMM_FILTERS.values())
    for e in _GEMM_FILTERS.values():
        K._STABLE_GROUP.pop(e[0].data_ptr(), None)
    _QUEUEABLE_GEMM.clear()
    _GEMM_FILTERS.clear()


def _refresh_gemm_filter(key, e):
    from . import tflib as lib
    ver = lib.epoch(e[2])
    if e[1] == ver:
        return
    w, kind = e[3], key[1]
    rows = w.shape[0] * w.shape[1] * w.shape[2]
    src = w.detach().reshape(rows, w.shape[3])
    if kind == 'cols':
        e[0][:rows].copy_(src)
    else:
        e[0][:, :rows].copy_(src.t())
    e[1] = ver


class GemmFilterFn(Function):
    """kind 'cols': w [R,S,C,K] -> [R*S*C padded to `pad`, K] (rows of zeros appended); kind 'taps': w [R,S,Co,Ci] -> [Ci, R*S*Co padded to
    `pad`] (the transposed-conv filter as Ci -> (tap, channel) columns).  Linear; the adjoint slices / transposes the gradient back."""

    @staticmethod
    def forward(ctx, w, kind, pad):
        from . import tflib as lib
        ctx.kind, ctx.shape = kind, tuple(w.shape)
        # the weight gradient of this buffer is queued per filter (deferred_wgrads): the first use returns the result buffer - a VIEW of
        # it goes on to the parameter, filled when the queue is flushed - and later uses return None, which must stay None (a materialised
        # zero would be added to the still empty buffer at once)
        ctx.set_materialize_grads(False)
        rows = w.shape[0] * w.shape[1] * w.shape[2]
        key = (w.data_ptr(), kind, pad)
        e = _GEMM_FILTERS.get(key)
        if e is None:
            if not _GEMM_FILTERS:
                lib.on_delete_all_params(_clear_gemm_filters)
            shape = (pad, w.shape[3]) if kind == 'cols' else (w.shape[3], pad)
            e = [torch.zeros(shape, dtype=torch.float32, device=w.device), None, lib.group_of(w), w]
            _GEMM_FILTERS[key] = e
            K._STABLE_PTRS.add(e[0].data_ptr())
            K._STABLE_GROUP[e[0].data_ptr()] = e[2]
            if kind == 'cols':
                _QUEUEABLE_GEMM.add(e[0].data_ptr())
        _refresh_gemm_filter(key, e)
        return e[0].view(e[0].shape)          # a fresh tensor object per call (autograd attaches this node to it), same storage

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        R, S, C, Kk = ctx.shape
        rows = R * S * C
        if ctx.kind == 'cols':
            return g.reshape(-1, Kk)[:rows].reshape(ctx.shape), None, None
        return g.reshape(Kk, -1)[:, :rows].t().reshape(ctx.shape), None, None


def _gemm_filter(w, kind, pad):
    """The padded GEMM filter of a few-channel route: cached for registry parameters, built per call for anything else."""
    rows = w.shape[0] * w.shape[1] * w.shape[2]
    if isinstance(w, torch.nn.Parameter):
        return GemmFilterFn.apply(w, kind, pad)
    w2 = w.reshape(rows, w.shape[3])
    if kind == 'cols':
        return torch.cat([w2, w2.new_zeros(pad - rows, w.shape[3])], 0) if pad > rows else w2
    w2 = w2.t()
    return (torch.cat([w2, w2.new_zeros(w.shape[3], pad - rows)], 1) if pad > rows else w2).contiguous()


def conv2d(x, w, b=None, stride=1, resid=None, x_up=False, out_nchw=False, relu_in=False, pool=False, fork=False, epi=None):
    """TF-SAME conv on a logical NCHW tensor (any strides) with HWIO filter `w`.
    relu_in=True computes conv(relu(x)) without materialising relu(x); pool=True returns
    mean_pool2(conv(x) + b) [+ resid]; x_up=True convolves upsample2(x)."""
    R, S, C, Kout = w.shape
    N, Cx, H, W = x.shape
    assert Cx == C, 'channel mismatch: x has %d, filter expects %d' % (Cx, C)
    if fork:
        # returns (y, x'): x' is x for the caller's shortcut branch (see ConvFn.forward)
        if pool or x_up or out_nchw or C <= 4 or not FORK_FUSION:
            assert epi is None
            return conv2d(x, w, b, stride, resid, x_up, out_nchw, relu_in, pool), x
        g = ConvGeom(C, H, W, Kout, R, S, stride, False)
        return ConvFn.apply(x, w, b, resid, g, None, relu_in, True, epi)
    fusable = RESAMPLE_FUSION and stride == 1 and R % 2 == 1 and S % 2 == 1 and C % 32 == 0 and Kout % 32 == 0
    if pool:
        if fusable and not out_nchw and H % 2 == 0 and W % 2 == 0:
            return conv2d_mean_pool(x, w, b, resid, relu_in)
        out = mean_pool2(conv2d(x, w, b, stride, None, x_up, False, relu_in))
        return out if resid is None else add(out, resid)
    if x_up and fusable and R > 1 and resid is None and not relu_in and not out_nchw:
        return upsample_conv2d(x, w, b)
    if (x_up and FEWCH_DECONV and RESAMPLE_FUSION and Kout <= 4 and C % 32 == 0 and R % 2 == 1 and S % 2 == 1 and R > 1 and stride == 1
            and resid is None and not relu_in and not pool and not fork):
        y = upsample_conv2d(x, w, b)                              # image-producing UpsampleConv (LS/wgan_LSUN_Bedrooms128.py:160)
        return to_nchw(y) if out_nchw else y
    if C <= 4 and not x_up and stride == 1 and K.fewch_handles(ConvGeom(C, H, W, Kout, R, S, stride, False)):
        pass          # direct few-channel kernels (csrc/fewch.hip) behind the ordinary conv entry points
    elif C <= 4 and not x_up and Kout % 4 == 0:
        # few input channels: expand patches once, then the conv / wgrad / dgrad are 1x1 convs on the
        # pipelined MFMA kernels (csrc/skinny.hip)
        g = ConvGeom(C, H, W, Kout, R, S, stride, False)
        cpad = -(-(R * S * C) // 32) * 32
        cols = Im2colFn.apply(x, g, cpad)
        w2 = _gemm_filter(w, 'cols', cpad)
        if relu_in:
            x = relu(x)
            cols = Im2colFn.apply(x, g, cpad)
        return conv2d(cols, w2.view(1, 1, cpad, Kout), b, 1, resid, False, out_nchw)
    if x_up:
        H, W = 2 * H, 2 * W
    g = ConvGeom(C, H, W, Kout, R, S, stride, x_up)
    out_strides = None
    if out_nchw:
        out_strides = (Kout * g.P * g.Q, g.P * g.Q, g.Q, 1)
    return ConvFn.apply(x, w, b, resid, g, out_strides, relu_in, False, epi)


# A/B switch: transposed convs with <= 4 output channels as one 1x1 conv onto (tap, channel) columns + col2im
FEWCH_DECONV = _os.environ.get('CTGAN_FEWCH_DECONV', '1') != '0'


def conv2d_transpose(x, w_hwoi, b=None, stride=2):
    """tf.nn.conv2d_transpose 'SAME' with filter [k,k,out,in]: the dgrad of the strided conv whose
    HWIO filter is `w_hwoi` (I = out, O = in)."""
    R, S, Cout, Cin = w_hwoi.shape
    N, Cx, H, W = x.shape
    assert Cx == Cin
    g = ConvGeom(Cout, H * stride, W * stride, Cin, R, S, stride, False)
    assert (g.P, g.Q) == (H, W)
    if FEWCH_DECONV and Cout <= 4 and Cin % 32 == 0:
        # many -> few channels (the image-producing layer, Deconv2D(DIM, 3, 5) TF/CT_gan_cifar.py:76): as a data gradient the
        # GEMM would have 3 useful output columns of a 32-wide tile.  Transposed: every input pixel emits its contribution
        # to all R*S*Cout (tap, channel) outputs - ONE dense 1x1 conv Cin -> R*S*Cout on the pipelined kernels - and the
        # adjoint of im2col (col2im) sums the overlapping taps: no wasted MACs, no dilation zeros.
        rsc = R * S * Cout
        cpad = -(-rsc // 32) * 32
        w2 = _gemm_filter(w_hwoi, 'taps', cpad)
        cols = conv2d(x, w2.view(1, 1, Cin, cpad))
        y = Col2imFn.apply(cols, g, N, None)
        if b is not None:
            y = ChannelAffineFn.apply(y, torch.ones_like(b), b)
        return y
    return ConvDgradFn.apply(x, w_hwoi, b, g, N, None, None)


def linear(x, w, b=None):
    """x [n,in] @ w [in,out] + b, as a 1x1 conv on a 1x1 image."""
    n, cin = x.shape
    cout = w.shape[1]
    x4 = x.reshape(n, cin, 1, 1)
    y = conv2d(x4, w.view(1, 1, cin, cout), b)
    return y.reshape(n, cout)


# --------------------------------------------------------------------------------- resampled convs
# pool2(conv3x3(x, w)) and conv3x3(upsample2(x), w) are single stride-2 (transposed) convs with the 4x4 filter
# spread(w) = sum of the four one-tap shifts of w: 16 taps per low-resolution pixel instead of 4 x 9 (2.25x
# fewer multiplies, no full-resolution intermediate, no pool / upsample kernel).  The identity is exact in
# real arithmetic; in fp32 it changes the summation order like any other GEMM tiling does.
RESAMPLE_FUSION = _os.environ.get('CTGAN_RESAMPLE_FUSION', '1') != '0'
# residual blocks: the shortcut's gradient is added in the epilogue of the main branch's first data gradient
FORK_FUSION = _os.environ.get('CTGAN_FORK_FUSION', '1') != '0'


class FilterSpreadFn(Function):
    """[R,S,C,K] -> scale * spread (flip: rotated + I/O swapped); linear, adjoint = FilterFoldFn."""

    @staticmethod
    def forward(ctx, w, scale, flip):
        ctx.scale, ctx.flip = scale, flip
        ctx.set_materialize_grads(False)      # an absent gradient (its wgrad was merged into another use's) must stay absent
        if isinstance(w, torch.nn.Parameter):
            buf = _cached_filter(w, K.FILTER_SPREAD_FLIP if flip else K.FILTER_SPREAD, (0, 0), scale)
            return buf.detach()
        return K.filter_spread(w.contiguous(), scale, flip)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        if _DEFER['on'] and not torch.is_grad_enabled() and any(grp.dw.data_ptr() == g.data_ptr() for grp in _DEFER['groups'].values()):
            # g is a queued weight gradient that is only filled when the queue is flushed: fold after that.  (A FINISHED gradient - a use
            # launched at once, as the 16-bit modes and f32x3 do - is folded now: each use's node would otherwise return its own unfilled
            # buffer and the engine would sum them before any fold has run)
            g = g.contiguous()
            R, S = g.shape[0] - 1, g.shape[1] - 1
            C, Ko = (g.shape[3], g.shape[2]) if ctx.flip else (g.shape[2], g.shape[3])
            out = torch.empty((R, S, C, Ko), dtype=torch.float32, device=g.device)
            scale, flip = ctx.scale, ctx.flip
            _DEFER['post'].append(('fold', g, scale, flip, out))
            return out, None, None
        return FilterFoldFn.apply(g, ctx.scale, ctx.flip), None, None


class FilterFoldFn(Function):
    @staticmethod
    def forward(ctx, w4, scale, flip):
        ctx.scale, ctx.flip = scale, flip
        return K.filter_fold(w4.contiguous(), scale, flip)

    @staticmethod
    def backward(ctx, g):
        return FilterSpreadFn.apply(g, ctx.scale, ctx.flip), None, None


def conv2d_mean_pool(x, w, b=None, resid=None, relu_in=False):
    """mean_pool2(conv2d(x, w) + b) [+ resid] as ONE stride-2 conv (TF/CT_gan_cifar_resnet.py:89-92)."""
    R, S, C, Kout = w.shape
    N, Cx, H, W = x.shape
    assert Cx == C and H % 2 == 0 and W % 2 == 0 and R % 2 == 1 and S % 2 == 1
    w4 = FilterSpreadFn.apply(w, 0.25, False)
    g = ConvGeom(C, H, W, Kout, R + 1, S + 1, 2, False)
    assert (g.pad_t, g.pad_l) == ((R - 1) // 2, (S - 1) // 2)
    return ConvFn.apply(x, w4, b, resid, g, None, relu_in)


def upsample_conv2d(x, w, b=None):
    """conv2d(upsample2(x), w) + b as ONE stride-2 transposed conv (TF/CT_gan_cifar_resnet.py:100-107)."""
    R, S, C, Kout = w.shape
    N, Cx, H, W = x.shape
    assert Cx == C and R % 2 == 1 and S % 2 == 1
    w4 = FilterSpreadFn.apply(w, 1.0, True)                       # [R+1,S+1,Kout,C]: HWIO filter of the adjoint conv
    if Kout <= 4:
        return conv2d_transpose(x, w4, b, stride=2)               # many -> few channels: 1x1 conv onto columns + col2im
    g = ConvGeom(Kout, 2 * H, 2 * W, C, R + 1, S + 1, 2, False)    # the strided conv whose data gradient this is
    assert (g.P, g.Q) == (H, W) and (g.pad_t, g.pad_l) == ((R - 1) // 2, (S - 1) // 2)
    return ConvDgradFn.apply(x, w4, b, g, N, None, None)


# --------------------------------------------------------------------------------- mask ops
class LReluFn(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        y = K.lrelu_fwd(x, alpha)
        ctx.alpha = alpha
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return _relu_mask(gy, y, ctx.alpha), None


def _relu_mask(gy, ref, alpha, scale=1.0):
    """gy * (ref > 0 ? 1 : alpha) * scale.  `ref` enters DETACHED: the mask's derivative w.r.t. ref is zero almost everywhere,
    and a graph edge to the forward activation would make the engine walk the whole forward graph in the double backward
    (ten data gradients of exact zeros per gradient-penalty pass - what TF executes as written)."""
    return LReluBwdFn.apply(gy, ref.detach(), alpha, scale)


class LReluBwdFn(Function):
    @staticmethod
    def forward(ctx, gy, ref, alpha, scale=1.0):
        ctx.alpha, ctx.scale = alpha, scale
        ctx.save_for_backward(ref)
        return K.lrelu_bwd(gy, ref, alpha, scale)

    @staticmethod
    def backward(ctx, ggx):
        (ref,) = ctx.saved_tensors
        return _relu_mask(ggx, ref, ctx.alpha, ctx.scale), None, None, None


def relu(x):
    return LReluFn.apply(x, 0.0)


def leaky_relu(x, alpha=0.2):
    return LReluFn.apply(x, alpha)


class DropoutFn(Function):
    """tf.nn.dropout with the uniform draw `u` explicit; linear in x, self-adjoint."""

    @staticmethod
    def forward(ctx, x, u, keep):
        ctx.keep = keep
        u = K.match_layout(u, x)
        ctx.save_for_backward(u)
        return K.dropout(x, u, keep)

    @staticmethod
    def backward(ctx, gy):
        (u,) = ctx.saved_tensors
        return DropoutFn.apply(gy, u, ctx.keep), None, None


class DropoutRngFn(Function):
    """tf.nn.dropout whose mask is regenerated from the Philox stream (seed, site, device step counter) wherever it
    is needed (forward, backward, double backward): no uniform tensor, no mask tensor.  Linear in x, self-adjoint."""

    @staticmethod
    def forward(ctx, x, keep, seed, sid, ctr, strides, bwd_fused=False):
        if strides is not None and tuple(x.stride()) != tuple(strides):
            x = K.copy4d(x, torch.empty_strided(x.shape, strides, dtype=x.dtype, device=x.device))   # same physical order as the forward
        ctx.cfg = (keep, seed, sid, ctr, tuple(x.stride()))
        ctx.bwd_fused = bool(bwd_fused)       # the consumer's data gradient already carries this mask (conv dgrad epilogue)
        return _taped(lambda: K.dropout_rng(x, keep, seed, sid, ctr))

    @staticmethod
    def backward(ctx, gy):
        if ctx.bwd_fused:
            return gy, None, None, None, None, None, None
        keep, seed, sid, ctr, strides = ctx.cfg
        return DropoutRngFn.apply(gy, keep, seed, sid, ctr, strides), None, None, None, None, None, None


# A/B switch: LeakyReLU + dropout of the DCGAN critics as one launch (forward / backward / double backward each)
LRELU_DROP_FUSION = _os.environ.get('CTGAN_LRELU_DROP', '1') != '0'


class LReluDropFn(Function):
    """dropout(LeakyReLU(x)) - the activation pair after every conv of the DCGAN critics (TF/CT_gan_cifar.py:84-98) - in one launch, mask from
    the Philox stream; the backward (a diagonal scaling by slope * mask / keep, hence its own adjoint: the double backward is the same op)
    in one launch as well, with the forward RESULT as the sign reference."""

    @staticmethod
    def forward(ctx, x, alpha, keep, seed, sid, ctr):
        if not K.is_dense(x):
            x = x.contiguous()
        y = K.lrelu_dropout_rng(x, x, alpha, keep, seed, sid, ctr)
        ctx.cfg = (alpha, keep, seed, sid, ctr)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return (LReluDropBwdFn.apply(gy, y.detach(), *ctx.cfg),) + (None,) * 5


class LReluDropBwdFn(Function):
    @staticmethod
    def forward(ctx, gy, ref, alpha, keep, seed, sid, ctr):
        if tuple(gy.stride()) != tuple(ref.stride()):      # the draws are indexed by physical offset: same layout as the forward
            gy = K.copy4d(gy, torch.empty_strided(gy.shape, ref.stride(), dtype=gy.dtype, device=gy.device)) if gy.dim() == 4 else gy.contiguous()
        ctx.cfg = (alpha, keep, seed, sid, ctr)
        ctx.save_for_backward(ref)
        return K.lrelu_dropout_rng(gy, ref, alpha, keep, seed, sid, ctr)

    @staticmethod
    def backward(ctx, ggx):
        (ref,) = ctx.saved_tensors
        return (LReluDropBwdFn.apply(ggx, ref, *ctx.cfg),) + (None,) * 6


def lrelu_dropout(x, alpha, keep_prob, rng):
    """dropout(leaky_relu(x, alpha), keep_prob) with the mask drawn from `rng`'s next dropout stream: one launch each way."""
    if keep_prob == 1.0:
        return leaky_relu(x, alpha)
    spec = drop_spec(rng, keep_prob)
    return LReluDropFn.apply(x, alpha, spec[0], spec[1], spec[2], spec[3])


class RowsCatDropFn(Function):
    """dropout([x ; x[:n_extra]]) in one launch; the dropout's backward is applied by the consumer (conv dgrad epilogue,
    `bwd_fused`), so the backward here is the concat's adjoint alone - one launch instead of slice + zero-fill + add."""

    @staticmethod
    def forward(ctx, x, n_extra, keep, seed, sid, ctr):
        ctx.cfg = (x.shape[0], n_extra)
        return _taped(lambda: K.rows_cat_dropout(x, n_extra, keep, seed, sid, ctr))

    @staticmethod
    def backward(ctx, g):
        n, n_extra = ctx.cfg
        if not K.is_dense(g) or (g.dim() == 4 and not g.is_contiguous() and not g.permute(0, 2, 3, 1).is_contiguous()):
            g = g.contiguous()
        return K.rows_cat_bwd(g, n, n_extra), None, None, None, None, None


class RowsSelectFn(Function):
    """Concatenation of row ranges of x (ranges may repeat rows); adjoint = per-range accumulation into the rows."""

    @staticmethod
    def forward(ctx, x, ranges):
        ctx.ranges, ctx.n = ranges, x.shape[0]
        parts = [x[a:b] for a, b in ranges]
        y = torch.cat(parts, 0)
        if x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous() and not y.permute(0, 2, 3, 1).is_contiguous():
            y = K.to_channels_last(y)
        return y

    @staticmethod
    def backward(ctx, g):
        if g.dim() == 4 and not g.permute(0, 2, 3, 1).is_contiguous():
            g = K.to_channels_last(g)
        out = None
        r0 = 0
        pieces = []
        for a, b in ctx.ranges:
            pieces.append((a, b, g[r0:r0 + (b - a)]))
            r0 += b - a
        # rows covered once are copied, rows covered several times are summed (axpby kernel)
        acc = {}
        order = []
        for a, b, piece in pieces:
            if (a, b) in acc:
                acc[(a, b)] = add(acc[(a, b)], piece) if torch.is_grad_enabled() else K.axpby(acc[(a, b)], piece, 1.0, 1.0)
            else:
                acc[(a, b)] = piece
                order.append((a, b))
        order.sort()
        assert order[0][0] == 0 and order[-1][1] == ctx.n and all(order[i][1] == order[i + 1][0] for i in range(len(order) - 1)), \
            'row ranges must tile the input (possibly repeated)'
        gx = torch.cat([acc[k] for k in order], 0)
        if g.dim() == 4 and not gx.permute(0, 2, 3, 1).is_contiguous():
            gx = K.to_channels_last(gx)
        return gx, None


def rows_select(x, ranges):
    """cat([x[a:b] for (a, b) in ranges]); the ranges must tile x's rows, repeats allowed (a shared trunk feeding several passes)."""
    return RowsSelectFn.apply(x, tuple((int(a), int(b)) for a, b in ranges))


def rows_cat_dropout(x, n_extra, spec):
    """spec = drop_spec(...): (keep, seed, site, ctr).  The consumer must apply the mask in its backward (in_drop=spec)."""
    return RowsCatDropFn.apply(x, int(n_extra), spec[0], spec[1], spec[2], spec[3])


def drop_spec(rng, keep_prob):
    """(keep, seed, site, counter) of the next dropout call site of `rng`; None for keep_prob == 1."""
    if keep_prob == 1.0:
        return None
    return (float(keep_prob), rng.seed, rng._sid(), rng.ctr)


def dropout(x, keep_prob, u=None, rng=None, spec=None, bwd_fused=False):
    """u: explicit uniform draw (parity tests); else the mask comes from `rng` (DeviceRNG) inside the kernel.
    spec: a drop_spec drawn earlier; bwd_fused: the consumer's data gradient applies the mask (conv epilogue)."""
    if keep_prob == 1.0:
        return x
    if u is None:
        if not K.is_dense(x):
            x = x.contiguous()
        if spec is None:
            spec = drop_spec(rng, keep_prob)
        return DropoutRngFn.apply(x, spec[0], spec[1], spec[2], spec[3], None, bwd_fused)
    return DropoutFn.apply(x, u, float(keep_prob))


# --------------------------------------------------------------------------------- resampling
class Pool2Fn(Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return _taped(lambda: K.pool2(x, scale))

    @staticmethod
    def backward(ctx, gy):
        return Upsample2Fn.apply(gy, ctx.scale), None


class Upsample2Fn(Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return K.upsample2(x, scale)

    @staticmethod
    def backward(ctx, gy):
        return Pool2Fn.apply(gy, ctx.scale), None


def mean_pool2(x):
    return Pool2Fn.apply(x, 0.25)


def upsample2(x):
    return Upsample2Fn.apply(x, 1.0)


class SpatialMeanFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[2], x.shape[3])
        return K.spatial_sum(x, 1.0 / (x.shape[2] * x.shape[3]))

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.hw
        return SpatialBcastFn.apply(g, H, W, 1.0 / (H * W))


class SpatialBcastFn(Function):
    @staticmethod
    def forward(ctx, g, H, W, scale):
        ctx.scale = scale
        return K.spatial_bcast(g, H, W, scale)

    @staticmethod
    def backward(ctx, gy):
        return _SpatialSumScaled.apply(gy, ctx.scale), None, None, None


class _SpatialSumScaled(Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        ctx.hw = (x.shape[2], x.shape[3])
        return K.spatial_sum(x, scale)

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.hw
        return SpatialBcastFn.apply(g, H, W, ctx.scale), None


def spatial_mean(x):
    return SpatialMeanFn.apply(x)


class Copy4dFn(Function):
    """Layout change (NCHW <-> channels-last); values unchanged, adjoint = copy back."""

    @staticmethod
    def forward(ctx, x, to_cl):
        ctx.to_cl = to_cl
        return K.to_channels_last(x) if to_cl else K.to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return Copy4dFn.apply(g, not ctx.to_cl), None


def to_channels_last(x):
    return Copy4dFn.apply(x, True)


def to_nchw(x):
    return Copy4dFn.apply(x, False)


class CropFn(Function):
    """x[:, :, :h, :w] as a dense channels-last tensor (MNIST generator, TF/CT_gan_mnist.py:76)."""

    @staticmethod
    def forward(ctx, x, h, w):
        ctx.shape = x.shape
        ctx.hw = (h, w)
        v = x[:, :, :h, :w]
        return K.copy4d(v, K.empty_cl(*v.shape, device=x.device))

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.hw
        N, C, H, W = ctx.shape
        full = K.empty_cl(N, C, H, W, device=g.device)
        full.zero_()
        K.copy4d(g, full[:, :, :h, :w])
        return full, None, None


def crop(x, h, w):
    return CropFn.apply(x, h, w)


class AddFn(Function):
    """a*x + b*y through the axpby kernel (used where the add cannot ride a conv epilogue)."""

    @staticmethod
    def forward(ctx, x, y, a, b):
        ctx.a, ctx.b = a, b
        return K.axpby(x, y, a, b)

    @staticmethod
    def backward(ctx, g):
        ga = g if ctx.a == 1.0 else ScaleFn.apply(g, ctx.a)
        gb = g if ctx.b == 1.0 else ScaleFn.apply(g, ctx.b)
        return ga, gb, None, None


class ScaleFn(Function):
    @staticmethod
    def forward(ctx, x, a):
        ctx.a = a
        return K.axpby(x, None, a, 0.0)

    @staticmethod
    def backward(ctx, g):
        return ScaleFn.apply(g, ctx.a), None


def add(x, y):
    return AddFn.apply(x, y, 1.0, 1.0)


# --------------------------------------------------------------------------------- activations (G)
class TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = K.tanh_fwd(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return K.tanh_bwd(gy.contiguous() if not K.is_dense(gy) else gy, y)


class SigmoidFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = K.sigmoid_fwd(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return K.sigmoid_bwd(gy.contiguous() if not K.is_dense(gy) else gy, y)


def tanh(x):
    return TanhFn.apply(x)


def sigmoid(x):
    return SigmoidFn.apply(x)


# --------------------------------------------------------------------------------- batch norm (G)
class BatchNormFn(Function):
    """Training-mode BN (+ optional per-label scale/offset, + optional fused ReLU)."""

    @staticmethod
    def forward(ctx, x, scale, offset, labels, groups, relu):
        y, mean, rstd, x4 = K.bn_fwd(x, scale, offset, labels, groups, relu)
        ctx.groups, ctx.relu = groups, relu
        ctx.labels = labels
        ctx.in_shape = x.shape
        ctx.save_for_backward(x4, mean, rstd, scale, offset)
        return y

    @staticmethod
    def backward(ctx, gy):
        x4, mean, rstd, scale, offset = ctx.saved_tensors
        gx, gs, go = K.bn_bwd(gy, x4, mean, rstd, scale, offset, ctx.labels, ctx.groups, ctx.relu)
        if len(ctx.in_shape) == 2:
            gx = gx.reshape(ctx.in_shape)
        return gx, gs.view(scale.shape), go.view(offset.shape), None, None, None


def batch_norm(x, scale, offset, labels=None, groups=1, relu=False):
    return BatchNormFn.apply(x, scale, offset, labels, groups, relu)


# --------------------------------------------------------------------------------- layer norm (config[4] critic)
# TF/tflib/ops/layernorm.py:6-20 = tf.nn.moments over (C,H,W) per sample + tf.nn.batch_normalization with a per-channel
# scale / offset.  The critic is differentiated TWICE through it (gradient penalty), so the operator is a composition
# of five kernel-backed maps whose backwards are again members of the set (mul, rsqrt, per-sample sum / broadcast,
# per-channel affine + the existing axpby / channel sum): every derivative order is a composition of the same kernels.
class MulFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return K.mul(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (MulFn.apply(g, b) if ctx.needs_input_grad[0] else None), (MulFn.apply(g, a) if ctx.needs_input_grad[1] else None)


class RsqrtFn(Function):
    @staticmethod
    def forward(ctx, v, eps):
        r = K.rsqrt(v, eps)
        ctx.save_for_backward(r)
        return r

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors                      # d/dv (v+eps)^(-1/2) = -1/2 r^3
        return ScaleFn.apply(MulFn.apply(g, MulFn.apply(MulFn.apply(r, r), r)), -0.5), None


class SampleSumFn(Function):
    """[N, ...] -> [N], scale * sum over the non-batch axes; adjoint = SampleBcastFn."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        ctx.save_for_backward(x)                      # only its shape / layout are used
        return K.sample_sum(x, scale)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return SampleBcastFn.apply(g.contiguous(), x, ctx.scale), None


class SampleBcastFn(Function):
    @staticmethod
    def forward(ctx, v, like, scale):
        ctx.scale = scale
        return K.sample_bcast(v, like, scale)

    @staticmethod
    def backward(ctx, g):
        return SampleSumFn.apply(g if K.is_dense(g) else g.contiguous(), ctx.scale), None, None


class ChannelAffineFn(Function):
    """y = x * s[c] + o[c] on a channels-last [N,C,H,W] (or [N,C]) tensor."""

    @staticmethod
    def forward(ctx, x, s, o):
        ctx.has_o = o is not None
        ctx.save_for_backward(x, s)
        return K.channel_affine(x, s, o)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = K.match_layout(g, x) if g.dim() == 4 else g.contiguous()
        gx = ChannelAffineFn.apply(g, s, None) if ctx.needs_input_grad[0] else None
        gs = _channel_sum(MulFn.apply(g, x)) if ctx.needs_input_grad[1] else None
        go = _channel_sum(g) if (ctx.has_o and ctx.needs_input_grad[2]) else None
        return gx, gs, go


class _ChannelSum2dFn(Function):
    """[N,C] -> [C]."""

    @staticmethod
    def forward(ctx, g):
        ctx.n = g.shape[0]
        return K.colsum_channels(g.reshape(g.shape[0], g.shape[1], 1, 1))

    @staticmethod
    def backward(ctx, gg):
        return gg.reshape(1, -1).expand(ctx.n, -1).contiguous()


def _channel_sum(t):
    return ChannelSumFn.apply(t) if t.dim() == 4 else _ChannelSum2dFn.apply(t)


# A/B switch: fused Layernorm kernels (csrc/layernorm.hip: 2-3 launches per map) instead of the ~9-kernel composition
LN_FUSED = _os.environ.get('CTGAN_LN_FUSED', '1') != '0'


class LayerNormFn(Function):
    """y = [relu](Layernorm(x) * scale + offset) with fused forward / backward / double-backward kernels.  The backward is itself a
    Function (LayerNormBwdFn) whose backward is the bwd2 kernel: the gradient penalty differentiates the critic twice.  With
    relu the result doubles as the mask of the backward maps (y > 0): no separate ReLU / ReLU-backward passes."""

    @staticmethod
    def forward(ctx, x, scale, offset, eps, relu):
        y, mean, rstd = K.layernorm_fwd(x, scale, offset, eps, relu)
        ctx.relu = bool(relu)
        if relu:
            ctx.save_for_backward(x, scale, mean, rstd, y)
        else:
            ctx.save_for_backward(x, scale, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, mean, rstd = ctx.saved_tensors[:4]
        ymask = ctx.saved_tensors[4].detach() if ctx.relu else None       # a constant of every derivative order
        want_params = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if want_params:
            gx, gs, go = LayerNormBwdFn.apply(gy, x, scale, mean, rstd, ymask, True)
            return gx, gs, go, None, None
        return LayerNormBwdFn.apply(gy, x, scale, mean, rstd, ymask, False), None, None, None, None


class LayerNormBwdFn(Function):
    """(gy, x, scale) -> gx [, gscale, goffset]; mean / rstd are functions of x that the bwd2 formula differentiates through."""

    @staticmethod
    def forward(ctx, gy, x, scale, mean, rstd, ymask, want_params):
        gx, gs, go = K.layernorm_bwd(gy, x, scale, mean, rstd, want_params, ymask)
        ctx.has_mask = ymask is not None
        if ymask is not None:
            ctx.save_for_backward(gy, x, scale, mean, rstd, ymask)
        else:
            ctx.save_for_backward(gy, x, scale, mean, rstd)
        ctx.set_materialize_grads(False)
        if want_params:
            return gx, gs, go
        return gx

    @staticmethod
    def backward(ctx, u, u_s=None, u_o=None):
        if u_s is not None or u_o is not None:
            raise NotImplementedError('derivatives of the Layernorm parameter gradients (third order) are not used by any loss')
        if u is None:
            return None, None, None, None, None, None, None
        if torch.is_grad_enabled():
            raise NotImplementedError('third-order derivatives through Layernorm')
        gy, x, scale, mean, rstd = ctx.saved_tensors[:5]
        ymask = ctx.saved_tensors[5] if ctx.has_mask else None
        cg, cx, cs = K.layernorm_bwd2(u, gy, x, scale, mean, rstd, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2], ymask)
        return cg, cx, cs, None, None, None, None


def layer_norm(x, scale, offset, eps=1e-5, relu=False):
    """Per-sample normalisation over all non-batch axes, then per-channel scale / offset (TF/tflib/ops/layernorm.py);
    relu=True also applies the nonlinearity that follows it in the critics' blocks (fused kernels: same pass)."""
    if x.dim() == 4 and not x.permute(0, 2, 3, 1).is_contiguous():
        x = to_channels_last(x)
    elif x.dim() == 2:
        x = x.contiguous()
    if LN_FUSED and K.layernorm_supported(x):
        return LayerNormFn.apply(x, scale, offset, float(eps), bool(relu))
    y = layer_norm_composed(x, scale, offset, eps)
    return globals()['relu'](y) if relu else y


def layer_norm_composed(x, scale, offset, eps=1e-5):
    """The same operator from five kernel-backed maps that are closed under differentiation (any channel count)."""
    inv = 1.0 / x[0].numel()
    m = SampleSumFn.apply(x, inv)
    xc = AddFn.apply(x, SampleBcastFn.apply(m, x, 1.0), 1.0, -1.0)
    v = SampleSumFn.apply(MulFn.apply(xc, xc), inv)               # biased variance (tf.nn.moments)
    r = RsqrtFn.apply(v, float(eps))
    xh = MulFn.apply(xc, SampleBcastFn.apply(r, x, 1.0))
    return ChannelAffineFn.apply(xh, scale, offset)


# --------------------------------------------------------------------------------- loss heads
class GradPenaltyFn(Function):
    """lambda * mean((||g_b|| - 1)^2)"""

    @staticmethod
    def forward(ctx, g, lam, defer_mean=False):
        # defer_mean: gp is a slot that critic_tail_heads(slopes=...) fills (the batch mean rides that function's last kernel)
        gp, slopes = K.gp_fwd(g, lam, defer_mean)
        ctx.lam = lam
        ctx.save_for_backward(g, slopes)
        ctx.mark_non_differentiable(slopes)
        ctx.set_materialize_grads(False)          # no zero-filled gradient for the (unused) slopes output: one fill launch per step less
        return gp, slopes

    @staticmethod
    def backward(ctx, gout, _):
        g, slopes = ctx.saved_tensors
        return K.gp_bwd(g, slopes, gout, ctx.lam), None, None


class ConsistencyFn(Function):
    @staticmethod
    def forward(ctx, d, d_, f, f_, lam2, M):
        ct, ct_i = K.ct_fwd(d, d_, f, f_, lam2, M)
        ctx.lam2, ctx.M = lam2, M
        ctx.save_for_backward(d, d_, f, f_, ct_i)
        return ct

    @staticmethod
    def backward(ctx, gout):
        d, d_, f, f_, ct_i = ctx.saved_tensors
        gd, gd_, gf, gf_ = K.ct_bwd(d, d_, f, f_, ct_i, gout, ctx.lam2, ctx.M)
        return gd, gd_, gf, gf_, None, None


class SoftmaxCEFn(Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss, probs, ncorrect = K.softmax_ce_fwd(logits, labels)
        ctx.save_for_backward(probs, labels)
        ctx.mark_non_differentiable(ncorrect)
        return loss, ncorrect

    @staticmethod
    def backward(ctx, gout, _):
        probs, labels = ctx.saved_tensors
        return K.softmax_ce_bwd(probs, labels, gout), None


class MeanDiffFn(Function):
    """sa*mean(x[:na]) + sb*mean(x[na:])"""

    @staticmethod
    def forward(ctx, x, na, nb, sa, sb):
        ctx.cfg = (na, nb, sa, sb)
        return K.mean_diff_fwd(x, na, nb, sa, sb)

    @staticmethod
    def backward(ctx, gout):
        return K.mean_diff_bwd(gout, *ctx.cfg), None, None, None, None


class CriticHeadsFn(Function):
    """All loss heads of the batched dropout passes in one kernel: (cost_without_gp, wgan, ct, acgan)."""

    @staticmethod
    def forward(ctx, d, f, a, labels, B, lam2, M, scale, gp=None):
        d, f = d.contiguous(), f.contiguous()
        a = a.contiguous() if a is not None else None
        out, ct_i, probs = K.critic_heads_fwd(d, f, a, labels, B, lam2, M, scale, gp.reshape(1) if gp is not None else None)
        ctx.has_gp = gp is not None
        ctx.cfg = (B, lam2, M, scale)
        ctx.labels = labels
        ctx.has_a = a is not None
        ctx.set_materialize_grads(False)
        if a is not None:
            ctx.save_for_backward(d, f, ct_i, probs)
        else:
            ctx.save_for_backward(d, f, ct_i)
        ctx.mark_non_differentiable(out[4])
        return out[0], out[1], out[2], out[3], out[4]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, _g4=None):
        B, lam2, M, scale = ctx.cfg
        if ctx.has_a:
            d, f, ct_i, probs = ctx.saved_tensors
        else:
            (d, f, ct_i), probs = ctx.saved_tensors, None
        if g1 is None and g2 is None and g3 is None:         # the usual case: only the summed cost is differentiated
            gout = g0.reshape(1).contiguous()
        else:
            z = lambda g: g.reshape(1) if g is not None else d.new_zeros(1)
            gout = torch.cat([z(g0), z(g1), z(g2), z(g3)])
        gd, gf, ga = K.critic_heads_bwd(d, f, probs, ctx.labels, ct_i, gout, B, lam2, M, scale)
        return gd, gf, ga, None, None, None, None, None, (g0.reshape(()) if (ctx.has_gp and g0 is not None) else None)


class CriticTailHeadsFn(Function):
    """reduce_mean + both Linear heads + all loss heads of the batched dropout passes, forward in two launches and backward
    in one (TF/CT_gan_cifar_resnet.py:179-186,244-248,288-291).  y = the last block's relu(dropout(.)) output [3B,nf,H,W]
    (dense channels-last), produced by a conv with epi['mask_done']: the gradient returned for y is already the gradient
    w.r.t. the conv result (mask and 1/keep applied here).  First-order only."""

    @staticmethod
    def forward(ctx, y, w_out, b_out, w_ac, b_ac, labels, B, lam2, M, scale, mask_scale, gp=None, slopes=None, gp_lambda=0.0, y_clean=None,
                clean_relu=False):
        # slopes: gp is the unwritten slot of gradient_penalty(defer_mean=True), filled here; y_clean: rows of the dropout-free pass,
        # their accuracies come back as the last output (both ride the two launches of the heads)
        out, f, d, a, ct_i, probs, acc = K.tail_critic_heads_fwd(y, B, w_out, b_out, w_ac, b_ac, labels, gp.reshape(1) if gp is not None else None,
                                                                 lam2, M, scale, slopes=slopes, gp_lambda=gp_lambda, y_clean=y_clean,
                                                                 clean_relu=clean_relu)
        ctx.has_gp = gp is not None
        ctx.cfg = (B, lam2, M, scale, mask_scale)
        ctx.labels = labels
        ctx.has_a = a is not None
        ctx.set_materialize_grads(False)
        if a is not None:
            ctx.save_for_backward(y, w_out, w_ac, d, f, ct_i, probs)
        else:
            ctx.save_for_backward(y, w_out, d, f, ct_i)
        if acc is None:
            acc = out.new_zeros(0)
        ctx.mark_non_differentiable(out[4], d, acc)
        return out[0], out[1], out[2], out[3], out[4], d, acc

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g0, g1, g2, g3, _g4=None, _gd=None, _gacc=None):
        B, lam2, M, scale, mask_scale = ctx.cfg
        if ctx.has_a:
            y, w_out, w_ac, d, f, ct_i, probs = ctx.saved_tensors
        else:
            (y, w_out, d, f, ct_i), w_ac, probs = ctx.saved_tensors, None, None
        if g1 is None and g2 is None and g3 is None:
            gout = g0.reshape(1).contiguous()
        else:
            z = lambda g: g.reshape(1) if g is not None else d.new_zeros(1)
            gout = torch.cat([z(g0), z(g1), z(g2), z(g3)])
        gy, gw_out, gb_out, gw_ac, gb_ac = K.tail_heads_bwd(y, d, f, probs, ctx.labels, ct_i, gout, B, lam2, M, scale, mask_scale, w_out, w_ac)
        return (gy, gw_out, gb_out, gw_ac, gb_ac, None, None, None, None, None, None,
                (g0.reshape(()) if (ctx.has_gp and g0 is not None) else None), None, None, None, None)


def critic_tail_heads(y, w_out, b_out, w_ac, b_ac, labels, B, lam2=2.0, M=0.0, acgan_scale=1.0, mask_scale=1.0, gp=None, slopes=None,
                      gp_lambda=0.0, y_clean=None, clean_relu=False):
    """-> (cost, wgan, ct, acgan, wgan + ct + gp, d [3B], acc [2] (empty without y_clean)); see CriticTailHeadsFn."""
    return CriticTailHeadsFn.apply(y, w_out, b_out, w_ac, b_ac, labels, int(B), float(lam2), float(M), float(acgan_scale),
                                   float(mask_scale), gp, slopes, float(gp_lambda), y_clean, bool(clean_relu))


class GenTailHeadsFn(Function):
    """Generator-step loss from the last critic block's relu(dropout(.)) output y (produced with epi['mask_done']): mean + both
    Linear heads + (-mean D + scale * CE) in two launches, the gradient w.r.t. the conv result in one (:321-330).  The critic's
    weights get no gradient in this step.  First-order only."""

    @staticmethod
    def forward(ctx, y, w_out, b_out, w_ac, b_ac, labels, ac_scale, mask_scale):
        out, probs, d = K.gen_heads_fwd(y, w_out, b_out, w_ac, b_ac, labels, ac_scale)
        ctx.cfg = (ac_scale, mask_scale)
        ctx.labels = labels
        ctx.has_a = w_ac is not None
        if ctx.has_a:
            ctx.save_for_backward(y, w_out, w_ac, probs)
        else:
            ctx.save_for_backward(y, w_out)
        ctx.mark_non_differentiable(d)
        ctx.set_materialize_grads(False)          # (no zero-filled gradient for the unused d output)
        return out.reshape(()), d

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g0, _gd=None):
        ac_scale, mask_scale = ctx.cfg
        if ctx.has_a:
            y, w_out, w_ac, probs = ctx.saved_tensors
        else:
            (y, w_out), w_ac, probs = ctx.saved_tensors, None, None
        gy = K.gen_heads_bwd(y, probs, ctx.labels, g0.reshape(1).contiguous(), ac_scale, mask_scale, w_out, w_ac)
        return gy, None, None, None, None, None, None, None


def gen_tail_heads(y, w_out, b_out, w_ac, b_ac, labels, ac_scale, mask_scale):
    """-> (cost, d [n])"""
    return GenTailHeadsFn.apply(y, w_out.detach(), b_out.detach() if b_out is not None else None,
                                w_ac.detach() if w_ac is not None else None, b_ac.detach() if b_ac is not None else None, labels,
                                float(ac_scale), float(mask_scale))


class GpHeadGradFn(Function):
    """dD/dz at the last block of the critic for the gradient-penalty branch (:284): D = mean_hw(relu(dropout(z))) . w_out, so
    gz = (y > 0) * w_out / hw / keep with y = relu(dropout(z)) - one launch instead of head forward + ones + Linear data
    gradient + broadcast + mask; its adjoint w.r.t. w_out is the only thing the double backward needs from it."""

    @staticmethod
    def forward(ctx, y, w_out, mask_scale):
        ctx.mask_scale = mask_scale
        ctx.save_for_backward(y, w_out)
        return K.gp_head_grad(y, w_out, mask_scale)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        y, w_out = ctx.saved_tensors
        if not gg.permute(0, 2, 3, 1).is_contiguous():
            gg = K.to_channels_last(gg)
        return None, K.gp_head_wgrad(gg, y, ctx.mask_scale, w_out), None


def gp_head_grad(y, w_out, mask_scale):
    return GpHeadGradFn.apply(y.detach(), w_out, float(mask_scale))


def critic_heads(d_all, f_all, a_all, labels, B, lam2=2.0, M=0.0, acgan_scale=1.0, gp=None):
    """(wgan + ct + gp + acgan_scale*acgan, wgan, ct, acgan, wgan + ct + gp) of the batched dropout passes (rows: real pass 1,
    fake pass 1, real pass 2); gp = the step's gradient-penalty scalar (differentiable input)."""
    return CriticHeadsFn.apply(d_all, f_all, a_all, labels, int(B), float(lam2), float(M), float(acgan_scale), gp)


def gradient_penalty(g, lam, defer_mean=False):
    """-> (gp, slopes).  defer_mean: gp holds no value until critic_tail_heads(..., gp, slopes=slopes, gp_lambda=lam) has run."""
    return GradPenaltyFn.apply(g, float(lam), bool(defer_mean))


def consistency_term(d, d_, f, f_, lam2=2.0, M=0.0):
    return ConsistencyFn.apply(d, d_, f, f_, float(lam2), float(M))


def softmax_cross_entropy(logits, labels):
    """mean sparse softmax CE and the number of argmax hits."""
    return SoftmaxCEFn.apply(logits, labels)


def mean_diff(x, na, nb, sa, sb):
    return MeanDiffFn.apply(x, int(na), int(nb), float(sa), float(sb))
