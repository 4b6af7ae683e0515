"""Tensor-level wrappers over the C-ABI (one function per entry point family).

PyTorch is used for device memory and the current HIP stream only; all arithmetic happens in
libctgan_hip.so.  Activations are logical NCHW `torch.Tensor`s like the reference's tensors; the
preferred physical layout is channels-last (NHWC), but every conv entry point takes explicit
strides, so NCHW images / sample tensors are consumed and produced in place.

No CPU fallback: every wrapper raises on a non-HIP tensor.
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import ConvDesc, I32x4, I64x4, check, lib

_ws_cache = {}

# Optional per-launch timing of the conv family (bench.py's roofline leg): when a list is
# installed here every conv launch is bracketed by HIP events on the launch stream and
# (kernel variant, algorithmic flops, start event, end event) is appended.
PROFILE = None


def _conv_flops(g, N):
    return 2.0 * N * g.P * g.Q * g.K * g.R * g.S * g.C


PROFILE_REPS = 1          # bench.py's roofline pass repeats each (idempotent) conv launch inside its event bracket
PROFILE_CLOCKS = []       # (device symbol, flops, ClockProbe) of the brackets that carry a shader-clock probe (the grouped weight gradient)
WGRAD_GROUP_EXTRA = 0     # bench.py: extra launches of the filter-column weight-gradient kernel per grouped call (in-situ timing by difference)


def _timed(g, N, launch):
    """Run `launch` (one C-ABI conv call).  With PROFILE set, bracket it with HIP events on the launch
    stream; PROFILE_REPS > 1 repeats the launch inside the bracket so that the event overhead
    (~10 us on this stack) is amortised and the per-launch time agrees with rocprofv3's."""
    if PROFILE is None:
        launch()
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    with ClockProbe() as probe:           # (the shader clock over the bracket: a one-wave kernel on a side stream)
        e0.record(st)
        for _ in range(PROFILE_REPS):
            launch()
        e1.record(st)
    PROFILE.append((last_kernel(), _conv_flops(g, N), e0, e1, PROFILE_REPS, (N, g.C, g.H, g.W, g.K, g.R, g.stride, int(g.x_up)), last_symbol()))
    PROFILE_CLOCKS.append((last_symbol(), _conv_flops(g, N), probe))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---- 16-bit matrix-core mode (csrc/igemm16.hip) ---------------------------------------------------------------------------
# None: every conv runs on the exact-fp32 MFMA family.  'bf16' / 'f16': the layers the 16-bit family takes (channel counts that
# are multiples of 32 / 64, even extents for stride-2 data gradients) multiply in bf16 / fp16 with fp32 accumulation; tensors
# stay fp32 in HBM, filters are packed to 16 bits once per weight version.  Few-channel layers and everything else stay fp32.
MMA_DTYPE = os.environ.get('CTGAN_MMA') or None      # experiments: start in a 16-bit mode without touching the caller
_MMA_CODE = {'bf16': 1, 'f16': 2, 'f32x3': 3}
# fp32 mode (MMA_DTYPE None): the layers on which the split mode 'f32x3' (fp32 operands as three bf16 terms, six bf16 MFMAs per product -
# fp32 accuracy, tests/test_gpu_kernels16.py) is the faster fp32 path - stride-1 convs / data gradients on whole-row or whole-image
# 128x128 tiles that fill the chip (ctgan_conv2d16_x3_prefers; 170-200 vs 110-125 TFLOP/s) - run on it; every other layer stays on the
# fp32 MFMA family.  Headline: 17.74 vs 19.04 ms per iteration.  CTGAN_X3_HYBRID=0: fp32 MFMA only (bench.py reports both).
X3_HYBRID = os.environ.get('CTGAN_X3_HYBRID', '1') != '0'


def _conv_mode(d, op, plain):
    """The 16-bit-family mode this launch runs in (None: fp32 MFMA family); `plain` = no epilogue the 16-bit family lacks."""
    if not plain:
        return None
    if MMA_DTYPE is not None:
        return MMA_DTYPE if lib.ctgan_conv2d16_supported(ctypes.byref(d), op, _MMA_CODE[MMA_DTYPE]) else None
    if X3_HYBRID and lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), op):
        return 'f32x3'
    return None


# Hybrid fp32 mode: the queued weight gradients of a step ride ONE grouped split-mode launch (ctgan_conv2d16_wgrad_group) where they
# qualify, instead of the fp32 family's grouped launch.  2 (default): also the large ones that mode 1 launches at once (wgrad_prefers_x3).
# Measured on one box, ms per iteration: 0 (fp32 group) 16.26, 1 15.56, 2 15.30.
X3_WGRAD_GROUP = int(os.environ.get('CTGAN_X3_WGRAD_GROUP', '2'))


def grouped16_mode():
    """The 16-bit family mode whose grouped weight-gradient launch (ctgan_conv2d16_wgrad_group) takes the queued weight gradients of a
    step: 'f32x3' in the hybrid fp32 mode and in the split mode, the mixed-precision mode itself under set_mma_dtype('bf16' / 'f16')."""
    if not X3_WGRAD_GROUP:
        return None
    if MMA_DTYPE is not None:
        return MMA_DTYPE
    return 'f32x3' if X3_HYBRID else None


def grouped16_takes(g, rows, xs=None):
    """Mixed-precision modes: is this weight gradient (geometry g over `rows` samples, x with strides xs - dense channels-last when None)
    queued for the grouped 16-bit launch?  Every problem the filter-column kernel takes (the library is asked: ctgan_conv2d16_wgrad_col_takes -
    128-multiples of channels, rows of 8-64 pixels, stride 1 / 2, images dense in memory) - one grouped launch per step balances them over the
    CUs - and otherwise the SMALL problems only (<= 16 K pixels, <= 512 channels): large ones outside the column kernel run faster on the
    wide tiles of their own launch."""
    if not (X3_WGRAD_GROUP and MMA_DTYPE in ('bf16', 'f16') and g.C % 128 == 0 and g.K % 128 == 0 and g.Q % 4 == 0 and not g.x_up
            and not fewch_handles(g)):
        return False
    if g.Q in (8, 16, 32, 64) and g.stride in (1, 2) and g.H == g.stride * g.P and g.W == g.stride * g.Q and g.P * g.Q >= 64 and not (g.P * g.Q & (g.P * g.Q - 1)):
        d = g.desc(rows, xs if xs is not None else (g.C * g.H * g.W, 1, g.W * g.C, g.C), (g.K * g.P * g.Q, 1, g.Q * g.K, g.K))
        if lib.ctgan_conv2d16_wgrad_col_takes(ctypes.byref(d), _MMA_CODE[MMA_DTYPE], rows):
            return True
    return g.C <= 512 and g.K <= 512 and rows * g.P * g.Q <= 16384


def grouped16_member(g):
    """Can the grouped 16-bit launch take this geometry at all (whatever its size)?  The capability behind grouped16_takes' size rule."""
    return (X3_WGRAD_GROUP and MMA_DTYPE in ('bf16', 'f16') and g.C % 128 == 0 and g.K % 128 == 0 and g.Q % 4 == 0 and not g.x_up
            and not fewch_handles(g))


def wgrad_prefers_x3(g, N, device=None):
    """True when the fp32 mode routes this weight gradient to the split mode (then it is launched at once, not queued for the fp32
    family's grouped launch).  Needs dense channels-last operands, which the callers of the large layers provide."""
    if MMA_DTYPE is not None or not X3_HYBRID or g.x_up or fewch_handles(g) or (device is not None and torch.device(device).type != 'cuda'):
        return False
    if X3_WGRAD_GROUP >= 2:
        return False              # everything is queued: the grouped split-mode launch takes the large problems as well
    d = g.desc(N, (g.C * g.H * g.W, 1, g.W * g.C, g.C), (g.K * g.P * g.Q, 1, g.Q * g.K, g.K))
    return bool(lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), 2))


_X3_LOG = os.environ.get('CTGAN_X3_LOG') == '1'      # diagnosis: which large stride-1 launches the hybrid routing leaves on the fp32 family, and why


def _x3_log(g, N, d, op, **why):
    if _X3_LOG and g.stride == 1 and g.R == 3 and N * g.P * g.Q >= 24576 and g.C % 32 == 0:
        print('x3-log op%d N%d C%d %dx%d K%d prefers=%d %s xs=%s ys=%s' % (op, N, g.C, g.H, g.W, g.K, lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), op),
                                                                 why, tuple(d.xs), tuple(d.ys)), flush=True)
_STABLE_PTRS = set()      # data_ptr of derived fp32 filters (spread filters) whose contents only change with the registry epoch
_STABLE_GROUP = {}        # ... -> the network ('Discriminator', 'Generator') whose updates change them (None: any update)
_pack16 = {}              # (data_ptr, op, dtype, geometry) -> [packed int16 buffer, registry epoch it was built for]


def set_mma_dtype(name):
    """name: None | 'bf16' | 'f16' | 'f32x3'.  Returns the previous setting."""
    global MMA_DTYPE
    if name is not None and name not in _MMA_CODE:
        raise ValueError("mma dtype must be None, 'bf16', 'f16' or 'f32x3'")
    old, MMA_DTYPE = MMA_DTYPE, name
    return old


class mma_dtype:
    """with kernels.mma_dtype('bf16'): ..."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.old = set_mma_dtype(self.name)

    def __exit__(self, *a):
        set_mma_dtype(self.old)
        return False


def clear_pack16_cache():
    _pack16.clear()


def _packed16(w, d, op, g, mode):
    """The 16-bit packed image of filter `w` for op (0 fwd, 1 dgrad) in `mode` - cached per registry epoch for parameters and for
    derived filters registered in _STABLE_PTRS (and then refreshed for every known image in ONE launch by prepare_packs, which the
    step calls right after it rebuilt the derived filters); packed per call for any other tensor."""
    from . import tflib
    n = lib.ctgan_conv2d16_filter_elems(ctypes.byref(d), op, _MMA_CODE[mode])
    stable = isinstance(w, torch.nn.Parameter) or w.data_ptr() in _STABLE_PTRS
    key = (w.data_ptr(), op, mode, g.C, g.H, g.W, g.K, g.R, g.S, g.stride)
    ent = _pack16.get(key) if stable else None
    if ent is not None:
        group = ent[6]
    else:
        group = (tflib.group_of(w) if isinstance(w, torch.nn.Parameter) else _STABLE_GROUP.get(w.data_ptr())) if stable else None
    ver = tflib.epoch(group)              # only updates of the filter's own network make its image stale
    if ent is not None and ent[1] == ver:
        if len(ent) > 7 and ent[7] is not None and w.is_cuda and torch.cuda.current_stream() != ent[7][1]:
            torch.cuda.current_stream().wait_event(ent[7][0])       # refreshed on another stream (prepare_packs under functional.prepare_filters_async)
        return ent[0]
    if ent is None:
        if not _pack16:
            tflib.on_delete_all_params(clear_pack16_cache)
        # (a strong reference to the filter's storage: the tensor OBJECT a conv wrapper receives is often a temporary - an autograd-saved
        # view - and while the entry lives the address cannot be reused by an unrelated tensor)
        ent = [torch.empty(n, dtype=torch.int16, device=w.device), None, w.detach(), g, op, mode, group]
        if stable:
            _pack16[key] = ent
    if _X3_LOG:
        print('x3-log pack op%d %s stable=%d cached=%d type=%s shape=%s' % (op, mode, stable, key in _pack16, type(w).__name__, tuple(w.shape)), flush=True)
    check(lib.ctgan_conv2d16_pack_filter(ctypes.byref(d), op, _MMA_CODE[mode], _ptr(w), _ptr(ent[0]), _stream()), 'conv2d16_pack_filter')
    ent[1] = ver
    if len(ent) > 7:
        ent[7] = None                     # packed on the consumer's own stream just now
    return ent[0]


PACK_BATCH_MAX_ELEMS = 1 << 21      # 16-bit elements per image (all planes; the folded 4x4x128x128 filters with their fragment image: 1.57 M)


def prepare_packs():
    """Refresh every cached packed image whose weight version is stale - one launch per mode (ctgan_conv2d16_pack_batch) instead of
    one per filter and operator at first use.  The caller has already rebuilt the derived (spread) filters on this stream."""
    from . import tflib
    if not _pack16:
        return
    by_mode = {}
    for key, ent in list(_pack16.items()):
        # small images only: a large one (DCGAN 5x5x256x512, config[4] 3x3x1024x1024) costs more to pack than a launch, and a step that
        # does not use it (the generator's data-gradient images during the critic steps) should not pay for it - those stay lazy
        if ent[1] == tflib.epoch(ent[6]) or ent[0].numel() > PACK_BATCH_MAX_ELEMS:
            continue
        by_mode.setdefault(ent[5], []).append((ent, ent[2]))
    for mode, items in by_mode.items():
        n = len(items)
        descs = (ConvDesc * n)()
        ops = (ctypes.c_int32 * n)()
        ws = (ctypes.c_void_p * n)()
        wps = (ctypes.c_void_p * n)()
        for i, (ent, w) in enumerate(items):
            descs[i] = ent[3].desc(1, (0, 0, 0, 0), (0, 0, 0, 0))
            # the packed layouts depend on channel counts, taps, stride and pads only; the stride checks of the entry points do not apply
            descs[i].xs = I64x4(4, 1, 4, 4); descs[i].ys = I64x4(4, 1, 4, 4)
            ops[i] = ent[4]
            ws[i] = w.data_ptr()
            wps[i] = ent[0].data_ptr()
        check(lib.ctgan_conv2d16_pack_batch(descs, ops, n, _MMA_CODE[mode], ws, wps, _stream()), 'conv2d16_pack_batch')
        mark = None
        if items[0][0][0].is_cuda:                # (stream, event) of this refresh: a consumer on another stream waits for it (_packed16)
            st = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(st)
            mark = (ev, st)
        for ent, _ in items:
            ent[1] = tflib.epoch(ent[6])
            if len(ent) > 7:
                ent[7] = mark
            else:
                ent.append(mark)


def _need_dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('ctgan_amd kernels need HIP device tensors (got %s); there is no CPU fallback'
                               % t.device)
        if t.dtype not in (torch.float32, torch.int32, torch.int16):
            raise TypeError('ctgan_amd kernels are fp32/int32 (got %s)' % t.dtype)


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream): reuse is ordered by the stream it belongs to, so
    branches of a step that run on different streams never share split-K slabs.  Buffers requested while a hipGraph is being
    captured live in that graph's private pool: they are kept in a separate table that `reset_capture_workspaces()` empties
    before the next set of graphs is captured (a buffer from the pool of a destroyed graph must not be baked into another)."""
    capturing = device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
    cache = _ws_capture_cache if capturing else _ws_cache
    key = (device.type, device.index, torch.cuda.current_stream().cuda_stream if device.type == 'cuda' else 0)
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        cache[key] = buf
    return buf


_ws_capture_cache = {}


def reset_capture_workspaces():
    _ws_capture_cache.clear()


def empty_cl(n, c, h, w, device, dtype=torch.float32):
    """Logical [n,c,h,w] tensor with NHWC physical layout (explicit, also for degenerate dims)."""
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def is_dense_like(a, b):
    return a.shape == b.shape and a.stride() == b.stride()


def is_dense(t):
    """True if t's elements occupy one gap-free block (any dim permutation)."""
    if t.numel() == 0:
        return True
    dims = sorted([(st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1])
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True


def empty_like_dense(t):
    """New tensor with the same shape AND strides as the dense tensor t."""
    assert is_dense(t)
    return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)


def same_pads(in_size, k, stride):
    """TF 'SAME': out = ceil(in/s); total = max((out-1)*s+k-in, 0); leading pad = total//2."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2


class ConvGeom:
    """Static geometry of one SAME conv (hashable; shared by fwd/dgrad/wgrad)."""
    __slots__ = ('C', 'H', 'W', 'K', 'R', 'S', 'stride', 'P', 'Q', 'pad_t', 'pad_l', 'x_up')

    def __init__(self, C, H, W, K, R, S, stride=1, x_up=False):
        self.C, self.H, self.W, self.K, self.R, self.S, self.stride, self.x_up = C, H, W, K, R, S, stride, bool(x_up)
        self.P, self.pad_t = same_pads(H, R, stride)
        self.Q, self.pad_l = same_pads(W, S, stride)

    def desc(self, N, xs, ys):
        d = ConvDesc()
        d.N, d.C, d.H, d.W, d.K, d.R, d.S = N, self.C, self.H, self.W, self.K, self.R, self.S
        d.P, d.Q, d.stride, d.pad_t, d.pad_l = self.P, self.Q, self.stride, self.pad_t, self.pad_l
        d.x_up = 1 if self.x_up else 0
        xs, ys = list(xs), list(ys)
        hp, wp = (self.H // 2, self.W // 2) if self.x_up else (self.H, self.W)
        # strides of size-1 dims are never used for addressing: zero them so that degenerate
        # layouts (Linear as a 1x1 conv on a 1x1 image) still qualify for the vector loaders
        if hp == 1: xs[2] = 0
        if wp == 1: xs[3] = 0
        if self.P == 1: ys[2] = 0
        if self.Q == 1: ys[3] = 0
        d.xs = I64x4(*xs)
        d.ys = I64x4(*ys)
        return d


def fewch_handles(g):
    """Mirror of ctgan_fewch_handles (csrc/fewch.hip): convs with <= 4 channels on one side that run on the direct
    FMA kernels (forward, data gradient and weight gradient) instead of the im2col / GEMM route."""
    taps = {(3, 3, 3), (1, 1, 3), (5, 5, 1), (3, 3, 1)}
    if g.x_up:
        return False
    if g.C <= 4 and (g.R, g.S, g.C) in taps and g.K in (64, 128, 256):
        return True
    return g.K <= 4 and g.stride == 1 and (g.R, g.S, g.K) in taps and g.C in (64, 128, 256)


def _x_phys_shape(g, N):
    return (N, g.C, g.H // 2, g.W // 2) if g.x_up else (N, g.C, g.H, g.W)


def _ext(drop, out_mask=None):
    """drop = (keep, seed, stream_id, ctr) -> ctgan_epilogue_ext*, or None.  Ranged form (forward launches shared by several
    passes): {'ranges': [(end_sample, spec or None), ...]} - consecutive sample ranges with their own dropout.
    out_mask (forward convs): tensor with the result's strides, the result is kept where it is > 0."""
    e = _ext_struct(drop, out_mask)
    return None if e is None else ctypes.byref(e)


def _act_ext(act):
    """act = {'alpha': a, 'drop': spec or ranged dict, 'ref': tensor or None} -> ctgan_epilogue_ext* with the fused LeakyReLU + dropout
    pair switched on (csrc/igemm16.hip conv16_act)."""
    e = _ext_struct(act['drop'])
    e.act, e.act_alpha = 1, float(act['alpha'])
    ref = act.get('ref')
    e.act_ref = ref.data_ptr() if ref is not None else None
    return ctypes.byref(e)


def _act_apply(y, act):
    """The LeakyReLU + dropout pair of `act` as its own launch on the dense result y (the fallback of the fused epilogue: the same draws)."""
    ref = act.get('ref')
    ref = y if ref is None else ref
    drop = act['drop']
    if not isinstance(drop, dict):
        return lrelu_dropout_rng(y, ref, act['alpha'], *drop)
    rs = drop['ranges']
    if len(rs) == 1:
        return lrelu_dropout_rng(y, ref, act['alpha'], *rs[0][1])
    assert len(rs) == 2 and rs[1][0] == y.shape[0] and rs[0][1][0] == rs[1][1][0]
    a, b = rs[0][1], rs[1][1]
    return lrelu_dropout_rng2(y, ref, rs[0][0], act['alpha'], a[0], a[1], a[2], b[2], a[3])


def _ext_struct(drop, out_mask=None):
    if drop is None and out_mask is None:
        return None
    from ._lib import EpilogueExt
    if drop is None:
        e = EpilogueExt(0.0, 0, 0, None)               # keep outside (0,1): no dropout
        e.out_mask = out_mask.data_ptr()
        return e
    assert out_mask is None
    if isinstance(drop, dict):
        rs = drop['ranges']
        assert 1 <= len(rs) <= 3
        spec0 = next(sp for _, sp in rs if sp is not None)
        e = EpilogueExt(1.0, spec0[1], 0, spec0[3].data_ptr())
        e.n_ranges = len(rs)
        for i, (end, sp) in enumerate(rs):
            assert sp is None or (sp[1] == spec0[1] and sp[3].data_ptr() == spec0[3].data_ptr())
            e.range_end[i] = int(end)
            e.range_keep[i] = float(sp[0]) if sp is not None else 1.0
            e.range_stream_id[i] = int(sp[2]) if sp is not None else 0
        return e
    keep, seed, sid, ctr = drop
    assert ctr.is_cuda and ctr.dtype == torch.int64
    return EpilogueExt(keep, seed, sid, ctr.data_ptr())


def _dropout_ranges(y, drop):
    """Fallback of the ranged epilogue dropout: one dropout pass per range, on the range's own rows."""
    r0 = 0
    for end, sp in drop['ranges']:
        if sp is not None and sp[0] < 1.0:
            y[r0:end].copy_(dropout_rng(y[r0:end], *sp))
        r0 = end
    return y


# A/B switch: the LeakyReLU + dropout pair inside the 16-bit slice kernels' epilogue (default) / as its own launch
ACT_EPILOGUE = os.environ.get('CTGAN_ACT_EPILOGUE', '1') != '0'


def _conv_act(op, a, w, bias, g, N, relu_in, act, wt=None):
    """conv_fwd (op 0) / conv_dgrad (op 1) followed by the LeakyReLU + dropout pair `act`: one launch where the 16-bit slice kernels take
    the conv (ctgan_epilogue_ext.act), otherwise the plain conv and lrelu_dropout_rng(2) - the same result bit for bit."""
    ref = act.get('ref')
    _need_dev(ref)
    if op == 0:
        assert tuple(a.shape) == _x_phys_shape(g, N) and tuple(w.shape) == (g.R, g.S, g.C, g.K) and w.is_contiguous()
        y = empty_cl(N, g.K, g.P, g.Q, a.device)
        d = g.desc(N, a.stride(), y.stride())
    else:
        assert tuple(a.shape) == (N, g.K, g.P, g.Q) and tuple(w.shape) == (g.R, g.S, g.C, g.K) and w.is_contiguous()
        y = empty_cl(N, g.C, g.H, g.W, a.device)
        d = g.desc(N, y.stride(), a.stride())
    if ref is not None:
        assert tuple(ref.shape) == tuple(y.shape) and ref.stride() == y.stride()
    mode = _conv_mode(d, op, ACT_EPILOGUE and MMA_DTYPE in ('bf16', 'f16') and not fewch_handles(g) and not g.x_up)
    if mode is not None:
        wp = _packed16(w, d, op, g, mode)
        code = _MMA_CODE[mode]
        nb = lib.ctgan_conv2d16_workspace_bytes(ctypes.byref(d), op)
        ws = workspace(nb, a.device) if nb else None
        try:
            if op == 0:
                _timed(g, N, lambda: check(lib.ctgan_conv2d16_fwd_ex(ctypes.byref(d), code, _ptr(a), _ptr(wp), _ptr(bias), None, _ptr(y), 2 if relu_in else 0,
                                                                       _act_ext(act), _ptr(ws), nb, _stream()), 'conv2d16_fwd_ex'))
            else:
                _timed(g, N, lambda: check(lib.ctgan_conv2d16_dgrad_ex(ctypes.byref(d), code, _ptr(a), _ptr(wp), None, None, None, _ptr(y), 0,
                                                                         _act_ext(act), _ptr(ws), nb, _stream()), 'conv2d16_dgrad_ex'))
            return y
        except NotImplementedError:
            pass
    y = conv_fwd(a, w, bias, g, relu_in=relu_in) if op == 0 else conv_dgrad(a, w, g, N, wt=wt)
    return _act_apply(y, act)


def conv_fwd(x, w, bias, g, resid=None, relu=False, out_strides=None, relu_in=False, drop=None, resid_up=False, mask=None, act=None):
    """y = conv(x,w) [+bias] [kept where mask > 0] [+resid] [relu] [dropout]; relu_in: conv(relu(x)).  x logical [N,C,H(/2),W(/2)],
    w HWIO.  drop = (keep, seed, stream_id, ctr): tf.nn.dropout of the result inside the epilogue (== dropout_rng(y, ...)).
    mask (same shape and strides as the result; only without resid / relu / drop): == lrelu_bwd(conv(x,w)+bias, mask, 0).
    act = {'alpha', 'drop', 'ref'} (only with bias): the LeakyReLU + dropout pair on the result (== lrelu_dropout_rng(y, ref or y, ...));
    inside the epilogue of the 16-bit slice kernels, its own launch elsewhere."""
    _need_dev(x, w, bias, resid, mask)
    if act is not None:
        assert resid is None and not relu and drop is None and mask is None and out_strides is None
        return _conv_act(0, x, w, bias, g, x.shape[0], relu_in, act)
    if mask is not None:
        assert resid is None and not relu and drop is None
        return _conv_fwd_masked(x, w, bias, g, out_strides, relu_in, mask)
    N = x.shape[0]
    assert tuple(x.shape) == _x_phys_shape(g, N), (tuple(x.shape), _x_phys_shape(g, N))
    assert tuple(w.shape) == (g.R, g.S, g.C, g.K) and w.is_contiguous()
    if out_strides is None:
        y = empty_cl(N, g.K, g.P, g.Q, x.device)
    else:
        y = torch.empty_strided((N, g.K, g.P, g.Q), out_strides, dtype=torch.float32, device=x.device)
    if resid is not None and resid_up:
        assert tuple(resid.shape) == (N, g.K, g.P // 2, g.Q // 2) and resid.permute(0, 2, 3, 1).is_contiguous()
    elif resid is not None:
        assert is_dense_like(resid, y)
    d = g.desc(N, x.stride(), y.stride())
    fl = (1 if relu else 0) | (2 if relu_in else 0) | (8 if (resid_up and resid is not None) else 0)
    plain = drop is None and not fewch_handles(g)
    mode = _conv_mode(d, 0, plain and not (fl & 8))
    if (mode is None and (fl & 8) and plain and (MMA_DTYPE == 'f32x3' or (MMA_DTYPE is None and X3_HYBRID))
            and lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), 0)):
        mode = 'f32x3'                                # the halo-patch kernel reads the residual through the 2x upsample itself
    _x3_log(g, N, d, 0, drop=drop is not None, resid_up=bool(fl & 8), x_up=bool(g.x_up))
    if (mode is None and drop is not None and not (fl & 8) and g.stride == 1 and not fewch_handles(g)
            and (MMA_DTYPE == 'f32x3' or (MMA_DTYPE is None and X3_HYBRID)) and lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), 0)):
        # the halo-patch kernel has the fp32 family's dropout epilogue (same Philox draws at the same offsets)
        wp = _packed16(w, d, 0, g, 'f32x3')
        try:
            _timed(g, N, lambda: check(lib.ctgan_conv2d16_fwd_ex(ctypes.byref(d), 3, _ptr(x), _ptr(wp), _ptr(bias), _ptr(resid), _ptr(y), fl, _ext(drop), None, 0, _stream()), 'conv2d16_fwd_ex'))
            return y
        except NotImplementedError:
            pass
    if mode is not None:
        wp = _packed16(w, d, 0, g, mode)
        code = _MMA_CODE[mode]
        nb = lib.ctgan_conv2d16_workspace_bytes(ctypes.byref(d), 0)
        ws = workspace(nb, x.device) if nb else None
        _timed(g, N, lambda: check(lib.ctgan_conv2d16_fwd(ctypes.byref(d), code, _ptr(x), _ptr(wp), _ptr(bias), _ptr(resid), _ptr(y), fl, _ptr(ws), nb, _stream()), 'conv2d16_fwd'))
        return y
    if fl & 8:
        try:
            _timed(g, N, lambda: check(lib.ctgan_conv2d_fwd_ex(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(resid), _ptr(y), fl, _ext(drop), _stream()), 'conv2d_fwd'))
            return y
        except NotImplementedError:          # no vector epilogue: materialise the upsampled residual
            resid = upsample2(resid, 1.0)
            fl &= ~8
    if drop is not None:
        try:
            _timed(g, N, lambda: check(lib.ctgan_conv2d_fwd_ex(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(resid), _ptr(y), fl, _ext(drop), _stream()), 'conv2d_fwd'))
            return y
        except NotImplementedError:          # no vector epilogue for this call: dropout as its own pass
            pass
    _timed(g, N, lambda: check(lib.ctgan_conv2d_fwd(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(resid), _ptr(y), fl, _stream()), 'conv2d_fwd'))
    if drop is None:
        return y
    return _dropout_ranges(y, drop) if isinstance(drop, dict) else dropout_rng(y, *drop)


def _conv_fwd_masked(x, w, bias, g, out_strides, relu_in, mask):
    """conv_fwd with the out_mask epilogue (ctgan_epilogue_ext.out_mask) on whichever kernel family serves the launch."""
    N = x.shape[0]
    assert tuple(x.shape) == _x_phys_shape(g, N) and tuple(w.shape) == (g.R, g.S, g.C, g.K) and w.is_contiguous()
    if out_strides is None:
        y = empty_cl(N, g.K, g.P, g.Q, x.device)
    else:
        y = torch.empty_strided((N, g.K, g.P, g.Q), out_strides, dtype=torch.float32, device=x.device)
    if tuple(mask.shape) != tuple(y.shape) or mask.stride() != y.stride():
        return lrelu_bwd(conv_fwd(x, w, bias, g, out_strides=out_strides, relu_in=relu_in), mask, 0.0)
    d = g.desc(N, x.stride(), y.stride())
    fl = 2 if relu_in else 0
    mode = _conv_mode(d, 0, not fewch_handles(g))
    _x3_log(g, N, d, 0, drop=False, resid_up=False, x_up=bool(g.x_up))
    try:
        if mode is not None:
            wp = _packed16(w, d, 0, g, mode)
            nb = lib.ctgan_conv2d16_workspace_bytes(ctypes.byref(d), 0)
            ws = workspace(nb, x.device) if nb else None
            _timed(g, N, lambda: check(lib.ctgan_conv2d16_fwd_ex(ctypes.byref(d), _MMA_CODE[mode], _ptr(x), _ptr(wp), _ptr(bias), None, _ptr(y), fl,
                                                                 _ext(None, mask), _ptr(ws), nb, _stream()), 'conv2d16_fwd_ex'))
        else:
            _timed(g, N, lambda: check(lib.ctgan_conv2d_fwd_ex(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), None, _ptr(y), fl, _ext(None, mask),
                                                               _stream()), 'conv2d_fwd'))
        return y
    except NotImplementedError:
        return lrelu_bwd(conv_fwd(x, w, bias, g, out_strides=out_strides, relu_in=relu_in), mask, 0.0)


def repack_filter(w, g):
    """The filter in the layout conv_dgrad multiplies with (opaque: stride 1 -> wT[r',s',k,c] =
    w[R-1-r',S-1-s',c,k]; stride 2 -> the four output-phase sub-filters); reuse it via `wt=`."""
    _need_dev(w)
    assert tuple(w.shape) == (g.R, g.S, g.C, g.K) and w.is_contiguous()
    d = g.desc(1, (0, 0, 0, 0), (0, 0, 0, 0))
    wt = torch.empty(lib.ctgan_conv2d_workspace_bytes(ctypes.byref(d), 1) // 4, dtype=torch.float32, device=w.device)
    check(lib.ctgan_conv2d_repack_filter(ctypes.byref(d), _ptr(w), _ptr(wt), _stream()), 'conv2d_repack_filter')
    return wt


_DGRAD16 = {}


def dgrad_runs_16bit(g):
    """bf16 / fp16 modes: does the plain data gradient of geometry g (dense channels-last operands) run on the 16-bit family?  It then
    reads its own packed image of the filter and never the fp32 family's repacked one - callers skip building that (functional._repacked)."""
    if MMA_DTYPE not in ('bf16', 'f16') or g.x_up or fewch_handles(g):
        return False
    key = (MMA_DTYPE, g.C, g.H, g.W, g.K, g.R, g.S, g.stride, g.pad_t, g.pad_l)
    r = _DGRAD16.get(key)
    if r is None:
        d = g.desc(1, (g.C * g.H * g.W, 1, g.W * g.C, g.C), (g.K * g.P * g.Q, 1, g.Q * g.K, g.K))
        r = _DGRAD16[key] = bool(lib.ctgan_conv2d16_supported(ctypes.byref(d), 1, _MMA_CODE[MMA_DTYPE]))
    return r


def dgrad_wants_repack(g):
    """True when conv_dgrad can run the vector kernels on a pre-repacked filter."""
    small_linear = g.R == 1 and g.S == 1 and g.H == 1 and g.W == 1 and g.K <= 16
    return g.C % 4 == 0 and g.K % 32 == 0 and not small_linear


def conv_dgrad(gy, w, g, N, out_strides=None, bias=None, wt=None, mask=None, resid=None, drop=None, act=None):
    """dx = conv^T(gy, w) [+ bias] [kept where mask > 0] [+ resid] [dropout]; dx logical [N,C,H,W] (channels-last
    unless out_strides given).  `wt` = repack_filter(w, g) computed earlier (skips the per-call repack).
    drop = (keep, seed, stream_id, ctr): the result is multiplied by that dropout's mask (== dropout_rng(dx, ...)).
    act = {'alpha', 'drop', 'ref'}: the backward of a LeakyReLU + dropout pair on the result (== lrelu_dropout_rng(dx, ref, ...), ref =
    the pair's forward result) - conv_fwd's act."""
    _need_dev(gy, w, bias, wt, mask, resid)
    assert not g.x_up
    if act is not None:
        assert bias is None and mask is None and resid is None and drop is None and out_strides is None
        return _conv_act(1, gy, w, None, g, N, False, act, wt=wt)
    assert tuple(gy.shape) == (N, g.K, g.P, g.Q)
    assert tuple(w.shape) == (g.R, g.S, g.C, g.K) and w.is_contiguous()
    if out_strides is None:
        dx = empty_cl(N, g.C, g.H, g.W, gy.device)
    else:
        dx = torch.empty_strided((N, g.C, g.H, g.W), out_strides, dtype=torch.float32, device=gy.device)
    d = g.desc(N, dx.stride(), gy.stride())
    if mask is not None:
        mask = match_layout(mask, dx)
    if resid is not None:
        resid = match_layout(resid, dx)
    mode = _conv_mode(d, 1, drop is None and not fewch_handles(g))
    _x3_log(g, N, d, 1, drop=drop is not None)
    if (mode is None and drop is not None and g.stride == 1 and not fewch_handles(g)
            and (MMA_DTYPE == 'f32x3' or (MMA_DTYPE is None and X3_HYBRID)) and lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), 1)):
        # the halo-patch kernels carry the fp32 family's epilogue dropout (same Philox draws at the same offsets)
        wp = _packed16(w, d, 1, g, 'f32x3')
        try:
            _timed(g, N, lambda: check(lib.ctgan_conv2d16_dgrad_ex(ctypes.byref(d), 3, _ptr(gy), _ptr(wp), _ptr(bias), _ptr(mask), _ptr(resid), _ptr(dx), 0,
                                                                     _ext(drop), None, 0, _stream()), 'conv2d16_dgrad_ex'))
            return dx
        except NotImplementedError:
            pass
    if mode is not None:
        wp = _packed16(w, d, 1, g, mode)
        code = _MMA_CODE[mode]
        nb = lib.ctgan_conv2d16_workspace_bytes(ctypes.byref(d), 1)
        ws = workspace(nb, gy.device) if nb else None
        _timed(g, N, lambda: check(lib.ctgan_conv2d16_dgrad(ctypes.byref(d), code, _ptr(gy), _ptr(wp), _ptr(bias), _ptr(mask), _ptr(resid), _ptr(dx), 0, _ptr(ws), nb, _stream()), 'conv2d16_dgrad'))
        return dx
    if wt is not None:
        assert wt.numel() * 4 == lib.ctgan_conv2d_workspace_bytes(ctypes.byref(d), 1)
        filt, ws_p, ws_n, fl = wt, None, 0, 1
    else:
        ws = workspace(lib.ctgan_conv2d_workspace_bytes(ctypes.byref(d), 1), gy.device)
        filt, ws_p, ws_n, fl = w, ws, ws.numel(), 0
    if drop is not None:
        try:
            _timed(g, N, lambda: check(lib.ctgan_conv2d_dgrad_ex(ctypes.byref(d), _ptr(gy), _ptr(filt), _ptr(bias), _ptr(mask), _ptr(resid), _ptr(dx), _ptr(ws_p), ws_n, fl, _ext(drop), _stream()), 'conv2d_dgrad'))
            return dx
        except NotImplementedError:
            pass
    _timed(g, N, lambda: check(lib.ctgan_conv2d_dgrad(ctypes.byref(d), _ptr(gy), _ptr(filt), _ptr(bias), _ptr(mask), _ptr(resid), _ptr(dx), _ptr(ws_p), ws_n, fl, _stream()), 'conv2d_dgrad'))
    if drop is None:
        return dx
    return _dropout_ranges(dx, drop) if isinstance(drop, dict) else dropout_rng(dx, *drop)


def conv_wgrad(x, gy, g, with_bias=False, relu_x=False, out=None):
    """dw[R,S,C,K] = sum over pixels of x (gathered) * gy; with_bias also returns db[K] = sum of gy.
    out = (dw, db or None): result buffers to write into (default: fresh tensors)."""
    _need_dev(x, gy)
    N = x.shape[0]
    assert tuple(x.shape) == _x_phys_shape(g, N)
    assert tuple(gy.shape) == (N, g.K, g.P, g.Q)
    if out is not None:
        dw, db = out
        assert tuple(dw.shape) == (g.R, g.S, g.C, g.K) and dw.is_contiguous() and (db is not None) == bool(with_bias)
    else:
        dw = torch.empty((g.R, g.S, g.C, g.K), dtype=torch.float32, device=x.device)
        db = torch.empty(g.K, dtype=torch.float32, device=x.device) if with_bias else None
    if with_bias and not gy.permute(0, 2, 3, 1).is_contiguous():
        gy = to_channels_last(gy)
    d = g.desc(N, x.stride(), gy.stride())
    if (MMA_DTYPE is not None or X3_HYBRID) and not fewch_handles(g) and not g.x_up:
        gy16 = gy if gy.permute(0, 2, 3, 1).is_contiguous() else to_channels_last(gy)
        d16 = g.desc(N, x.stride(), gy16.stride())
        mode = _conv_mode(d16, 2, True)
        if mode is not None:
            ws = workspace(lib.ctgan_conv2d16_wgrad_workspace_bytes(ctypes.byref(d16), _MMA_CODE[mode]), x.device)
            code = _MMA_CODE[mode]
            fused_b = with_bias and g.K % 4 == 0         # bias gradient inside the same launch (no separate column-sum pass)
            fl16 = 2 if relu_x else 0

            def launch(extra=0):
                check(lib.ctgan_conv2d16_wgrad_bias(ctypes.byref(d16), code, _ptr(x), _ptr(gy16), _ptr(dw), _ptr(db) if fused_b else None,
                                                    _ptr(ws), ws.numel(), fl16 | extra, _stream()), 'conv2d16_wgrad')
            if PROFILE is None:
                launch()
            else:               # bench.py: the GEMM alone inside the event bracket (what rocprofv3 lists under its symbol), its reduction after it
                _timed(g, N, lambda: launch(0x100))
                launch(0x200)
            if with_bias:
                if not fused_b:
                    db.copy_(colsum_channels(gy16))
                return dw, db
            return dw
    nb = lib.ctgan_conv2d_workspace_bytes(ctypes.byref(d), 2)
    ws = workspace(nb, x.device)
    _timed(g, N, lambda: check(lib.ctgan_conv2d_wgrad(ctypes.byref(d), _ptr(x), _ptr(gy), _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), 2 if relu_x else 0, _stream()), 'conv2d_wgrad'))
    return (dw, db) if with_bias else dw


WGRAD_MAX_SEGS = 3


def conv_wgrad_multi(segs, g, dw, db=None):
    """dw (and db) = sum over segments of the weight (bias) gradient: segs = [(x, gy, relu_x, with_bias), ...], all of
    geometry g and identical strides.  One launch + one reduction (csrc: ctgan_conv2d_wgrad_multi); raises
    NotImplementedError for shapes outside the pipelined kernel."""
    n = len(segs)
    assert 1 <= n <= WGRAD_MAX_SEGS
    x0, gy0 = segs[0][0], segs[0][1]
    for x, gy, _, _ in segs:
        _need_dev(x, gy)
        assert tuple(x.shape[1:]) == _x_phys_shape(g, 1)[1:] and tuple(gy.shape[1:]) == (g.K, g.P, g.Q)
        assert x.stride() == x0.stride() or x.shape[0] == 1 or x0.shape[0] == 1 or x.stride()[1:] == x0.stride()[1:]
        assert x.stride()[1:] == x0.stride()[1:] and gy.stride()[1:] == gy0.stride()[1:] and x.shape[0] == gy.shape[0]
    assert tuple(dw.shape) == (g.R, g.S, g.C, g.K) and dw.is_contiguous()
    d = g.desc(x0.shape[0], x0.stride(), gy0.stride())
    # per-image strides must not depend on the segment's batch size (dense tensors: they do not)
    Ns = (ctypes.c_int32 * n)(*[sg[0].shape[0] for sg in segs])
    X = (ctypes.c_void_p * n)(*[sg[0].data_ptr() for sg in segs])
    Y = (ctypes.c_void_p * n)(*[sg[1].data_ptr() for sg in segs])
    Fl = (ctypes.c_int32 * n)(*[(2 if sg[2] else 0) | (4 if sg[3] else 0) for sg in segs])
    nb = lib.ctgan_conv2d_wgrad_multi_workspace_bytes(ctypes.byref(d), n, Ns)
    ws = workspace(nb, x0.device)
    tot = sum(sg[0].shape[0] for sg in segs)
    _timed(g, tot, lambda: check(lib.ctgan_conv2d_wgrad_multi(ctypes.byref(d), n, X, Y, Ns, Fl, _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                                 'conv2d_wgrad_multi'))
    return dw, db


WGRAD_GROUP_LIMIT = 32


def conv_wgrad_group(groups):
    """groups = [(segs, g, dw, db), ...] - each entry as for conv_wgrad_multi, for DIFFERENT filters: one launch per
    tile configuration for all of them plus one launch for all split-K reductions (csrc: ctgan_conv2d_wgrad_group).
    Raises NotImplementedError (nothing launched) if any entry is outside the pipelined kernel."""
    from ._lib import WgradGroup
    n = len(groups)
    assert 1 <= n <= WGRAD_GROUP_LIMIT
    arr = (WgradGroup * n)()
    dev = groups[0][0][0][0].device
    for i, grp in enumerate(groups):
        segs, g, dw, db = grp[:4]
        add_dw, add_db = (grp[4], grp[5]) if len(grp) > 4 else (None, None)     # finished addends (may be dw / db themselves)
        assert 1 <= len(segs) <= WGRAD_MAX_SEGS
        x0, gy0 = segs[0][0], segs[0][1]
        for x, gy, _, _ in segs:
            _need_dev(x, gy)
            assert tuple(x.shape[1:]) == _x_phys_shape(g, 1)[1:] and tuple(gy.shape[1:]) == (g.K, g.P, g.Q)
            assert x.stride()[1:] == x0.stride()[1:] and gy.stride()[1:] == gy0.stride()[1:] and x.shape[0] == gy.shape[0]
        assert tuple(dw.shape) == (g.R, g.S, g.C, g.K) and dw.is_contiguous()
        assert (db is not None) == any(sg[3] for sg in segs)
        G = arr[i]
        G.d = g.desc(x0.shape[0], x0.stride(), gy0.stride())
        G.nseg = len(segs)
        for k, sg in enumerate(segs):
            G.Ns[k] = sg[0].shape[0]
            G.seg_flags[k] = (2 if sg[2] else 0) | (4 if sg[3] else 0)
            G.xs[k] = sg[0].data_ptr()
            G.dys[k] = sg[1].data_ptr()
        G.dw = dw.data_ptr()
        G.db = db.data_ptr() if db is not None else None
        assert add_dw is None or (add_dw.shape == dw.shape and add_dw.is_contiguous())
        G.add_dw = add_dw.data_ptr() if add_dw is not None else None
        G.add_db = add_db.data_ptr() if (add_db is not None and db is not None) else None
    # hybrid fp32 mode / split mode: the members the split-mode kernels take ride ONE grouped split-mode call (filter-column kernel +
    # slice kernel, csrc/wgrad16c.hip / igemm16.hip).  All-or-nothing: every member is validated - the split-mode subset AND the fp32
    # family's remainder - before the first launch, so that a NotImplementedError always means "nothing was launched" (ADVICE r3).
    x3 = []
    gmode = grouped16_mode()
    if gmode is not None:
        code = _MMA_CODE[gmode]
        x3 = [i for i in range(n) if lib.ctgan_conv2d16_wgrad_group_workspace_bytes(ctypes.byref(arr[i]), 1, code) > 0]
    rest = [i for i in range(n) if i not in set(x3)]
    arr_r = None
    if x3 and rest:
        arr_r = (WgradGroup * len(rest))()
        for k, i in enumerate(rest):
            arr_r[k] = arr[i]
        if lib.ctgan_conv2d_wgrad_group_workspace_bytes(arr_r, len(rest)) == 0:
            raise NotImplementedError('conv2d_wgrad_group: unsupported group')
    if x3:
        arr3 = (WgradGroup * len(x3))()
        for k, i in enumerate(x3):
            arr3[k] = arr[i]
        nb3 = lib.ctgan_conv2d16_wgrad_group_workspace_bytes(arr3, len(x3), code)
        if nb3 == 0:
            raise NotImplementedError('conv2d16_wgrad_group: unsupported group')
        ws3 = workspace(nb3, dev)
        if PROFILE is None:
            for _ in range(WGRAD_GROUP_EXTRA):
                # bench.py's in-situ timing: the filter-column kernel launched once more (same operands, same result) - the replay time of a
                # step graph captured with this switch minus that of the step's own graph = the kernel's time where it runs
                check(lib.ctgan_conv2d16_wgrad_group(arr3, len(x3), code, _ptr(ws3), ws3.numel(), 1 | 16, _stream()), 'conv2d16_wgrad_group')
            check(lib.ctgan_conv2d16_wgrad_group(arr3, len(x3), code, _ptr(ws3), ws3.numel(), 3, _stream()), 'conv2d16_wgrad_group')
        else:
            st = torch.cuda.current_stream()
            # ONE launch per bracket: this launch is hundreds of microseconds long (the ~10 us of event overhead is 3 %), and repeated back
            # to back it runs 15 % slower than in the step - the sustained MFMA stream pulls the clock down (410 vs 333 us, round 4)
            # (the two kernels of a mixed call - filter-column and slice - in brackets of their own, each under its own device symbol)
            fl = [_conv_flops(groups[i][1], sum(sg[0].shape[0] for sg in groups[i][0])) for i in x3]
            for which in (0, 1):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                # (a spin of ~0.1 ms in front of the bracket: the host has recorded e0 and submitted the launch before the device gets there -
                # on an idle queue the bracket would include the host's submission time)
                torch.cuda._sleep(200000)
                with ClockProbe() as probe:
                    e0.record(st)
                    check(lib.ctgan_conv2d16_wgrad_group(arr3, len(x3), code, _ptr(ws3), ws3.numel(), 1 | (16 << which), _stream()), 'conv2d16_wgrad_group')
                    e1.record(st)
                mask = int(lib.ctgan_debug_last_wgrad_group_col_mask())
                flops = sum(f for k, f in enumerate(fl) if bool((mask >> k) & 1) == (which == 0))
                if flops:
                    PROFILE.append((last_kernel(), flops, e0, e1, 1, ('group', len(x3)), last_symbol()))
                    PROFILE_CLOCKS.append((last_symbol(), flops, probe))
            check(lib.ctgan_conv2d16_wgrad_group(arr3, len(x3), code, _ptr(ws3), ws3.numel(), 2, _stream()), 'conv2d16_wgrad_group')
        if not rest:
            return
        groups = [groups[i] for i in rest]
        arr, n = arr_r, len(rest)
    nb = lib.ctgan_conv2d_wgrad_group_workspace_bytes(arr, n)
    if nb == 0:
        raise NotImplementedError('conv2d_wgrad_group: unsupported group')
    ws = workspace(nb, dev)            # (stream order: the split-mode launch above has read its slabs before these are written)
    if PROFILE is None:
        check(lib.ctgan_conv2d_wgrad_group(arr, n, _ptr(ws), ws.numel(), _stream()), 'conv2d_wgrad_group')
        return
    # bench.py's roofline leg: each grouped GEMM launch (one per tile configuration) alone inside its own event bracket (what rocprofv3
    # lists under that symbol), the batched split-K reduction after them
    st = torch.cuda.current_stream()
    for t in range(4):
        members = [i for i in range(n) if lib.ctgan_conv2d_wgrad_group_tile(ctypes.byref(arr[i])) == t]
        if not members:
            continue
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(PROFILE_REPS):
            check(lib.ctgan_conv2d_wgrad_group_ex(arr, n, _ptr(ws), ws.numel(), 1 | (16 << t), _stream()), 'conv2d_wgrad_group')
        e1.record(st)
        flops = sum(_conv_flops(groups[i][1], sum(sg[0].shape[0] for sg in groups[i][0])) for i in members)
        PROFILE.append((last_kernel(), flops, e0, e1, PROFILE_REPS, ('group', len(members)), last_symbol()))
    check(lib.ctgan_conv2d_wgrad_group_ex(arr, n, _ptr(ws), ws.numel(), 2, _stream()), 'conv2d_wgrad_group')


def im2col(x, g, cpad):
    """x logical [N,C,H,W] (any strides) -> channels-last [N,cpad,P,Q] patch tensor."""
    _need_dev(x)
    N = x.shape[0]
    assert tuple(x.shape) == (N, g.C, g.H, g.W) and not g.x_up
    cols = empty_cl(N, cpad, g.P, g.Q, x.device)
    d = g.desc(N, x.stride(), cols.stride())
    check(lib.ctgan_im2col(ctypes.byref(d), _ptr(x), cpad, _ptr(cols), _stream()), 'im2col')
    return cols


def col2im(cols, g, N, out_strides=None):
    """adjoint of im2col: channels-last [N,cpad,P,Q] -> dx logical [N,C,H,W]."""
    _need_dev(cols)
    cpad = cols.shape[1]
    assert tuple(cols.shape) == (N, cpad, g.P, g.Q) and cols.permute(0, 2, 3, 1).is_contiguous()
    if out_strides is None:
        dx = empty_cl(N, g.C, g.H, g.W, cols.device)
    else:
        dx = torch.empty_strided((N, g.C, g.H, g.W), out_strides, dtype=torch.float32, device=cols.device)
    d = g.desc(N, dx.stride(), cols.stride())
    check(lib.ctgan_col2im(ctypes.byref(d), _ptr(cols), cpad, _ptr(dx), _stream()), 'col2im')
    return dx


def last_kernel():
    return lib.ctgan_last_kernel().decode()


def last_symbol():
    """Device symbol of the last conv launch as rocprofv3 prints it (the variant name where the launcher records none)."""
    return lib.ctgan_last_symbol().decode()


def debug_force_generic(on):
    """Tests only: route every conv through the table-driven generic kernels."""
    lib.ctgan_debug_force_generic(1 if on else 0)


def debug_x3_halo_version(v):
    """Tests only: 1 = halo-patch kernel with the filter staged through LDS, 2 = filter fragments streamed from L2, 0 = default."""
    lib.ctgan_debug_x3_halo_version(int(v))


def debug_x3_s2fwd(on):
    """Tests / A-B: False = the stride-2 forward launches of the split mode on the slice kernel instead of conv16x3sf_kernel; 2 = on that
    kernel, without its K split for launches of 128 .. 383 tiles."""
    lib.ctgan_debug_x3_s2fwd(2 if on == 2 else (1 if on else 0))


def debug_x3_s2halo(on):
    """Tests / A-B: False = the stride-2 data gradients of the split mode on the slice kernel instead of the four-phase halo kernel."""
    lib.ctgan_debug_x3_s2halo(1 if on else 0)


def debug_m2f_px(on):
    """Tests / A-B: False = the 3x3 many -> few convs on the row-ring kernel instead of the one-pixel-per-lane kernel (csrc/fewch.hip)."""
    lib.ctgan_debug_m2f_px(1 if on else 0)


def debug_x3_hk(mode, max_wgs=0):
    """Tests / A-B: 0 = no launch on conv16x3hk_kernel (one channel chunk per wave), 1 = launches of <= max_wgs workgroups (default 768), 2 = all that qualify."""
    lib.ctgan_debug_x3_hk(int(mode), int(max_wgs))


if os.environ.get('CTGAN_X3_HK') is not None:      # (bench A/B: "0", "2", or "1:<max workgroups>")
    _hk = os.environ['CTGAN_X3_HK'].split(':')
    debug_x3_hk(int(_hk[0]), int(_hk[1]) if len(_hk) > 1 else 0)
if os.environ.get('CTGAN_M2F_PX') == '0':
    debug_m2f_px(False)
if os.environ.get('CTGAN_X3_S2HALO') == '0':      # (bench A/B; the routing query ctgan_conv2d16_x3_prefers follows the switch)
    debug_x3_s2halo(False)
if os.environ.get('CTGAN_X3_S2DGRAD_SF') == '-1':  # (bench A/B: the four-phase data gradients back on conv16x3p_kernel)
    lib.ctgan_debug_x3_s2dgrad_sf(-1)
if os.environ.get('CTGAN_X3_S2FWD') in ('0', '2'):   # (bench A/B: the strided forward launches of the split mode on the slice kernel / without the K split)
    debug_x3_s2fwd(int(os.environ['CTGAN_X3_S2FWD']))


def debug_last_wgrad_group_kinds():
    """Tests only: bit 0 = the last grouped 16-bit weight-gradient call launched the filter-column kernel, bit 1 = the slice kernel."""
    return int(lib.ctgan_debug_last_wgrad_group_kinds())


def chain8x8_supported(x, C, H, W):
    """Can conv_chain8x8 take chains on these images?  8 x 8 x 128, dense channels-last device tensors, and a mode in which the stride-1 3x3
    layers run in the split mode anyway (the hybrid fp32 routing or 'f32x3')."""
    return bool(x.is_cuda and C == 128 and H == 8 and W == 8 and (MMA_DTYPE == 'f32x3' or (MMA_DTYPE is None and X3_HYBRID))
                and x.dtype == torch.float32 and x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous())


def chain8x8_usable(x, C, H, W):
    """... and does the step want them (the CHAIN8X8 switch)?"""
    return bool(CHAIN8X8 and chain8x8_supported(x, C, H, W))


# A/B switch (off: measured level-to-slower, 12.85 against 12.80 ms per iteration - the chain is bound by the rate at which one CU can pull a layer's
# filter fragments): the 8x8 blocks of the critic's backward passes as one launch per chain (csrc/chain8x8.hip) instead of one per conv
CHAIN8X8 = os.environ.get('CTGAN_CHAIN8X8', '0') == '1'


def conv_chain8x8(x, steps, drops=(), seed=0, ctr=None):
    """Up to four 3x3 SAME convs on 8 x 8 x 128 images, one image per workgroup (ctgan_conv2d16_chain8x8).

    steps[0] is the pre step (no filter), steps[1:] the convs; each a dict with
      'w', 'op'   (convs only) the 3x3x128x128 HWIO filter and 0 = conv(., w) / 1 = its data gradient conv^T(., w)
      'mask'      keep the value where this tensor is > 0            'resid'  1 / 2: add the value saved in that slot
      'drop'      1 / 2: multiply by the dropout mask drops[drop-1]  'save'   1 / 2: save the value (after the dropout) in that slot
      'post_mask' keep where > 0 (after the save)                    'out'    True: return the step's result
    drops: up to two (keep, stream_id_lo, stream_id_hi, n_split): images below n_split draw stream_id_lo indexed from image 0, the others
    stream_id_hi indexed from image n_split (the two row ranges of a merged pass; n_split = 0: one stream).  seed / ctr: the Philox seed and the
    device step counter.  Returns the list of the requested results (dense channels-last, in step order)."""
    from ._lib import Chain8x8
    N, C, H, W = x.shape
    assert chain8x8_supported(x, C, H, W) and 2 <= len(steps) <= 5 and len(drops) <= 2
    _need_dev(x)
    c = Chain8x8()
    c.x, c.n_images, c.channels, c.height, c.width, c.n_convs = x.data_ptr(), N, C, H, W, len(steps) - 1
    g = ConvGeom(C, H, W, C, 3, 3, 1)
    outs, keep = [], []
    for i, st in enumerate(steps):
        cs = c.step[i]
        if i:
            w = st['w']
            assert tuple(w.shape) == (3, 3, C, C)
            d = g.desc(N, x.stride(), x.stride())
            wp = _packed16(w, d, int(st['op']), g, 'f32x3')
            keep.append(wp)
            cs.wp = wp.data_ptr()
        for key in ('mask', 'post_mask'):
            t = st.get(key)
            if t is not None:
                _need_dev(t)
                assert tuple(t.shape) == (N, C, H, W) and t.stride() == x.stride(), key
                setattr(cs, key, t.data_ptr())
        cs.resid, cs.save, cs.drop = int(st.get('resid', 0)), int(st.get('save', 0)), int(st.get('drop', 0))
        if st.get('out'):
            o = empty_cl(N, C, H, W, x.device)
            outs.append(o)
            cs.out = o.data_ptr()
    for i, (kp, lo, hi, split) in enumerate(drops):
        c.drop[i].keep, c.drop[i].stream_id_lo, c.drop[i].stream_id_hi, c.drop[i].n_split = float(kp), int(lo), int(hi), int(split)
    c.drop_seed = int(seed)
    if ctr is not None:
        assert ctr.is_cuda and ctr.dtype == torch.int64
        c.drop_ctr = ctr.data_ptr()
    flops = 2.0 * N * H * W * C * 9 * C * (len(steps) - 1)
    if PROFILE is None:
        check(lib.ctgan_conv2d16_chain8x8(ctypes.byref(c), _stream()), 'conv2d16_chain8x8')
    else:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        stq = torch.cuda.current_stream()
        with ClockProbe() as probe:
            e0.record(stq)
            for _ in range(PROFILE_REPS):
                check(lib.ctgan_conv2d16_chain8x8(ctypes.byref(c), _stream()), 'conv2d16_chain8x8')
            e1.record(stq)
        PROFILE.append((last_kernel(), flops, e0, e1, PROFILE_REPS, (N, C, H, W, C, 3, 1, len(steps) - 1), last_symbol()))
        PROFILE_CLOCKS.append((last_symbol(), flops, probe))
    return outs


class ClockProbe:
    """bench.py's roofline leg: the shader clock the chip sustains while the launches inside the `with` block run.

        with K.ClockProbe() as p:
            <launches on the current stream>
        torch.cuda.synchronize(); p.mhz()

    A one-wave kernel (ctgan_debug_clock_probe) on a side stream reads s_memrealtime (constant 100 MHz) and s_memtime (shader cycles) when
    the block is entered - the side stream waits for an event recorded there - and again when the current stream reaches the end of the
    block (a flag it sets), at the latest after `cap_ms`: it cannot outlive that even if both streams share a hardware queue."""
    _side = None

    def __init__(self, cap_ms=50.0):
        self.cap_ticks = int(cap_ms * 1e5)

    def __enter__(self):
        dev = torch.device('cuda', torch.cuda.current_device())
        if ClockProbe._side is None:
            ClockProbe._side = torch.cuda.Stream()
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.out = torch.zeros(5, dtype=torch.int64, device=dev)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        ClockProbe._side.wait_event(ev)
        check(lib.ctgan_debug_clock_probe(_ptr(self.flag), self.cap_ticks, _ptr(self.out), ctypes.c_void_p(ClockProbe._side.cuda_stream)), 'clock_probe')
        return self

    def __exit__(self, *exc):
        self.flag.fill_(1)                 # on the current stream, after the block's launches
        return False

    def result(self):
        """(shader MHz, region microseconds, flag_seen) - call after a synchronize."""
        r0, s0, r1, s1, seen = [int(v) for v in self.out.cpu().tolist()]
        dr = max(r1 - r0, 1)
        return 100.0 * (s1 - s0) / dr, dr / 100.0, bool(seen)

    def mhz(self):
        return self.result()[0]


def colsum_channels(gy):
    """sum over (n,h,w) of a channels-last [N,K,P,Q] tensor -> [K] (bias gradient)."""
    _need_dev(gy)
    N, K, P, Q = gy.shape
    if not gy.permute(0, 2, 3, 1).is_contiguous():
        gy = to_channels_last(gy)
    out = torch.empty(K, dtype=torch.float32, device=gy.device)
    rows = N * P * Q
    nb = lib.ctgan_colsum_workspace_bytes(rows, K)
    ws = workspace(nb, gy.device)
    check(lib.ctgan_colsum(_ptr(gy), rows, K, K, _ptr(out), _ptr(ws), ws.numel(), _stream()), 'colsum')
    return out


# ------------------------------------------------------------------------------- elementwise
def _ew_out(x):
    if not is_dense(x):
        raise ValueError('elementwise kernels need dense tensors')
    return empty_like_dense(x)


def lrelu_fwd(x, alpha):
    _need_dev(x)
    y = _ew_out(x)
    check(lib.ctgan_lrelu_fwd(_ptr(x), _ptr(y), x.numel(), alpha, _stream()), 'lrelu_fwd')
    return y


def lrelu_bwd(gy, ref, alpha, scale=1.0):
    _need_dev(gy, ref)
    gy = match_layout(gy, ref)
    gx = _ew_out(ref)
    if scale == 1.0:
        check(lib.ctgan_lrelu_bwd(_ptr(gy), _ptr(ref), _ptr(gx), ref.numel(), alpha, _stream()), 'lrelu_bwd')
    else:
        check(lib.ctgan_lrelu_bwd_scaled(_ptr(gy), _ptr(ref), _ptr(gx), ref.numel(), alpha, scale, _stream()), 'lrelu_bwd_scaled')
    return gx


def dropout(x, u, keep):
    _need_dev(x, u)
    u = match_layout(u, x)
    y = _ew_out(x)
    check(lib.ctgan_dropout(_ptr(x), _ptr(u), _ptr(y), x.numel(), keep, _stream()), 'dropout')
    return y


def dropout_rng(x, keep, seed, stream_id, ctr):
    """dropout whose uniform draw is element i of the Philox stream (seed, stream_id, ctr[0]) at x's PHYSICAL index i."""
    _need_dev(x)
    assert ctr.is_cuda and ctr.dtype == torch.int64
    y = _ew_out(x)
    check(lib.ctgan_dropout_rng(_ptr(x), _ptr(y), x.numel(), keep, seed, stream_id, _ptr(ctr), _stream()), 'dropout_rng')
    return y


def lrelu_dropout_rng(x, ref, alpha, keep, seed, stream_id, ctr, out=None):
    """x * (ref > 0 ? 1 : alpha) / keep * floor(keep + u) in one launch: dropout(LeakyReLU(x)) for ref = x; the backward of that pair for
    x = the gradient, ref = the forward result.  ref in x's physical layout; draws as dropout_rng.  out: result buffer in x's layout."""
    _need_dev(x, ref, out)
    assert ctr.is_cuda and ctr.dtype == torch.int64
    assert tuple(ref.shape) == tuple(x.shape) and ref.stride() == x.stride()
    if out is not None:
        assert is_dense(x) and tuple(out.shape) == tuple(x.shape) and out.stride() == x.stride()
        y = out
    else:
        y = _ew_out(x)
    check(lib.ctgan_lrelu_dropout_rng(_ptr(x), _ptr(ref), _ptr(y), x.numel(), alpha, keep, seed, stream_id, _ptr(ctr), _stream()),
          'lrelu_dropout_rng')
    return y


def lrelu_dropout_rng2(x, ref, n1_rows, alpha, keep, seed, stream_id, stream_id2, ctr):
    """lrelu_dropout_rng over a tensor whose leading n1_rows rows draw stream_id and whose remaining rows draw stream_id2 (each part indexed
    from its own first element: the draws of two separate launches) - one launch."""
    _need_dev(x, ref)
    assert ctr.is_cuda and ctr.dtype == torch.int64 and is_dense(x)
    assert tuple(ref.shape) == tuple(x.shape) and ref.stride() == x.stride() and x.stride(0) == x[0].numel()
    y = _ew_out(x)
    n1 = n1_rows * x[0].numel()
    check(lib.ctgan_lrelu_dropout_rng2(_ptr(x), _ptr(ref), _ptr(y), x.numel(), n1, alpha, keep, seed, stream_id, stream_id2, _ptr(ctr), _stream()),
          'lrelu_dropout_rng2')
    return y


def dropout_rng_mask(x, ref, keep, seed, stream_id, ctr, want_dropped=True):
    """-> (dropout_rng(x, ...) or None, lrelu_bwd(dropout_rng(x, ...), ref, 0)) in one launch; ref in x's physical layout."""
    _need_dev(x, ref)
    assert ctr.is_cuda and ctr.dtype == torch.int64
    assert tuple(ref.shape) == tuple(x.shape) and ref.stride() == x.stride()
    y = _ew_out(x) if want_dropped else None
    ym = _ew_out(x)
    check(lib.ctgan_dropout_rng_mask(_ptr(x), _ptr(ref), _ptr(y), _ptr(ym), x.numel(), keep, seed, stream_id, _ptr(ctr), _stream()),
          'dropout_rng_mask')
    return y, ym


def tanh_fwd(x):
    _need_dev(x)
    y = _ew_out(x)
    check(lib.ctgan_tanh_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), 'tanh_fwd')
    return y


def tanh_bwd(gy, y):
    _need_dev(gy, y)
    gy = match_layout(gy, y)
    gx = _ew_out(y)
    check(lib.ctgan_tanh_bwd(_ptr(gy), _ptr(y), _ptr(gx), y.numel(), _stream()), 'tanh_bwd')
    return gx


def sigmoid_fwd(x):
    _need_dev(x)
    y = _ew_out(x)
    check(lib.ctgan_sigmoid_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), 'sigmoid_fwd')
    return y


def sigmoid_bwd(gy, y):
    _need_dev(gy, y)
    gy = match_layout(gy, y)
    gx = _ew_out(y)
    check(lib.ctgan_sigmoid_bwd(_ptr(gy), _ptr(y), _ptr(gx), y.numel(), _stream()), 'sigmoid_bwd')
    return gx


def axpby(x, y, a, b, out=None):
    """a*x + b*y (y may be None); out: result buffer with x's layout (may be x itself: elementwise)."""
    _need_dev(x, y, out)
    if y is not None:
        y = match_layout(y, x)
    if out is None:
        out = _ew_out(x)
    else:
        assert out.shape == x.shape and out.stride() == x.stride()
    check(lib.ctgan_axpby(_ptr(x), _ptr(y), _ptr(out), x.numel(), a, b, _stream()), 'axpby')
    return out


def copy4d(x, out):
    """out[...] = x[...] for two logical-4D tensors of equal shape and arbitrary strides."""
    _need_dev(x, out)
    assert x.shape == out.shape and x.dim() == 4
    check(lib.ctgan_copy4d(_ptr(x), I64x4(*x.stride()), _ptr(out), I64x4(*out.stride()), I32x4(*x.shape), _stream()),
          'copy4d')
    return out


def to_channels_last(x):
    if x.permute(0, 2, 3, 1).is_contiguous():
        return x
    return copy4d(x, empty_cl(*x.shape, device=x.device))


def to_nchw(x):
    if x.is_contiguous():
        return x
    return copy4d(x, torch.empty(x.shape, dtype=x.dtype, device=x.device))


def match_layout(t, ref):
    """Return t with exactly ref's strides (repacking through copy4d if needed)."""
    if t.shape != ref.shape:
        raise ValueError('shape mismatch %s vs %s' % (tuple(t.shape), tuple(ref.shape)))
    if t.stride() == ref.stride():
        return t
    if t.dim() != 4:
        raise ValueError('layout mismatch on a non-4D tensor')
    return copy4d(t, empty_like_dense(ref))


def pool2(x, scale):
    """y[n,c,p,q] = scale * (sum of the 2x2 window of x)."""
    _need_dev(x)
    N, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    y = empty_cl(N, C, H // 2, W // 2, x.device)
    check(lib.ctgan_pool2(_ptr(x), I64x4(*x.stride()), _ptr(y), I64x4(*y.stride()), I32x4(*y.shape), scale, _stream()),
          'pool2')
    return y


def upsample2(x, scale):
    """y[n,c,h,w] = scale * x[n,c,h//2,w//2]."""
    _need_dev(x)
    N, C, H, W = x.shape
    y = empty_cl(N, C, 2 * H, 2 * W, x.device)
    check(lib.ctgan_upsample2(_ptr(x), I64x4(*x.stride()), _ptr(y), I64x4(*y.stride()), I32x4(*y.shape), scale,
                              _stream()), 'upsample2')
    return y


FILTER_ROTATE, FILTER_PHASES, FILTER_SPREAD, FILTER_SPREAD_FLIP = 0, 1, 2, 3


def filter_job_shape(kind, R, S, C, Ko):
    """Shape of the tensor a filter job writes (the dgrad layouts are opaque 1-D buffers)."""
    if kind == FILTER_ROTATE:
        return (R * S * C * Ko,)
    if kind == FILTER_PHASES:
        return (4 * ((R + 1) // 2) * ((S + 1) // 2) * C * Ko,)
    return (R + 1, S + 1, Ko, C) if kind == FILTER_SPREAD_FLIP else (R + 1, S + 1, C, Ko)


def filter_batch(jobs):
    """jobs: list of (src [R,S,C,K], dst, kind, pad_t, pad_l, scale[, pre, pre_scale]) - every derived filter in one launch
    per 40.  pre = FILTER_SPREAD / FILTER_SPREAD_FLIP: `kind` (ROTATE / PHASES) is taken of that spread of src (scale
    pre_scale) without materialising it: dst has the shape the layout of the spread buffer would have."""
    from ._lib import FilterJob
    if not jobs:
        return
    arr = (FilterJob * len(jobs))()
    for i, job in enumerate(jobs):
        src, dst, kind, pad_t, pad_l, scale = job[:6]
        pre, pre_scale = (job[6], job[7]) if len(job) > 6 else (0, 1.0)
        _need_dev(src, dst)
        R, S, C, Ko = src.shape
        if pre:
            eff = (R + 1, S + 1, Ko, C) if pre == FILTER_SPREAD_FLIP else (R + 1, S + 1, C, Ko)
        else:
            eff = (R, S, C, Ko)
        assert src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape) == filter_job_shape(kind, *eff)
        arr[i] = FilterJob(src.data_ptr(), dst.data_ptr(), R, S, C, Ko, kind, pad_t, pad_l, scale, pre, pre_scale)
    check(lib.ctgan_filter_batch(arr, len(jobs), _stream()), 'filter_batch')


def dgrad_filter_kind(g):
    """Which layout conv_dgrad wants pre-repacked for geometry g (mirror of dgrad_phase_mode in csrc/igemm.hip)."""
    phase = g.stride == 2 and g.H % 2 == 0 and g.W % 2 == 0 and g.C % 4 == 0 and g.K % 32 == 0 and g.P * 2 == g.H and g.Q * 2 == g.W
    return FILTER_PHASES if phase else FILTER_ROTATE


def filter_spread(w, scale, flip):
    """[R,S,C,K] -> scale * (sum of the four one-tap shifts) as [(R+1),(S+1),C,K]; flip: rotated and I/O swapped
    [(R+1),(S+1),K,C] (the conv2d_transpose filter of UpsampleConv)."""
    _need_dev(w)
    R, S, C, Ko = w.shape
    assert w.is_contiguous()
    out = torch.empty((R + 1, S + 1, Ko, C) if flip else (R + 1, S + 1, C, Ko), dtype=torch.float32, device=w.device)
    check(lib.ctgan_filter_spread(_ptr(w), _ptr(out), R, S, C, Ko, scale, 1 if flip else 0, _stream()), 'filter_spread')
    return out


def filter_fold(w4, scale, flip, out=None):
    """Adjoint of filter_spread: [(R+1),(S+1),C,K] (flip: [(R+1),(S+1),K,C]) -> [R,S,C,K]."""
    _need_dev(w4)
    assert w4.is_contiguous()
    R, S = w4.shape[0] - 1, w4.shape[1] - 1
    C, Ko = (w4.shape[3], w4.shape[2]) if flip else (w4.shape[2], w4.shape[3])
    if out is None:
        out = torch.empty((R, S, C, Ko), dtype=torch.float32, device=w4.device)
    assert tuple(out.shape) == (R, S, C, Ko) and out.is_contiguous()
    check(lib.ctgan_filter_fold(_ptr(w4), _ptr(out), R, S, C, Ko, scale, 1 if flip else 0, _stream()), 'filter_fold')
    return out


def filter_fold_batch(jobs):
    """jobs = [(w4, scale, flip, out), ...]: every fold in one launch (csrc: ctgan_filter_fold_batch)."""
    from ._lib import FoldJob
    if not jobs:
        return
    arr = (FoldJob * len(jobs))()
    for i, (w4, scale, flip, out) in enumerate(jobs):
        _need_dev(w4, out)
        assert w4.is_contiguous() and out.is_contiguous()
        R, S = w4.shape[0] - 1, w4.shape[1] - 1
        C, Ko = (w4.shape[3], w4.shape[2]) if flip else (w4.shape[2], w4.shape[3])
        assert tuple(out.shape) == (R, S, C, Ko)
        arr[i] = FoldJob(w4.data_ptr(), out.data_ptr(), R, S, C, Ko, scale, 1 if flip else 0)
    check(lib.ctgan_filter_fold_batch(arr, len(jobs), _stream()), 'filter_fold_batch')


def mul(x, y):
    """Elementwise product of two dense tensors of the same shape and layout."""
    _need_dev(x, y)
    y = match_layout(y, x) if x.dim() == 4 else y
    assert x.shape == y.shape and x.stride() == y.stride() and is_dense(x)
    out = _ew_out(x)
    check(lib.ctgan_mul(_ptr(x), _ptr(y), _ptr(out), x.numel(), _stream()), 'mul')
    return out


def rsqrt(x, eps):
    _need_dev(x)
    assert is_dense(x)
    y = _ew_out(x)
    check(lib.ctgan_rsqrt(_ptr(x), _ptr(y), x.numel(), eps, _stream()), 'rsqrt')
    return y


def sample_sum(x, scale):
    """[N, ...] dense -> [N]: scale * sum over everything but the batch axis."""
    _need_dev(x)
    assert is_dense(x) and x.stride(0) == x[0].numel()
    y = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(lib.ctgan_sample_sum(_ptr(x), _ptr(y), x.shape[0], x[0].numel(), scale, _stream()), 'sample_sum')
    return y


def sample_bcast(v, like, scale):
    """[N] -> tensor with the shape and layout of `like` (dense, batch-major): out[n, ...] = scale * v[n]."""
    _need_dev(v, like)
    assert is_dense(like) and like.stride(0) == like[0].numel() and v.is_contiguous() and v.numel() == like.shape[0]
    y = _ew_out(like)
    check(lib.ctgan_sample_bcast(_ptr(v), _ptr(y), like.shape[0], like[0].numel(), scale, _stream()), 'sample_bcast')
    return y


def channel_affine(x, scale, offset=None):
    """x [N,C,H,W] channels-last (or [N,C]): x * scale[c] + offset[c]."""
    _need_dev(x, scale, offset)
    C = x.shape[1]
    assert (x.dim() == 2 and x.is_contiguous()) or (x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous())
    y = _ew_out(x)
    check(lib.ctgan_channel_affine(_ptr(x), _ptr(scale), _ptr(offset), _ptr(y), x.numel() // C, C, _stream()), 'channel_affine')
    return y


# ---- fused Layernorm (csrc/layernorm.hip) ------------------------------------------------------------------------------------
def _ln_dims(x):
    """(N, D, C) of a dense channel-fastest tensor ([N,C,H,W] channels-last or [N,C] contiguous), or None."""
    if x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous():
        return x.shape[0], x.shape[1] * x.shape[2] * x.shape[3], x.shape[1]
    if x.dim() == 2 and x.is_contiguous():
        return x.shape[0], x.shape[1], x.shape[1]
    return None


def layernorm_supported(x):
    d = _ln_dims(x)
    return d is not None and bool(lib.ctgan_layernorm_supported(d[1], d[2]))


def _ln_ws(N, D, C, dev):
    return workspace(lib.ctgan_layernorm_workspace_bytes(N, D, C), dev)


def layernorm_fwd(x, scale, offset, eps, relu=False):
    """-> (y, mean [N], rstd [N])   (TF/tflib/ops/layernorm.py:6-20); relu: y = relu(Layernorm(x)) in the same pass"""
    _need_dev(x, scale, offset)
    N, D, C = _ln_dims(x)
    y = _ew_out(x)
    mean = torch.empty(N, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = _ln_ws(N, D, C, x.device)
    check(lib.ctgan_layernorm_fwd(_ptr(x), _ptr(scale), _ptr(offset), _ptr(y), _ptr(mean), _ptr(rstd), N, D, C, float(eps), 1 if relu else 0, _ptr(ws),
                                  ws.numel(), _stream()), 'layernorm_fwd')
    return y, mean, rstd


def layernorm_bwd(gy, x, scale, mean, rstd, want_params, ymask=None):
    """-> (gx, gscale | None, goffset | None); ymask = the fused-ReLU forward result (gy counts where it is positive)"""
    _need_dev(gy, x, scale, mean, rstd, ymask)
    N, D, C = _ln_dims(x)
    gy = match_layout(gy, x)
    gx = _ew_out(x)
    gs = torch.empty(C, dtype=torch.float32, device=x.device) if want_params else None
    go = torch.empty(C, dtype=torch.float32, device=x.device) if want_params else None
    ws = _ln_ws(N, D, C, x.device)
    check(lib.ctgan_layernorm_bwd(_ptr(gy), _ptr(x), _ptr(scale), _ptr(mean), _ptr(rstd), _ptr(ymask), _ptr(gx), _ptr(gs), _ptr(go), N, D, C, _ptr(ws), ws.numel(),
                                  _stream()), 'layernorm_bwd')
    return gx, gs, go


def layernorm_bwd2(u, gy, x, scale, mean, rstd, want_gy=True, want_x=True, want_scale=True, ymask=None):
    """Adjoint of layernorm_bwd: cotangent u of gx -> (cot_gy, cot_x, cot_scale), None where not wanted."""
    _need_dev(u, gy, x, scale, mean, rstd, ymask)
    N, D, C = _ln_dims(x)
    u, gy = match_layout(u, x), match_layout(gy, x)
    cg = _ew_out(x) if want_gy else None
    cx = _ew_out(x) if want_x else None
    cs = torch.empty(C, dtype=torch.float32, device=x.device) if want_scale else None
    ws = _ln_ws(N, D, C, x.device)
    check(lib.ctgan_layernorm_bwd2(_ptr(u), _ptr(gy), _ptr(x), _ptr(scale), _ptr(mean), _ptr(rstd), _ptr(ymask), _ptr(cg), _ptr(cx), _ptr(cs), N, D, C, _ptr(ws),
                                   ws.numel(), _stream()), 'layernorm_bwd2')
    return cg, cx, cs


def spatial_sum(x, scale):
    """[N,C,H,W] channels-last -> [N,C], scale * sum over (h,w)."""
    _need_dev(x)
    x = to_channels_last(x)
    N, C, H, W = x.shape
    y = torch.empty((N, C), dtype=torch.float32, device=x.device)
    check(lib.ctgan_spatial_sum(_ptr(x), _ptr(y), N, H * W, C, scale, _stream()), 'spatial_sum')
    return y


def spatial_bcast(g, H, W, scale):
    """[N,C] -> channels-last [N,C,H,W] = scale * g[n,c]."""
    _need_dev(g)
    g = g.contiguous()
    N, C = g.shape
    y = empty_cl(N, C, H, W, g.device)
    check(lib.ctgan_spatial_bcast(_ptr(g), _ptr(y), N, H * W, C, scale, _stream()), 'spatial_bcast')
    return y


def real_prep(x_int, noise, denom):
    _need_dev(x_int, noise)
    assert x_int.dtype == torch.int32 and x_int.is_contiguous()
    y = torch.empty(x_int.shape, dtype=torch.float32, device=x_int.device)
    if noise is not None:
        assert noise.is_contiguous() and noise.shape == x_int.shape
    check(lib.ctgan_real_prep(_ptr(x_int), _ptr(noise), _ptr(y), x_int.numel(), denom, _stream()), 'real_prep')
    return y


def interpolate(real, fake, alpha):
    _need_dev(real, fake, alpha)
    B, D = real.shape
    assert real.is_contiguous() and fake.is_contiguous() and fake.shape == real.shape and alpha.numel() == B
    out = torch.empty_like(real)
    check(lib.ctgan_interpolate(_ptr(real), _ptr(fake), _ptr(alpha.contiguous()), _ptr(out), B, D, _stream()),
          'interpolate')
    return out


# ------------------------------------------------------------------------------- batch norm
def bn_stats(x4, groups, eps=1e-5):
    """Training-mode moments of a channels-last [N,C,H,W] tensor over (n,h,w) per statistic group -> mean[groups,C], rstd[groups,C]
    (the first two launches of bn_fwd)."""
    _need_dev(x4)
    N, C, H, W = x4.shape
    assert x4.permute(0, 2, 3, 1).is_contiguous()
    mean = torch.empty((groups, C), dtype=torch.float32, device=x4.device)
    rstd = torch.empty((groups, C), dtype=torch.float32, device=x4.device)
    nb = lib.ctgan_bn_workspace_bytes(N, H * W, C, groups, 1)
    ws = workspace(nb, x4.device)
    check(lib.ctgan_bn_stats(_ptr(x4), N, H * W, C, groups, eps, _ptr(mean), _ptr(rstd), _ptr(ws), ws.numel(), _stream()), 'bn_stats')
    return mean, rstd


def conv_fwd_bn_in_supported(x, g, tanh=False, labels=None, resid=None):
    """Does conv_fwd_bn_in have a kernel for this launch?  Asked BEFORE the moments are computed, so that a caller who falls back to the
    separate batch norm does not pay a bn_stats launch it then repeats."""
    if fewch_handles(g):
        return labels is None and resid is None
    if tanh:
        return False
    if MMA_DTYPE == 'f32x3':
        return True
    if MMA_DTYPE is None and X3_HYBRID:
        y_strides = (g.K * g.P * g.Q, 1, g.Q * g.K, g.K)          # channels-last result (empty_cl)
        d = g.desc(x.shape[0], x.stride(), y_strides)
        return bool(lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), 0))
    return False


def conv_fwd_bn_in(x, w, bias, g, mean, rstd, scale, offset, groups, relu_in=True, tanh=False, out_strides=None, labels=None, resid=None,
                   resid_up=False):
    """conv(relu?(bn(x))) [+ bias] [+ resid (through a nearest-2x upsample: resid_up)] [tanh] with the training-mode batch norm applied while
    the conv stages its input (ctgan_epilogue_ext.in_bn_*): the many -> few pixel kernel (3x3, <= 4 output channels, 32-pixel rows; no labels,
    no resid) or, in the split mode, the fragment-streaming halo kernel (stride 1, tiles inside one image, relu_in) - NotImplementedError
    elsewhere.  mean / rstd [groups, C] from bn_stats; scale / offset [C], or [n_labels, C] tables with labels (int32 per sample)."""
    from ._lib import EpilogueExt
    _need_dev(x, w, bias, mean, rstd, scale, offset, labels, resid)
    N = x.shape[0]
    assert tuple(x.shape) == (N, g.C, g.H, g.W) and tuple(w.shape) == (g.R, g.S, g.C, g.K) and w.is_contiguous() and not g.x_up
    assert tuple(mean.shape) == (groups, g.C) and tuple(rstd.shape) == (groups, g.C) and scale.shape[-1] == g.C and offset.shape[-1] == g.C
    assert labels is not None or scale.numel() == g.C
    if out_strides is None:
        y = empty_cl(N, g.K, g.P, g.Q, x.device)
    else:
        y = torch.empty_strided((N, g.K, g.P, g.Q), out_strides, dtype=torch.float32, device=x.device)
    d = g.desc(N, x.stride(), y.stride())
    e = EpilogueExt(0.0, 0, 0, None)
    e.in_bn_mean, e.in_bn_rstd, e.in_bn_scale, e.in_bn_offset = mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), offset.data_ptr()
    e.in_bn_groups, e.out_tanh = int(groups), 1 if tanh else 0
    e.in_bn_labels = labels.data_ptr() if labels is not None else None
    if fewch_handles(g):
        if labels is not None or resid is not None:
            raise NotImplementedError('conv_fwd_bn_in: the many -> few kernel takes neither labels nor a residual')
        _timed(g, N, lambda: check(lib.ctgan_conv2d_fwd_ex(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), None, _ptr(y), 2 if relu_in else 0, ctypes.byref(e), _stream()), 'conv2d_fwd'))
        return y
    if tanh or not (MMA_DTYPE == 'f32x3' or (MMA_DTYPE is None and X3_HYBRID and lib.ctgan_conv2d16_x3_prefers(ctypes.byref(d), 0))):
        raise NotImplementedError('conv_fwd_bn_in: batch norm on load exists in the split mode\'s halo kernel and in the many -> few pixel kernel')
    if resid is not None and resid_up:
        assert tuple(resid.shape) == (N, g.K, g.P // 2, g.Q // 2) and resid.permute(0, 2, 3, 1).is_contiguous()
    elif resid is not None:
        assert is_dense_like(resid, y)
    wp = _packed16(w, d, 0, g, 'f32x3')
    fl = (2 if relu_in else 0) | (8 if (resid_up and resid is not None) else 0)
    _timed(g, N, lambda: check(lib.ctgan_conv2d16_fwd_ex(ctypes.byref(d), 3, _ptr(x), _ptr(wp), _ptr(bias), _ptr(resid), _ptr(y), fl, ctypes.byref(e), None, 0, _stream()),
                               'conv2d16_fwd_ex'))
    return y


def bn_fwd(x, scale, offset, labels, groups, relu, eps=1e-5):
    """x channels-last [N,C,H,W] (or [N,C]); returns y, mean[groups,C], rstd[groups,C]."""
    _need_dev(x, scale, offset, labels)
    x4 = x if x.dim() == 4 else x.view(x.shape[0], x.shape[1], 1, 1)
    x4 = to_channels_last(x4)
    N, C, H, W = x4.shape
    hw = H * W
    mean = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    rstd = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    nb = lib.ctgan_bn_workspace_bytes(N, hw, C, groups, 1)
    ws = workspace(nb, x.device)
    check(lib.ctgan_bn_stats(_ptr(x4), N, hw, C, groups, eps, _ptr(mean), _ptr(rstd), _ptr(ws), ws.numel(), _stream()),
          'bn_stats')
    y = empty_cl(N, C, H, W, x.device)
    check(lib.ctgan_bn_apply(_ptr(x4), _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(offset), _ptr(labels), _ptr(y), N, hw,
                             C, groups, 1 if relu else 0, _stream()), 'bn_apply')
    if x.dim() == 2:
        y = y.view(N, C)
    return y, mean, rstd, x4


def bn_bwd(gy, x4, mean, rstd, scale, offset, labels, groups, relu):
    _need_dev(gy, x4, mean, rstd, scale, offset, labels)
    N, C, H, W = x4.shape
    hw = H * W
    gy4 = gy if gy.dim() == 4 else gy.reshape(N, C, 1, 1)
    gy4 = to_channels_last(gy4)
    n_labels = scale.shape[0] if labels is not None else 1
    gx = empty_cl(N, C, H, W, gy.device)
    gscale = torch.empty((n_labels, C), dtype=torch.float32, device=gy.device)
    goffset = torch.empty((n_labels, C), dtype=torch.float32, device=gy.device)
    nb = lib.ctgan_bn_workspace_bytes(N, hw, C, groups, n_labels)
    ws = workspace(nb, gy.device)
    check(lib.ctgan_bn_bwd(_ptr(gy4), _ptr(x4), _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(offset), _ptr(labels),
                           _ptr(gx), _ptr(gscale), _ptr(goffset), N, hw, C, groups, n_labels, 1 if relu else 0,
                           _ptr(ws), ws.numel(), _stream()), 'bn_bwd')
    return gx, gscale, goffset


# ------------------------------------------------------------------------------- loss heads
def gp_fwd(g, lam, defer_mean=False):
    """-> (gp, slopes).  defer_mean: only the slopes are computed here; gp is an UNWRITTEN slot that tail_critic_heads_fwd(slopes=...,
    gp=...) fills with lam * mean((slopes - 1)^2) in its one-workgroup kernel before using it (one launch per critic step less)."""
    _need_dev(g)
    g = g.contiguous()
    B, D = g.shape
    slopes = torch.empty(B, dtype=torch.float32, device=g.device)
    gp = torch.empty((), dtype=torch.float32, device=g.device)
    check(lib.ctgan_gp_fwd(_ptr(g), B, D, lam, _ptr(slopes), None if defer_mean else _ptr(gp), _stream()), 'gp_fwd')
    return gp, slopes


def gp_bwd(g, slopes, gout, lam):
    _need_dev(g, slopes, gout)
    g = g.contiguous()
    B, D = g.shape
    gg = torch.empty_like(g)
    check(lib.ctgan_gp_bwd(_ptr(g), _ptr(slopes), _ptr(gout.contiguous()), B, D, lam, _ptr(gg), _stream()), 'gp_bwd')
    return gg


def gp_bwd_mean(g, slopes, gout, lam, out5=None):
    """gp_bwd + the penalty's value in the same launch: -> (gg, gp); with out5 (the five sums of tail_critic_heads_fwd evaluated WITHOUT
    the penalty) gp is also added into out5[0] (cost) and out5[4] (wgan + ct + gp)."""
    _need_dev(g, slopes, gout, out5)
    g = g.contiguous()
    B, D = g.shape
    gg = torch.empty_like(g)
    gp = torch.empty((), dtype=torch.float32, device=g.device)
    assert out5 is None or (out5.is_contiguous() and out5.numel() == 5)
    check(lib.ctgan_gp_bwd_mean(_ptr(g), _ptr(slopes), _ptr(gout.contiguous()), B, D, lam, _ptr(gg), _ptr(gp), _ptr(out5), _stream()), 'gp_bwd_mean')
    return gg, gp


def ct_fwd(d, d_, f, f_, lam2, M):
    _need_dev(d, d_, f, f_)
    d, d_, f, f_ = d.contiguous(), d_.contiguous(), f.contiguous(), f_.contiguous()
    B, NF = f.shape
    ct_i = torch.empty(B, dtype=torch.float32, device=d.device)
    ct = torch.empty((), dtype=torch.float32, device=d.device)
    check(lib.ctgan_ct_fwd(_ptr(d), _ptr(d_), _ptr(f), _ptr(f_), B, NF, lam2, M, _ptr(ct_i), _ptr(ct), _stream()),
          'ct_fwd')
    return ct, ct_i


def ct_bwd(d, d_, f, f_, ct_i, gout, lam2, M):
    _need_dev(d, d_, f, f_, ct_i, gout)
    d, d_, f, f_ = d.contiguous(), d_.contiguous(), f.contiguous(), f_.contiguous()
    B, NF = f.shape
    gd, gd_ = torch.empty_like(d), torch.empty_like(d_)
    gf, gf_ = torch.empty_like(f), torch.empty_like(f_)
    check(lib.ctgan_ct_bwd(_ptr(d), _ptr(d_), _ptr(f), _ptr(f_), _ptr(ct_i), _ptr(gout.contiguous()), B, NF, lam2, M,
                           _ptr(gd), _ptr(gd_), _ptr(gf), _ptr(gf_), _stream()), 'ct_bwd')
    return gd, gd_, gf, gf_


def softmax_ce_fwd(logits, labels):
    _need_dev(logits, labels)
    logits = logits.contiguous()
    B, NC = logits.shape
    assert labels.dtype == torch.int32 and labels.numel() == B
    probs = torch.empty_like(logits)
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    ncorrect = torch.empty((), dtype=torch.float32, device=logits.device)
    check(lib.ctgan_softmax_ce_fwd(_ptr(logits), _ptr(labels.contiguous()), B, NC, _ptr(probs), _ptr(loss),
                                   _ptr(ncorrect), _stream()), 'softmax_ce_fwd')
    return loss, probs, ncorrect


def softmax_ce_bwd(probs, labels, gout):
    _need_dev(probs, labels, gout)
    B, NC = probs.shape
    gl = torch.empty_like(probs)
    check(lib.ctgan_softmax_ce_bwd(_ptr(probs), _ptr(labels.contiguous()), _ptr(gout.contiguous()), B, NC, _ptr(gl),
                                   _stream()), 'softmax_ce_bwd')
    return gl


def mean_diff_fwd(x, na, nb, sa, sb):
    _need_dev(x)
    x = x.contiguous()
    assert x.numel() == na + nb
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(lib.ctgan_mean_diff_fwd(_ptr(x), na, nb, sa, sb, _ptr(out), _stream()), 'mean_diff_fwd')
    return out


def mean_diff_bwd(gout, na, nb, sa, sb):
    _need_dev(gout)
    gx = torch.empty(na + nb, dtype=torch.float32, device=gout.device)
    check(lib.ctgan_mean_diff_bwd(_ptr(gout.contiguous()), na, nb, sa, sb, _ptr(gx), _stream()), 'mean_diff_bwd')
    return gx


# ------------------------------------------------------------------------------- optimizer / rng
def adam_step(theta, g, m, v, state, beta1, beta2, eps=1e-8, grad_scale=1.0):
    """In-place TF-Adam on flat fp32 buffers; `state` = device float[4] {lr, beta1^t, beta2^t, -}."""
    _need_dev(theta, g, m, v, state)
    for t in (theta, g, m, v):
        assert t.is_contiguous() and t.numel() == theta.numel()
    check(lib.ctgan_adam_step(_ptr(theta), _ptr(g), _ptr(m), _ptr(v), theta.numel(), _ptr(state), beta1, beta2, eps,
                              grad_scale, _stream()), 'adam_step')


def step_advance(state, beta1, beta2, rng_ctr=None, rng_by=1):
    """adam_advance(state) and rng_advance(rng_ctr, rng_by) in one launch - the end of a step."""
    _need_dev(state)
    assert rng_ctr is None or (rng_ctr.is_cuda and rng_ctr.dtype == torch.int64)
    check(lib.ctgan_step_advance(_ptr(state), beta1, beta2, _ptr(rng_ctr), rng_by, _stream()), 'step_advance')


ADAM_PACKED_MAX = 64


def adam_step_packed(srcs, dst_offs, counts, flat, theta, m, v, state, beta1, beta2, eps=1e-8, grad_scale=1.0):
    """pack(srcs -> flat) + adam_step in one launch (len(srcs) <= ADAM_PACKED_MAX)."""
    _need_dev(flat, theta, m, v, state, *[t for t in srcs if t is not None])
    n = len(srcs)
    assert n <= ADAM_PACKED_MAX
    for t, c in zip(srcs, counts):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == c)
    P = (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in srcs])
    O = (ctypes.c_int64 * n)(*dst_offs)
    C = (ctypes.c_int64 * n)(*counts)
    check(lib.ctgan_adam_step_packed(P, O, C, n, _ptr(flat), _ptr(theta), _ptr(m), _ptr(v), _ptr(state), beta1, beta2, eps, grad_scale,
                                     _stream()), 'adam_step_packed')


def pack(srcs, dst_offs, counts, flat):
    """flat[off_i : off_i+n_i] = srcs[i] (None = zeros) for all i in one launch (per 64 tensors)."""
    _need_dev(flat, *[t for t in srcs if t is not None])
    n = len(srcs)
    P = (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in srcs])
    O = (ctypes.c_int64 * n)(*dst_offs)
    C = (ctypes.c_int64 * n)(*counts)
    check(lib.ctgan_pack(P, O, C, n, _ptr(flat), _stream()), 'pack')


def critic_heads_fwd(d, f, a, labels, B, lam2, M, scale, gp=None):
    """-> (out[5] = cost (incl. gp), wgan, ct, acgan, wgan + ct + gp; ct_i [B]; probs [B,ncls] or None)."""
    _need_dev(d, f, a, labels, gp)
    assert d.is_contiguous() and f.is_contiguous() and d.numel() == 3 * B and f.shape[0] == 3 * B
    assert a is None or (a.is_contiguous() and a.shape[0] == 3 * B)
    out = torch.empty(5, dtype=torch.float32, device=d.device)
    ct_i = torch.empty(B, dtype=torch.float32, device=d.device)
    ncls = a.shape[1] if a is not None else 0
    probs = torch.empty(B, ncls, dtype=torch.float32, device=d.device) if a is not None else None
    check(lib.ctgan_critic_heads_fwd(_ptr(d), _ptr(f), _ptr(a), _ptr(labels), _ptr(gp), B, f.shape[1], ncls, lam2, M, scale, _ptr(ct_i),
                                     _ptr(probs), _ptr(out), _stream()), 'critic_heads_fwd')
    return out, ct_i, probs


def critic_heads_bwd(d, f, probs, labels, ct_i, gout, B, lam2, M, scale):
    _need_dev(d, f, probs, labels, ct_i, gout)
    assert gout.is_contiguous() and gout.numel() in (1, 4)
    gd = torch.empty_like(d); gf = torch.empty_like(f)
    ncls = probs.shape[1] if probs is not None else 0
    ga = torch.empty(3 * B, ncls, dtype=torch.float32, device=d.device) if probs is not None else None
    check(lib.ctgan_critic_heads_bwd(_ptr(d), _ptr(f), _ptr(probs), _ptr(labels), _ptr(ct_i), _ptr(gout), gout.numel(), B, f.shape[1], ncls, lam2, M,
                                     scale, _ptr(gd), _ptr(gf), _ptr(ga), _stream()), 'critic_heads_bwd')
    return gd, gf, ga


def _cl_rows(y):
    """(n, hw, nf) of a dense channels-last 4-D tensor [n,nf,H,W] (the layout the fused head kernels read)."""
    assert y.dim() == 4 and y.permute(0, 2, 3, 1).is_contiguous(), 'fused heads want a dense channels-last tensor'
    return y.shape[0], y.shape[2] * y.shape[3], y.shape[1]


def tail_heads_fwd(y, w_out, b_out, w_ac, b_ac, relu=False):
    """f [n,nf] = mean_hw y (of relu(y) if relu), d [n] = f.w_out + b_out, a [n,ncls] = f.w_ac + b_ac (either head may be
    None) - one launch."""
    _need_dev(y, w_out, b_out, w_ac, b_ac)
    n, hw, nf = _cl_rows(y)
    f = torch.empty(n, nf, dtype=torch.float32, device=y.device)
    d = torch.empty(n, dtype=torch.float32, device=y.device) if w_out is not None else None
    ncls = w_ac.shape[1] if w_ac is not None else 0
    a = torch.empty(n, ncls, dtype=torch.float32, device=y.device) if w_ac is not None else None
    assert w_out is None or (w_out.is_contiguous() and w_out.numel() == nf)
    assert w_ac is None or (w_ac.is_contiguous() and w_ac.shape[0] == nf)
    check(lib.ctgan_tail_heads_fwd(_ptr(y), n, hw, nf, 1 if relu else 0, _ptr(w_out), _ptr(b_out), _ptr(w_ac), _ptr(b_ac), ncls, _ptr(f),
                                   _ptr(d), _ptr(a), _stream()), 'tail_heads_fwd')
    return f, d, a


def tail_critic_heads_fwd(y, B, w_out, b_out, w_ac, b_ac, labels, gp, lam2, M, scale, slopes=None, gp_lambda=0.0, y_clean=None,
                          clean_relu=False):
    """tail_heads_fwd over the 3B rows of a critic step + critic_heads_fwd, two launches.
    -> (out[5], f [3B,nf], d [3B], a [3B,ncls] or None, ct_i [B], probs [B,ncls] or None, acc [2] or None).
    slopes (with gp = the slot of gp_fwd(defer_mean=True)): gp is computed from the slopes and written in place first.
    y_clean [2B,nf,H,W] (dense channels-last): its class-head logits and the two accuracies ride the same two launches."""
    _need_dev(y, w_out, b_out, w_ac, b_ac, labels, gp, slopes, y_clean)
    n, hw, nf = _cl_rows(y)
    assert n == 3 * B and w_out.is_contiguous() and w_out.numel() == nf
    dev = y.device
    if slopes is not None or y_clean is not None:
        assert slopes is None or (gp is not None and slopes.is_contiguous() and slopes.numel() == B)
        ncls = w_ac.shape[1] if w_ac is not None else 0
        f_c = a_c = acc = None
        if y_clean is not None:
            n2, hw2, nf2 = _cl_rows(y_clean)
            assert (n2, hw2, nf2) == (2 * B, hw, nf) and w_ac is not None
            f_c = torch.empty(n2, nf, dtype=torch.float32, device=dev)
            a_c = torch.empty(n2, ncls, dtype=torch.float32, device=dev)
            acc = torch.empty(2, dtype=torch.float32, device=dev)
        f = torch.empty(n, nf, dtype=torch.float32, device=dev)
        d = torch.empty(n, dtype=torch.float32, device=dev)
        a = torch.empty(n, ncls, dtype=torch.float32, device=dev) if w_ac is not None else None
        probs = torch.empty(B, ncls, dtype=torch.float32, device=dev) if w_ac is not None else None
        ce_i = torch.empty(B, dtype=torch.float32, device=dev) if w_ac is not None else None
        ct_i = torch.empty(B, dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        check(lib.ctgan_tail_critic_heads_fwd2(_ptr(y), B, hw, nf, _ptr(w_out), _ptr(b_out), _ptr(w_ac), _ptr(b_ac), ncls, _ptr(labels), _ptr(gp),
                                               _ptr(slopes), gp_lambda, _ptr(y_clean), 1 if clean_relu else 0, _ptr(f_c), _ptr(a_c), _ptr(acc),
                                               lam2, M, scale, _ptr(f), _ptr(d), _ptr(a), _ptr(ct_i), _ptr(probs), _ptr(ce_i), _ptr(out),
                                               _stream()), 'tail_critic_heads_fwd2')
        return out, f, d, a, ct_i, probs, acc
    f = torch.empty(n, nf, dtype=torch.float32, device=dev)
    d = torch.empty(n, dtype=torch.float32, device=dev)
    ncls = w_ac.shape[1] if w_ac is not None else 0
    a = torch.empty(n, ncls, dtype=torch.float32, device=dev) if w_ac is not None else None
    probs = torch.empty(B, ncls, dtype=torch.float32, device=dev) if w_ac is not None else None
    ce_i = torch.empty(B, dtype=torch.float32, device=dev) if w_ac is not None else None
    ct_i = torch.empty(B, dtype=torch.float32, device=dev)
    out = torch.empty(5, dtype=torch.float32, device=dev)
    check(lib.ctgan_tail_critic_heads_fwd(_ptr(y), B, hw, nf, _ptr(w_out), _ptr(b_out), _ptr(w_ac), _ptr(b_ac), ncls, _ptr(labels), _ptr(gp),
                                          lam2, M, scale, _ptr(f), _ptr(d), _ptr(a), _ptr(ct_i), _ptr(probs), _ptr(ce_i), _ptr(out),
                                          _stream()), 'tail_critic_heads_fwd')
    return out, f, d, a, ct_i, probs, None


def tail_heads_bwd(y, d, f, probs, labels, ct_i, gout, B, lam2, M, scale, mask_scale, w_out, w_ac, out=None, y_gp=None, out_gp=None):
    """-> (gy like y: gradient w.r.t. the last block's pre-activation; gw_out, gb_out, gw_ac, gb_ac) - one launch.
    out: buffer for gy (y's shape, dense channels-last - e.g. the leading rows of a larger batch).
    y_gp, out_gp: rows of the gradient-penalty pass - the same launch also writes out_gp = gp_head_grad(y_gp, w_out, mask_scale)."""
    _need_dev(y, d, f, probs, labels, ct_i, gout, w_out, w_ac, out, y_gp, out_gp)
    n, hw, nf = _cl_rows(y)
    assert n == 3 * B and gout.is_contiguous() and gout.numel() in (1, 4)
    if out is not None:
        assert tuple(out.shape) == tuple(y.shape) and _cl_rows(out) == (n, hw, nf)
        gy = out
    else:
        gy = empty_cl(n, nf, y.shape[2], y.shape[3], y.device)
    gw_out = torch.empty_like(w_out); gb_out = torch.empty(1, dtype=torch.float32, device=y.device)
    ncls = w_ac.shape[1] if w_ac is not None else 0
    gw_ac = torch.empty_like(w_ac) if w_ac is not None else None
    gb_ac = torch.empty(ncls, dtype=torch.float32, device=y.device) if w_ac is not None else None
    n_gp = 0
    if y_gp is not None:
        n_gp = y_gp.shape[0]
        assert out_gp is not None and tuple(out_gp.shape) == tuple(y_gp.shape) and _cl_rows(y_gp) == (n_gp, hw, nf) and _cl_rows(out_gp) == (n_gp, hw, nf)
        assert w_out.is_contiguous()
    check(lib.ctgan_tail_heads_bwd_gp(_ptr(y), _ptr(d), _ptr(f), _ptr(probs), _ptr(labels), _ptr(ct_i), _ptr(gout), gout.numel(), B, hw, nf, ncls,
                                      lam2, M, scale, mask_scale, _ptr(w_out), _ptr(w_ac), _ptr(gy), _ptr(gw_out), _ptr(gb_out), _ptr(gw_ac),
                                      _ptr(gb_ac), _ptr(y_gp), n_gp, _ptr(out_gp), _stream()), 'tail_heads_bwd')
    return gy, gw_out, gb_out, gw_ac, gb_ac


def gen_heads_fwd(y, w_out, b_out, w_ac, b_ac, labels, ac_scale):
    """-> (cost [1] = -mean(d) + ac_scale * CE(a, labels), probs [n,ncls] or None, d [n]) from the last critic block's output."""
    _need_dev(y, w_out, b_out, w_ac, b_ac, labels)
    n, hw, nf = _cl_rows(y)
    dev = y.device
    f = torch.empty(n, nf, dtype=torch.float32, device=dev); d = torch.empty(n, dtype=torch.float32, device=dev)
    ncls = w_ac.shape[1] if w_ac is not None else 0
    a = torch.empty(n, ncls, dtype=torch.float32, device=dev) if w_ac is not None else None
    probs = torch.empty(n, ncls, dtype=torch.float32, device=dev) if w_ac is not None else None
    out = torch.empty(1, dtype=torch.float32, device=dev)
    check(lib.ctgan_gen_heads_fwd(_ptr(y), n, hw, nf, _ptr(w_out), _ptr(b_out), _ptr(w_ac), _ptr(b_ac), ncls, _ptr(labels), ac_scale, _ptr(f),
                                  _ptr(d), _ptr(a), _ptr(probs), _ptr(out), _stream()), 'gen_heads_fwd')
    return out, probs, d


def gen_heads_bwd(y, probs, labels, gout, ac_scale, mask_scale, w_out, w_ac):
    """gradient of that cost w.r.t. the last critic conv's result (mask and 1/keep included), one launch."""
    _need_dev(y, probs, labels, gout, w_out, w_ac)
    n, hw, nf = _cl_rows(y)
    gy = empty_cl(n, nf, y.shape[2], y.shape[3], y.device)
    ncls = w_ac.shape[1] if w_ac is not None else 0
    check(lib.ctgan_gen_heads_bwd(_ptr(y), _ptr(probs), _ptr(labels), _ptr(gout), n, hw, nf, ncls, ac_scale, mask_scale, _ptr(w_out), _ptr(w_ac),
                                  _ptr(gy), _stream()), 'gen_heads_bwd')
    return gy


def gp_head_grad(y, w_out, mask_scale, out=None):
    """gz = (y > 0) * w_out / hw * mask_scale  (dD/dz of D = mean_hw(relu(dropout(z))) . w_out); out: buffer for gz (as tail_heads_bwd)."""
    _need_dev(y, w_out, out)
    n, hw, nf = _cl_rows(y)
    assert w_out.is_contiguous() and w_out.numel() == nf
    if out is not None:
        assert tuple(out.shape) == tuple(y.shape) and _cl_rows(out) == (n, hw, nf)
        gz = out
    else:
        gz = empty_cl(n, nf, y.shape[2], y.shape[3], y.device)
    check(lib.ctgan_gp_head_grad(_ptr(y), _ptr(w_out), n, hw, nf, mask_scale, _ptr(gz), _stream()), 'gp_head_grad')
    return gz


def gp_head_wgrad(gg, y, mask_scale, like, add_to=None):
    """gw_out = mask_scale / hw * sum over (row, hw) with y > 0 of gg  (adjoint of gp_head_grad w.r.t. w_out).
    add_to: a finished gradient of the same weight - the result is ADDED onto it in place (and it is returned)."""
    _need_dev(gg, y, add_to)
    n, hw, nf = _cl_rows(y)
    assert tuple(gg.shape) == tuple(y.shape) and gg.permute(0, 2, 3, 1).is_contiguous()
    ws = torch.empty(64 * nf, dtype=torch.float32, device=y.device)
    if add_to is not None:
        assert add_to.is_contiguous() and add_to.numel() == nf
        check(lib.ctgan_gp_head_wgrad_acc(_ptr(gg), _ptr(y), n, hw, nf, mask_scale, _ptr(add_to), _ptr(ws), _stream()), 'gp_head_wgrad_acc')
        return add_to
    gw = torch.empty_like(like)
    check(lib.ctgan_gp_head_wgrad(_ptr(gg), _ptr(y), n, hw, nf, mask_scale, _ptr(gw), _ptr(ws), _stream()), 'gp_head_wgrad')
    return gw


def gp_finish(ga, gs, scale):
    """ga [B,C,H,W] (plain NCHW, contiguous) += scale * upsample2(gs [B,C,H/2,W/2], any strides), in place; -> slopes [B] = per-sample L2
    norm of the result.  The end of the penalty's first backward: gradient through the first conv + through the pooled shortcut."""
    _need_dev(ga, gs)
    B, C, H, W = ga.shape
    assert ga.is_contiguous() and tuple(gs.shape) == (B, C, H // 2, W // 2) and H % 2 == 0 and W % 2 == 0
    slopes = torch.empty(B, dtype=torch.float32, device=ga.device)
    check(lib.ctgan_gp_finish(_ptr(ga), _ptr(gs), I64x4(*gs.stride()), B, C, H, W, scale, _ptr(slopes), _stream()), 'gp_finish')
    return slopes


def accuracy2(logits, labels, B):
    _need_dev(logits, labels)
    assert logits.is_contiguous() and logits.shape[0] == 2 * B
    acc = torch.empty(2, dtype=torch.float32, device=logits.device)
    check(lib.ctgan_accuracy2(_ptr(logits), _ptr(labels), B, logits.shape[1], _ptr(acc), _stream()), 'accuracy2')
    return acc


def adam_advance(state, beta1, beta2):
    _need_dev(state)
    check(lib.ctgan_adam_advance(_ptr(state), beta1, beta2, _stream()), 'adam_advance')


def rng_uniform(out, seed, stream_id, ctr, lo=0.0, hi=1.0):
    _need_dev(out)
    assert is_dense(out) and out.dtype == torch.float32
    check(lib.ctgan_rng_uniform(_ptr(out), out.numel(), seed, stream_id, _ptr(ctr), lo, hi, _stream()), 'rng_uniform')
    return out


def rng_normal(out, seed, stream_id, ctr):
    _need_dev(out)
    assert is_dense(out) and out.dtype == torch.float32
    check(lib.ctgan_rng_normal(_ptr(out), out.numel(), seed, stream_id, _ptr(ctr), _stream()), 'rng_normal')
    return out


def rng_labels(out, nlab, seed, stream_id, ctr):
    assert out.is_cuda and out.dtype == torch.int32 and out.is_contiguous()
    check(lib.ctgan_rng_labels(_ptr(out), out.numel(), nlab, seed, stream_id, _ptr(ctr), _stream()), 'rng_labels')
    return out


def critic_prep(x_int, fake, seed, sid_deq, sid_alpha, ctr, lo, hi, denom):
    """-> (rf [2B,d] = [real ; fake], interp [B,d], both) in one launch: dequantised reals, x_hat and the batch of the two dropout
    passes, and `both` [3B,d] = the buffer they are adjacent views of; the draws are those of rng_uniform on [B,d] (stream sid_deq) and
    [B,1] (stream sid_alpha)."""
    _need_dev(x_int, fake)
    B, d = x_int.shape
    assert x_int.dtype == torch.int32 and x_int.is_contiguous() and fake.is_contiguous() and tuple(fake.shape) == (B, d)
    assert ctr.is_cuda and ctr.dtype == torch.int64
    both = torch.empty(3 * B, d, dtype=torch.float32, device=x_int.device)      # one buffer: [real ; fake ; x_hat] is also a batch
    rf, interp = both[:2 * B], both[2 * B:]
    check(lib.ctgan_critic_prep(_ptr(x_int), _ptr(fake), B, d, seed, sid_deq, sid_alpha, _ptr(ctr), lo, hi, denom, _ptr(rf), _ptr(interp),
                                _stream()), 'critic_prep')
    return rf, interp, both


def rows_cat_dropout(x, n_extra, keep, seed, stream_id, ctr):
    """dropout([x ; x[:n_extra]], keep) along dim 0 in one launch (x dense, rows contiguous); keep = 1: plain concat."""
    _need_dev(x)
    assert is_dense(x) and ctr.is_cuda and ctr.dtype == torch.int64
    n = x.shape[0]
    row = x.numel() // n
    if x.dim() == 4 and not x.is_contiguous():
        out = empty_cl(n + n_extra, x.shape[1], x.shape[2], x.shape[3], x.device)
        assert x.permute(0, 2, 3, 1).is_contiguous()
    else:
        assert x.is_contiguous()
        out = torch.empty((n + n_extra,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    check(lib.ctgan_rows_cat_dropout(_ptr(x), n, n_extra, row, keep, seed, stream_id, _ptr(ctr), _ptr(out), _stream()), 'rows_cat_dropout')
    return out


def rows_gather_dropout(src, segs, seed, ctr):
    """dst = concatenation of row segments of src (dense, rows contiguous), each with its own dropout: segs = [(src_row0, rows,
    keep, stream_id, index_row0), ...] (csrc: ctgan_rows_gather_dropout).  One launch."""
    from ._lib import RowSegment
    _need_dev(src)
    assert is_dense(src) and ctr.is_cuda and ctr.dtype == torch.int64 and 1 <= len(segs) <= 6
    n_dst = sum(sg[1] for sg in segs)
    row = src.numel() // src.shape[0]
    if src.dim() == 4 and not src.is_contiguous():
        assert src.permute(0, 2, 3, 1).is_contiguous()
        out = empty_cl(n_dst, src.shape[1], src.shape[2], src.shape[3], src.device)
    else:
        assert src.is_contiguous()
        out = torch.empty((n_dst,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    arr = (RowSegment * len(segs))()
    for i, (r0, rows, keep, sid, idx0) in enumerate(segs):
        assert 0 <= r0 and r0 + rows <= src.shape[0]
        arr[i] = RowSegment(r0, rows, keep, sid, idx0)
    check(lib.ctgan_rows_gather_dropout(_ptr(src), arr, len(segs), row, seed, _ptr(ctr), _ptr(out), _stream()), 'rows_gather_dropout')
    return out


def rows_cat_bwd(g, n_src, n_extra, n_pass=0):
    """adjoint of [x ; x[:n_extra]]: g[:n_src] with g[n_src:n_src + n_extra] added onto its first n_extra rows; n_pass further rows of g
    behind the concat pass straight through (result: n_src + n_pass rows)."""
    _need_dev(g)
    assert is_dense(g) and g.shape[0] == n_src + n_extra + n_pass
    row = g.numel() // g.shape[0]
    if g.dim() == 4 and not g.is_contiguous():
        assert g.permute(0, 2, 3, 1).is_contiguous()
        out = empty_cl(n_src + n_pass, g.shape[1], g.shape[2], g.shape[3], g.device)
    else:
        assert g.is_contiguous()
        out = torch.empty((n_src + n_pass,) + tuple(g.shape[1:]), dtype=torch.float32, device=g.device)
    if n_pass:
        check(lib.ctgan_rows_cat_bwd2(_ptr(g), n_src, n_extra, n_pass, row, _ptr(out), _stream()), 'rows_cat_bwd2')
    else:
        check(lib.ctgan_rows_cat_bwd(_ptr(g), n_src, n_extra, row, _ptr(out), _stream()), 'rows_cat_bwd')
    return out


def rng_advance(ctr, by=1):
    assert ctr.is_cuda and ctr.dtype == torch.int64
    check(lib.ctgan_rng_advance(_ptr(ctr), by, _stream()), 'rng_advance')
