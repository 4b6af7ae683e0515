"""ctypes binding of libctgan_hip.so (the C-ABI declared in include/ctgan_hip.h).

The product path has no CPU fallback: if the shared library is missing or an entry point is
absent this module raises at import, and every wrapper in `kernels.py` raises on non-HIP tensors.
Build with `python __graft_entry__.py` (or `make -C ctgan_amd/csrc`).
"""
import ctypes
import os

# torch must be imported BEFORE the library is loaded: torch ships its own HIP runtime and places it
# in the global symbol scope; loading ours first would bind the kernels to the system runtime and
# leave the process with two HIP runtimes (streams / allocations of one are invalid in the other).
import torch  # noqa: F401  (also the device-memory / stream provider of every wrapper)
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CTGAN_LIB') or os.path.join(_HERE, 'libctgan_hip.so')      # CTGAN_LIB: A/B builds (tools only)


class ConvDesc(ctypes.Structure):
    """struct ctgan_conv_desc (include/ctgan_hip.h)."""
    _fields_ = [
        ('N', c_int32), ('C', c_int32), ('H', c_int32), ('W', c_int32),
        ('K', c_int32), ('R', c_int32), ('S', c_int32),
        ('P', c_int32), ('Q', c_int32),
        ('stride', c_int32), ('pad_t', c_int32), ('pad_l', c_int32),
        ('x_up', c_int32), ('reserved', c_int32),
        ('xs', c_int64 * 4), ('ys', c_int64 * 4),
    ]


class FilterJob(ctypes.Structure):
    """struct ctgan_filter_job (include/ctgan_hip.h)."""
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p),
                ('R', c_int32), ('S', c_int32), ('C', c_int32), ('K', c_int32),
                ('kind', c_int32), ('pad_t', c_int32), ('pad_l', c_int32), ('scale', ctypes.c_float),
                ('pre', c_int32), ('pre_scale', ctypes.c_float)]


class WgradGroup(ctypes.Structure):
    """struct ctgan_wgrad_group (include/ctgan_hip.h)."""
    _fields_ = [('d', ConvDesc), ('nseg', c_int32), ('Ns', c_int32 * 3), ('seg_flags', c_int32 * 3),
                ('xs', ctypes.c_void_p * 3), ('dys', ctypes.c_void_p * 3), ('dw', ctypes.c_void_p), ('db', ctypes.c_void_p),
                ('add_dw', ctypes.c_void_p), ('add_db', ctypes.c_void_p)]


class ChainStep(ctypes.Structure):
    """struct ctgan_chain_step (include/ctgan_hip.h)."""
    _fields_ = [('wp', ctypes.c_void_p), ('mask', ctypes.c_void_p), ('post_mask', ctypes.c_void_p), ('out', ctypes.c_void_p),
                ('resid', c_int32), ('save', c_int32), ('drop', c_int32), ('reserved', c_int32)]


class ChainDrop(ctypes.Structure):
    """struct ctgan_chain_drop."""
    _fields_ = [('keep', ctypes.c_float), ('n_split', c_int32), ('stream_id_lo', ctypes.c_uint64), ('stream_id_hi', ctypes.c_uint64)]


class Chain8x8(ctypes.Structure):
    """struct ctgan_chain8x8."""
    _fields_ = [('x', ctypes.c_void_p), ('n_images', c_int32), ('channels', c_int32), ('height', c_int32), ('width', c_int32),
                ('n_convs', c_int32), ('reserved', c_int32), ('step', ChainStep * 5), ('drop', ChainDrop * 2),
                ('drop_seed', ctypes.c_uint64), ('drop_ctr', ctypes.c_void_p)]


class RowSegment(ctypes.Structure):
    """struct ctgan_row_segment (include/ctgan_hip.h)."""
    _fields_ = [('src_row0', c_int64), ('rows', c_int64), ('keep', ctypes.c_float), ('stream_id', ctypes.c_uint64), ('index_row0', c_int64)]


class FoldJob(ctypes.Structure):
    """struct ctgan_fold_job (include/ctgan_hip.h)."""
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('R', c_int32), ('S', c_int32), ('C', c_int32), ('K', c_int32),
                ('scale', ctypes.c_float), ('flip', c_int32)]


class EpilogueExt(ctypes.Structure):
    """struct ctgan_epilogue_ext (include/ctgan_hip.h)."""
    _fields_ = [('drop_keep', ctypes.c_float), ('drop_seed', ctypes.c_uint64), ('drop_stream_id', ctypes.c_uint64),
                ('drop_ctr', ctypes.c_void_p), ('n_ranges', c_int32), ('range_end', c_int32 * 3), ('range_keep', ctypes.c_float * 3),
                ('range_stream_id', ctypes.c_uint64 * 3), ('out_mask', ctypes.c_void_p),
                ('act', c_int32), ('act_alpha', ctypes.c_float), ('act_ref', ctypes.c_void_p),
                ('in_bn_mean', ctypes.c_void_p), ('in_bn_rstd', ctypes.c_void_p), ('in_bn_scale', ctypes.c_void_p), ('in_bn_offset', ctypes.c_void_p),
                ('in_bn_groups', c_int32), ('out_tanh', c_int32), ('in_bn_labels', ctypes.c_void_p)]


I64x4 = c_int64 * 4
I32x4 = c_int32 * 4
_p = c_void_p          # device pointers and the stream travel as void*
_D = POINTER(ConvDesc)

# test / A-B switches and launch introspection (include/ctgan_hip_debug.h): exported by the same library, NOT part of the drop-in ABI
DEBUG_SIGNATURES = {
    'ctgan_debug_force_generic': (None, [c_int]),
    'ctgan_debug_reduce_lanes': (None, [c_int]),
    'ctgan_debug_x3_halo_version': (None, [c_int]),
    'ctgan_debug_x3_s2halo': (None, [c_int]),
    'ctgan_debug_x3_s2fwd': (None, [c_int]),
    'ctgan_debug_x3_s2dgrad_sf': (None, [c_int]),
    'ctgan_debug_m2f_px': (None, [c_int]),
    'ctgan_debug_last_wgrad_group_kinds': (c_int, []),
    'ctgan_debug_last_wgrad_group_col_mask': (ctypes.c_uint, []),
    'ctgan_debug_x3_hk': (None, [c_int, c_int]),
    'ctgan_debug_clock_probe': (c_int, [c_void_p, ctypes.c_uint64, c_void_p, c_void_p]),
}

# name -> (restype, argtypes); the single source the symbol-export test checks against include/ctgan_hip.h
SIGNATURES = {
    'ctgan_version': (c_int, []),
    'ctgan_last_error': (c_char_p, []),
    'ctgan_last_kernel': (c_char_p, []),
    'ctgan_last_symbol': (c_char_p, []),
    'ctgan_conv2d16_chain8x8': (c_int, [POINTER(Chain8x8), _p]),
    'ctgan_conv2d_wgrad_multi_workspace_bytes': (c_size_t, [POINTER(ConvDesc), c_int32, POINTER(c_int32)]),
    'ctgan_conv2d_wgrad_multi': (c_int, [POINTER(ConvDesc), c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), _p, _p, _p,
                                         c_size_t, _p]),
    'ctgan_conv2d_wgrad_group_workspace_bytes': (c_size_t, [POINTER(WgradGroup), c_int32]),
    'ctgan_conv2d_wgrad_group': (c_int, [POINTER(WgradGroup), c_int32, _p, c_size_t, _p]),
    'ctgan_conv2d_wgrad_group_ex': (c_int, [POINTER(WgradGroup), c_int32, _p, c_size_t, c_int, _p]),
    'ctgan_conv2d16_wgrad_group_workspace_bytes': (c_size_t, [POINTER(WgradGroup), c_int32, c_int]),
    'ctgan_conv2d16_wgrad_group': (c_int, [POINTER(WgradGroup), c_int32, c_int, _p, c_size_t, c_int, _p]),
    'ctgan_conv2d_wgrad_group_tile': (c_int, [POINTER(WgradGroup)]),
    'ctgan_conv2d_workspace_bytes': (c_size_t, [_D, c_int]),
    'ctgan_conv2d_fwd': (c_int, [_D, _p, _p, _p, _p, _p, c_int, _p]),
    'ctgan_conv2d_fwd_ex': (c_int, [_D, _p, _p, _p, _p, _p, c_int, POINTER(EpilogueExt), _p]),
    'ctgan_conv2d_dgrad_ex': (c_int, [_D, _p, _p, _p, _p, _p, _p, _p, c_size_t, c_int, POINTER(EpilogueExt), _p]),
    'ctgan_conv2d_dgrad': (c_int, [_D, _p, _p, _p, _p, _p, _p, _p, c_size_t, c_int, _p]),
    'ctgan_conv2d_repack_filter': (c_int, [_D, _p, _p, _p]),
    'ctgan_conv2d_wgrad': (c_int, [_D, _p, _p, _p, _p, _p, c_size_t, c_int, _p]),
    'ctgan_conv2d16_supported': (c_int, [_D, c_int, c_int]),
    'ctgan_conv2d16_x3_prefers': (c_int, [_D, c_int]),
    'ctgan_conv2d16_wgrad_col_takes': (c_int, [_D, c_int, c_int32]),
    'ctgan_conv2d16_filter_elems': (c_size_t, [_D, c_int, c_int]),
    'ctgan_conv2d16_pack_filter': (c_int, [_D, c_int, c_int, _p, _p, _p]),
    'ctgan_conv2d16_pack_batch': (c_int, [POINTER(ConvDesc), POINTER(c_int32), c_int32, c_int, POINTER(c_void_p), POINTER(c_void_p), _p]),
    'ctgan_conv2d16_workspace_bytes': (c_size_t, [_D, c_int]),
    'ctgan_conv2d16_fwd': (c_int, [_D, c_int, _p, _p, _p, _p, _p, c_int, _p, c_size_t, _p]),
    'ctgan_conv2d16_fwd_ex': (c_int, [_D, c_int, _p, _p, _p, _p, _p, c_int, POINTER(EpilogueExt), _p, c_size_t, _p]),
    'ctgan_conv2d16_dgrad': (c_int, [_D, c_int, _p, _p, _p, _p, _p, _p, c_int, _p, c_size_t, _p]),
    'ctgan_conv2d16_dgrad_ex': (c_int, [_D, c_int, _p, _p, _p, _p, _p, _p, c_int, POINTER(EpilogueExt), _p, c_size_t, _p]),
    'ctgan_conv2d16_wgrad_workspace_bytes': (c_size_t, [_D, c_int]),
    'ctgan_conv2d16_wgrad': (c_int, [_D, c_int, _p, _p, _p, _p, c_size_t, c_int, _p]),
    'ctgan_conv2d16_wgrad_bias': (c_int, [_D, c_int, _p, _p, _p, _p, _p, c_size_t, c_int, _p]),
    'ctgan_layernorm_supported': (c_int, [c_int64, c_int32]),
    'ctgan_layernorm_workspace_bytes': (c_size_t, [c_int32, c_int64, c_int32]),
    'ctgan_layernorm_fwd': (c_int, [_p, _p, _p, _p, _p, _p, c_int32, c_int64, c_int32, c_float, c_int32, _p, c_size_t, _p]),
    'ctgan_layernorm_bwd': (c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, c_int32, c_int64, c_int32, _p, c_size_t, _p]),
    'ctgan_layernorm_bwd2': (c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, c_int32, c_int64, c_int32, _p, c_size_t, _p]),
    'ctgan_im2col': (c_int, [_D, _p, c_int32, _p, _p]),
    'ctgan_col2im': (c_int, [_D, _p, c_int32, _p, _p]),
    'ctgan_colsum': (c_int, [_p, c_int64, c_int32, c_int64, _p, _p, c_size_t, _p]),
    'ctgan_colsum_workspace_bytes': (c_size_t, [c_int64, c_int32]),
    'ctgan_lrelu_fwd': (c_int, [_p, _p, c_int64, c_float, _p]),
    'ctgan_lrelu_bwd': (c_int, [_p, _p, _p, c_int64, c_float, _p]),
    'ctgan_lrelu_bwd_scaled': (c_int, [_p, _p, _p, c_int64, c_float, c_float, _p]),
    'ctgan_dropout': (c_int, [_p, _p, _p, c_int64, c_float, _p]),
    'ctgan_tanh_fwd': (c_int, [_p, _p, c_int64, _p]),
    'ctgan_tanh_bwd': (c_int, [_p, _p, _p, c_int64, _p]),
    'ctgan_sigmoid_fwd': (c_int, [_p, _p, c_int64, _p]),
    'ctgan_sigmoid_bwd': (c_int, [_p, _p, _p, c_int64, _p]),
    'ctgan_axpby': (c_int, [_p, _p, _p, c_int64, c_float, c_float, _p]),
    'ctgan_copy4d': (c_int, [_p, POINTER(c_int64), _p, POINTER(c_int64), POINTER(c_int32), _p]),
    'ctgan_pool2': (c_int, [_p, POINTER(c_int64), _p, POINTER(c_int64), POINTER(c_int32), c_float, _p]),
    'ctgan_upsample2': (c_int, [_p, POINTER(c_int64), _p, POINTER(c_int64), POINTER(c_int32), c_float, _p]),
    'ctgan_filter_spread': (c_int, [_p, _p, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, _p]),
    'ctgan_filter_fold': (c_int, [_p, _p, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, _p]),
    'ctgan_filter_batch': (c_int, [_p, c_int32, _p]),
    'ctgan_filter_fold_batch': (c_int, [POINTER(FoldJob), c_int32, _p]),
    'ctgan_mul': (c_int, [_p, _p, _p, c_int64, _p]),
    'ctgan_rsqrt': (c_int, [_p, _p, c_int64, c_float, _p]),
    'ctgan_sample_sum': (c_int, [_p, _p, c_int32, c_int64, c_float, _p]),
    'ctgan_sample_bcast': (c_int, [_p, _p, c_int32, c_int64, c_float, _p]),
    'ctgan_channel_affine': (c_int, [_p, _p, _p, _p, c_int64, c_int32, _p]),
    'ctgan_spatial_sum': (c_int, [_p, _p, c_int32, c_int32, c_int32, c_float, _p]),
    'ctgan_spatial_bcast': (c_int, [_p, _p, c_int32, c_int32, c_int32, c_float, _p]),
    'ctgan_real_prep': (c_int, [_p, _p, _p, c_int64, c_float, _p]),
    'ctgan_interpolate': (c_int, [_p, _p, _p, _p, c_int32, c_int32, _p]),
    'ctgan_bn_stats': (c_int, [_p, c_int32, c_int32, c_int32, c_int32, c_float, _p, _p, _p, c_size_t, _p]),
    'ctgan_bn_apply': (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_int32, _p]),
    'ctgan_bn_bwd': (c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_int32,
                             c_int32, _p, c_size_t, _p]),
    'ctgan_bn_workspace_bytes': (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    'ctgan_gp_fwd': (c_int, [_p, c_int32, c_int32, c_float, _p, _p, _p]),
    'ctgan_gp_bwd': (c_int, [_p, _p, _p, c_int32, c_int32, c_float, _p, _p]),
    'ctgan_ct_fwd': (c_int, [_p, _p, _p, _p, c_int32, c_int32, c_float, c_float, _p, _p, _p]),
    'ctgan_ct_bwd': (c_int, [_p, _p, _p, _p, _p, _p, c_int32, c_int32, c_float, c_float, _p, _p, _p, _p, _p]),
    'ctgan_softmax_ce_fwd': (c_int, [_p, _p, c_int32, c_int32, _p, _p, _p, _p]),
    'ctgan_softmax_ce_bwd': (c_int, [_p, _p, _p, c_int32, c_int32, _p, _p]),
    'ctgan_mean_diff_fwd': (c_int, [_p, c_int32, c_int32, c_float, c_float, _p, _p]),
    'ctgan_mean_diff_bwd': (c_int, [_p, c_int32, c_int32, c_float, c_float, _p, _p]),
    'ctgan_critic_heads_fwd': (c_int, [_p, _p, _p, _p, _p, c_int32, c_int32, c_int32, c_float, c_float, c_float, _p, _p, _p, _p]),
    'ctgan_critic_heads_bwd': (c_int, [_p, _p, _p, _p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, _p, _p, _p, _p]),
    'ctgan_tail_heads_fwd': (c_int, [_p, c_int32, c_int32, c_int32, c_int32, _p, _p, _p, _p, c_int32, _p, _p, _p, _p]),
    'ctgan_tail_critic_heads_fwd': (c_int, [_p, c_int32, c_int32, c_int32, _p, _p, _p, _p, c_int32, _p, _p, c_float, c_float, c_float,
                                            _p, _p, _p, _p, _p, _p, _p, _p]),
    'ctgan_tail_critic_heads_fwd2': (c_int, [_p, c_int32, c_int32, c_int32, _p, _p, _p, _p, c_int32, _p, _p, _p, c_float, _p, c_int32, _p, _p, _p,
                                             c_float, c_float, c_float, _p, _p, _p, _p, _p, _p, _p, _p]),
    'ctgan_tail_heads_bwd': (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_float,
                                     _p, _p, _p, _p, _p, _p, _p, _p]),
    'ctgan_tail_heads_bwd_gp': (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_float,
                                        _p, _p, _p, _p, _p, _p, _p, _p, c_int32, _p, _p]),
    'ctgan_gen_heads_fwd': (c_int, [_p, c_int32, c_int32, c_int32, _p, _p, _p, _p, c_int32, _p, c_float, _p, _p, _p, _p, _p, _p]),
    'ctgan_gen_heads_bwd': (c_int, [_p, _p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, _p, _p, _p, _p]),
    'ctgan_gp_head_grad': (c_int, [_p, _p, c_int32, c_int32, c_int32, c_float, _p, _p]),
    'ctgan_gp_head_wgrad_acc': (c_int, [_p, _p, c_int32, c_int32, c_int32, c_float, _p, _p, _p]),
    'ctgan_gp_finish': (c_int, [_p, _p, POINTER(c_int64), c_int32, c_int32, c_int32, c_int32, c_float, _p, _p]),
    'ctgan_gp_head_wgrad': (c_int, [_p, _p, c_int32, c_int32, c_int32, c_float, _p, _p, _p]),
    'ctgan_accuracy2': (c_int, [_p, _p, c_int32, c_int32, _p, _p]),
    'ctgan_adam_step': (c_int, [_p, _p, _p, _p, c_int64, _p, c_float, c_float, c_float, c_float, _p]),
    'ctgan_adam_advance': (c_int, [_p, c_float, c_float, _p]),
    'ctgan_step_advance': (c_int, [_p, c_float, c_float, _p, c_uint64, _p]),
    'ctgan_adam_step_packed': (c_int, [POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64), c_int32, _p, _p, _p, _p, _p, c_float, c_float,
                                       c_float, c_float, _p]),
    'ctgan_pack': (c_int, [POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64), c_int32, _p, _p]),
    'ctgan_dropout_rng': (c_int, [_p, _p, c_int64, c_float, c_uint64, c_uint64, _p, _p]),
    'ctgan_dropout_rng_mask': (c_int, [_p, _p, _p, _p, c_int64, c_float, c_uint64, c_uint64, _p, _p]),
    'ctgan_lrelu_dropout_rng': (c_int, [_p, _p, _p, c_int64, c_float, c_float, c_uint64, c_uint64, _p, _p]),
    'ctgan_lrelu_dropout_rng2': (c_int, [_p, _p, _p, c_int64, c_int64, c_float, c_float, c_uint64, c_uint64, c_uint64, _p, _p]),
    'ctgan_rng_uniform': (c_int, [_p, c_int64, c_uint64, c_uint64, _p, c_float, c_float, _p]),
    'ctgan_rng_normal': (c_int, [_p, c_int64, c_uint64, c_uint64, _p, _p]),
    'ctgan_rng_labels': (c_int, [_p, c_int64, c_int32, c_uint64, c_uint64, _p, _p]),
    'ctgan_rng_advance': (c_int, [_p, c_uint64, _p]),
    'ctgan_critic_prep': (c_int, [_p, _p, c_int32, c_int32, c_uint64, c_uint64, c_uint64, _p, c_float, c_float, c_float, _p, _p, _p]),
    'ctgan_rows_cat_dropout': (c_int, [_p, c_int64, c_int64, c_int64, c_float, c_uint64, c_uint64, _p, _p, _p]),
    'ctgan_rows_cat_bwd': (c_int, [_p, c_int64, c_int64, c_int64, _p, _p]),
    'ctgan_rows_cat_bwd2': (c_int, [_p, c_int64, c_int64, c_int64, c_int64, _p, _p]),
    'ctgan_gp_bwd_mean': (c_int, [_p, _p, _p, c_int32, c_int32, c_float, _p, _p, _p, _p]),
    'ctgan_rows_gather_dropout': (c_int, [_p, POINTER(RowSegment), c_int32, c_int64, c_uint64, _p, _p, _p]),
}


class CtganError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'ctgan_amd: %s is missing - build the HIP extension first (python __graft_entry__.py, or '
            'make -C ctgan_amd/csrc).  There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError('ctgan_amd: %s does not export %s (stale build?)' % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        msg = lib.ctgan_last_error().decode(errors='replace')
        if rc == -2:
            raise NotImplementedError('%s: %s' % (what, msg))
        if rc == -1:
            raise ValueError('%s: %s' % (what, msg))
        raise CtganError('%s failed (%d): %s' % (what, rc, msg))
