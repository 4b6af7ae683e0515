"""The C-ABI library loads and exports every symbol include/ctgan_hip.h declares (no compute calls,
no GPU needed); the product never imports the oracle; the ctypes table matches the header."""
import pytest
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(header='ctgan_hip.h'):
    src = open(os.path.join(ROOT, 'include', header)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ctgan_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from ctgan_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build the extension first: python __graft_entry__.py'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 40
    missing = [n for n in names + _header_functions('ctgan_hip_debug.h') if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.ctgan_version() == 1


def test_ctypes_table_matches_header():
    from ctgan_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_functions()
    assert sorted(_lib.DEBUG_SIGNATURES) == _header_functions('ctgan_hip_debug.h')


def test_product_header_carries_no_debug_switches():
    """The drop-in ABI (include/ctgan_hip.h) declares operators only; the test / A-B switches live in ctgan_hip_debug.h."""
    assert not [n for n in _header_functions() if 'debug' in n]
    assert all(n.startswith('ctgan_debug_') for n in _header_functions('ctgan_hip_debug.h'))


def test_conv_desc_layout_matches_header():
    """struct ctgan_conv_desc: 14 int32 then 2 x int64[4] (8-byte aligned) = 120 bytes."""
    from ctgan_amd._lib import ConvDesc
    assert ctypes.sizeof(ConvDesc) == 14 * 4 + 2 * 32
    assert ConvDesc.xs.offset == 56 and ConvDesc.ys.offset == 88


def test_error_reporting_without_gpu():
    """Argument validation happens before any launch, so it is observable on a CPU-only box."""
    from ctgan_amd import _lib
    rc = _lib.lib.ctgan_dropout(None, None, None, 4, 0.0, None)
    assert rc == -1 and b'keep' in _lib.lib.ctgan_last_error()
    rc = _lib.lib.ctgan_conv2d_fwd(None, None, None, None, None, None, 0, None)
    assert rc == -1


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'ctgan_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'cpu_kernels' in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_kernels_refuse_cpu_tensors():
    import pytest
    import torch
    import ctgan_amd.kernels as K
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        K.lrelu_fwd(torch.zeros(4), 0.0)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        K.conv_fwd(torch.zeros(1, 4, 2, 2), torch.zeros(1, 1, 4, 4), None, K.ConvGeom(4, 2, 2, 4, 1, 1))


def test_reference_error_behaviour_is_kept(cpu_kernels):
    """Same exceptions / messages as the reference operators for unsupported configurations
    (TF/tflib/ops/deconv2d.py:38-39, cond_batchnorm.py:8-9, linear.py:102-104, CT_gan_cifar_resnet.py:125-126)."""
    import pytest
    import torch
    import ctgan_amd.gan_cifar_resnet as R
    from ctgan_amd.tflib.ops import cond_batchnorm, deconv2d, linear
    x = torch.zeros(1, 4, 2, 2)
    with pytest.raises(Exception, match='Unsupported configuration'):
        deconv2d.Deconv2D('d', 4, 4, 5, x, mask_type=('a', 1))
    with pytest.raises(Exception, match='unsupported'):
        cond_batchnorm.Batchnorm('b', [0, 1], x, labels=None, n_labels=10)
    with pytest.raises(Exception, match='Invalid initialization!'):
        linear.Linear('l', 4, 4, torch.zeros(1, 4), initialization='nope')
    with pytest.raises(Exception, match='invalid resample value'):
        R.ResidualBlock('r', 4, 4, 3, x, resample='sideways')


def test_shared_library_was_built_from_the_sources_in_this_tree():
    """Provenance: __graft_entry__.build() stamps the library with the sha256 of the sources it compiled; a stale prebuilt
    library (sources edited, library not rebuilt) fails here and shows as `build.matches_tree: false` in the bench line."""
    import json
    import os
    import __graft_entry__ as ge
    stamp_path = os.path.join(ge.ROOT, 'ctgan_amd', 'libctgan_hip.build.json')
    if not os.path.exists(stamp_path):
        import pytest
        pytest.skip('library not built through __graft_entry__.build()')
    stamp = json.load(open(stamp_path))
    assert stamp['sources_sha256'] == ge.source_digest(), 'libctgan_hip.so is older than the kernel sources: run python __graft_entry__.py'


@pytest.mark.timeout(60)
def test_multi_segment_weight_gradient_planner_terminates_for_small_segments():
    """Host-side planners only (no launch, no GPU): the split-K plan of a filter used by several SMALL segments - more segments than
    planned splits - must come back.  ctgan_conv2d_wgrad_group_workspace_bytes looped forever on the 128x128 ResNet's 8x8 1x1 shortcut
    at B = 4 (three segments of 8 / 4 / 4 samples, one planned split): every segment needs a split of its own."""
    import ctgan_amd.kernels as K
    from ctgan_amd._lib import WgradGroup, lib
    for C, H, Ko, k, st, Ns in [(64, 8, 128, 1, 1, (8, 4, 4)), (64, 8, 128, 1, 1, (1, 1, 1)), (128, 4, 128, 3, 1, (2, 2, 2)),
                                (128, 8, 128, 3, 1, (192, 64)), (32, 2, 64, 3, 1, (1, 1))]:
        g = K.ConvGeom(C, H, H, Ko, k, k, st, False)
        arr = (WgradGroup * 2)()
        for G in arr:
            G.d = g.desc(Ns[0], (C * H * H, 1, H * C, C), (Ko * g.P * g.Q, 1, g.Q * Ko, Ko))
            G.nseg = len(Ns)
            for i, n in enumerate(Ns):
                G.Ns[i] = n
        one = lib.ctgan_conv2d_wgrad_group_workspace_bytes(arr, 1)
        two = lib.ctgan_conv2d_wgrad_group_workspace_bytes(arr, 2)
        slab = 4 * (k * k * C + 1) * Ko
        assert one >= len(Ns) * slab and one % 256 == 0 and two == 2 * one, (C, H, Ns, one, two)
        ns = (ctypes.c_int32 * len(Ns))(*Ns)
        assert lib.ctgan_conv2d_wgrad_multi_workspace_bytes(ctypes.byref(arr[0].d), len(Ns), ns) >= len(Ns) * slab
