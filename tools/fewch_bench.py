#!/usr/bin/env python
"""Micro-benchmark of the few-channel direct kernels (csrc/fewch.hip) on the CT-WGAN shapes, with the HBM floor
(bytes of the wide tensor / 4 TB/s) beside each time.   usage: python tools/fewch_bench.py   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K


def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if len(sys.argv) > 1 and sys.argv[1] == 'ring':
    K.debug_m2f_px(False)
for (N, C, H, Ko, k) in [(64, 3, 32, 128, 3), (128, 3, 32, 128, 3), (192, 3, 32, 128, 3), (64, 128, 32, 3, 3), (128, 128, 32, 3, 3), (320, 128, 32, 3, 3), (64, 3, 16, 128, 1), (128, 3, 16, 128, 1)]:
    g = K.ConvGeom(C, H, H, Ko, k, k, 1, False)
    few_in = C <= 4
    x = (torch.randn(N, C, H, H, device='cuda') if few_in else K.empty_cl(N, C, H, H, 'cuda').normal_())
    gy = (K.empty_cl(N, Ko, H, H, 'cuda').normal_() if few_in else torch.randn(N, Ko, H, H, device='cuda'))
    w = torch.randn(k, k, C, Ko, device='cuda') * 0.05
    b = torch.randn(Ko, device='cuda')
    wide_mb = N * H * H * max(C, Ko) * 4 / 1e6
    t_f = timeit(lambda: K.conv_fwd(x, w, b, g)); n_f = K.last_symbol() or K.last_kernel()
    t_d = timeit(lambda: K.conv_dgrad(gy, w, g, N, out_strides=tuple(x.stride()) if few_in else None)); n_d = K.last_symbol() or K.last_kernel()
    t_w = timeit(lambda: K.conv_wgrad(x, gy, g, with_bias=True)); n_w = K.last_kernel()
    print('%-26s wide %5.1f MB floor(8 TB/s) %5.1f us | fwd %6.1f us %-24s | dgrad %6.1f us %-24s | wgrad %6.1f us %s' % (
        (N, C, H, Ko, k), wide_mb, wide_mb / 8.0, t_f, n_f, t_d, n_d, t_w, n_w))
