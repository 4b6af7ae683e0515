#!/usr/bin/env python
"""A/B of the four-phase stride-2 data gradient: conv16x3p_kernel (four phases from one dy patch) against conv16x3sf_kernel (one phase per workgroup,
slice staging, filter fragments from L2; ctgan_debug_x3_s2dgrad_sf) - agreement and time per launch.  usage: python tools/sf_dgrad_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K


def timed(fn, reps=40):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (N, C, H, Ko) in [(192, 128, 32, 128), (128, 128, 32, 128), (320, 128, 32, 128), (320, 128, 16, 128), (192, 128, 16, 256), (64, 128, 32, 128)]:
    g = torch.Generator().manual_seed(N + H)
    geom = K.ConvGeom(C, H, H, Ko, 4, 4, 2, False)
    w = (torch.randn(4, 4, C, Ko, generator=g) / (16 * Ko) ** 0.5).cuda()
    K._STABLE_PTRS.add(w.data_ptr())
    gy = K.empty_cl(N, Ko, geom.P, geom.Q, 'cuda').copy_(torch.randn(N, Ko, geom.P, geom.Q, generator=g).cuda())
    bc = torch.randn(C, generator=g).cuda()
    m = K.empty_cl(N, C, H, H, 'cuda').normal_(); rr = K.empty_cl(N, C, H, H, 'cuda').normal_()
    out = {}
    with K.mma_dtype('f32x3'):
        for sw, code in ((0, -1), (1, 0)):
            K.lib.ctgan_debug_x3_s2dgrad_sf(code)
            a = K.conv_dgrad(gy, w, geom, N); ka = K.last_kernel()
            b = K.conv_dgrad(gy, w, geom, N, bias=bc, mask=m, resid=rr)
            t = timed(lambda: K.conv_dgrad(gy, w, geom, N))
            out[sw] = (a, b, ka, t)
    K.lib.ctgan_debug_x3_s2dgrad_sf(0)
    da = float((out[0][0] - out[1][0]).abs().max() / out[0][0].abs().max())
    db = float((out[0][1] - out[1][1]).abs().max() / out[0][1].abs().max())
    fl = 2.0 * N * geom.P * geom.Q * Ko * 16 * C
    print('%-24s %-26s %6.1f us %5.0f TF | %-26s %6.1f us %5.0f TF | max diff %.2e / %.2e' % (
        (N, C, H, Ko), out[0][2], out[0][3], fl / out[0][3] / 1e6, out[1][2], out[1][3], fl / out[1][3] / 1e6, da, db))
