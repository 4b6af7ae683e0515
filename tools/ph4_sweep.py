#!/usr/bin/env python
"""Tile-config sweep (env CTGAN_FWD_CFG) on the 4-phase stride-2 data gradient and the 4x4 stride-2 forward conv of the
resampled layers.   usage: CTGAN_FWD_CFG=k python tools/ph4_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
row = [os.environ.get('CTGAN_FWD_CFG', 'auto')]
for (N, H) in [(64, 32), (128, 32), (320, 32), (64, 16), (128, 16)]:
    g = K.ConvGeom(128, H, H, 128, 4, 4, 2, False)
    x = K.empty_cl(N, 128, H, H, 'cuda').normal_(); w = torch.randn(4, 4, 128, 128, device='cuda') * 0.05
    gy = K.empty_cl(N, 128, H // 2, H // 2, 'cuda').normal_()
    wt = K.repack_filter(w, g)
    fl = 2.0 * N * (H // 2) ** 2 * 128 * 16 * 128
    t = timeit(lambda: K.conv_dgrad(gy, w, g, N, wt=wt)); kd = K.last_kernel().split('<')[1].split('>')[0]
    t2 = timeit(lambda: K.conv_fwd(x, w, None, g)); kf = K.last_kernel().split('<')[1].split('>')[0]
    row.append('%dx%d^2 dgrad %6.1fus %5.1fTF %s | fwd %6.1fus %5.1fTF %s' % (N, H, t, fl / t / 1e6, kd, t2, fl / t2 / 1e6, kf))
print('\n   '.join(row))
