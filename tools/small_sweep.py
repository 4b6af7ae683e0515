#!/usr/bin/env python
"""fwd tile config sweep on the small-M conv shapes (env CTGAN_FWD_CFG).  usage: CTGAN_FWD_CFG=k python tools/small_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
row = [os.environ.get('CTGAN_FWD_CFG', 'auto')]
for (N, H) in [(64, 8), (128, 8), (192, 8), (64, 16), (128, 16), (320, 16), (64, 4), (128, 4)]:
    g = K.ConvGeom(128, H, H, 128, 3, 3, 1, False)
    x = K.empty_cl(N, 128, H, H, 'cuda').normal_(); w = torch.randn(3, 3, 128, 128, device='cuda') * 0.05
    t = timeit(lambda: K.conv_fwd(x, w, None, g, relu_in=True))
    row.append('%dx%d^2: %5.1fus %5.1fTF %s' % (N, H, t, 2.0 * N * H * H * 128 * 1152 / t / 1e6, K.last_kernel().split('<')[1].split('>')[0]))
print(' | '.join(row))
