#!/usr/bin/env python
"""Sweep the pipelined-conv tile configurations over the small/medium shapes (run per config via env)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
SHAPES = [(192, 8), (128, 8), (64, 8), (256, 8), (128, 4), (64, 4), (128, 16), (64, 16), (192, 16), (32, 16), (64, 32), (192, 32)]
res = []
for N, H in SHAPES:
    g = K.ConvGeom(128, H, H, 128, 3, 3, 1, False)
    x = K.empty_cl(N, 128, H, H, 'cuda').normal_(); w = torch.randn(3, 3, 128, 128, device='cuda') * 0.05
    K.conv_fwd(x, w, None, g); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): K.conv_fwd(x, w, None, g)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e-3
    res.append('%dx%d:%.1fus/%.0fTF' % (N, H, t * 1e6, 2.0 * N * H * H * 128 * 1152 / t / 1e12))
print(os.environ.get('CTGAN_FWD_CFG', 'auto'), K.last_kernel(), ' '.join(res))
