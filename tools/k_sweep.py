#!/usr/bin/env python
"""Where does igemm_fwd_pipe<128x128> lose time: per tile (prologue/epilogue) or per K slice?
usage: python tools/k_sweep.py   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

print('dbg', os.environ.get('CTGAN_DBG', '0'))
for N in (32, 64, 128, 256):
    for C in (128, 256, 512):
        for H in (32,):
            g = K.ConvGeom(C, H, H, 128, 3, 3, 1, False)
            x = K.empty_cl(N, C, H, H, 'cuda').normal_()
            w = torch.randn(3, 3, C, 128, device='cuda') * 0.05
            fl = 2.0 * N * H * H * 128 * 9 * C
            t = timeit(lambda: K.conv_fwd(x, w, None, g))
            print('N %4d C %4d H %3d tiles %5d iters %4d  %8.1f us %6.1f TF  %s' % (N, C, H, N * H * H // 128, 9 * C // 32, t * 1e6, fl / t / 1e12, K.last_kernel()[:40]))
