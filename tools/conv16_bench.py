#!/usr/bin/env python
"""Micro-benchmark of the 16-bit conv family on given geometries: fwd / dgrad / wgrad TFLOP/s (20 launches per event bracket).
usage: python tools/conv16_bench.py [bf16|f16]    env CTGAN_DBG16 = diagnosis bits (results are then wrong by design)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
K.X3_HYBRID = False      # families are compared explicitly here: 'f32' means the fp32 MFMA family on every layer

SHAPES = [(192, 1024, 8, 8, 1024, 3, 1), (192, 128, 64, 64, 128, 3, 1), (192, 256, 32, 32, 256, 3, 1), (192, 128, 64, 64, 256, 3, 2),
          (64, 256, 8, 8, 512, 5, 2), (192, 128, 16, 16, 256, 5, 2), (64, 1024, 8, 8, 1024, 3, 1)]
# the headline's layers (CIFAR ResNet critic over 3B = 192 rows / GP pass over 64, DIM 128; 4x4 stride 2 = the folded ConvMeanPool)
RESNET = [(192, 128, 32, 32, 128, 3, 1), (192, 128, 32, 32, 128, 4, 2), (192, 128, 16, 16, 128, 3, 1), (192, 128, 16, 16, 128, 4, 2),
          (192, 128, 8, 8, 128, 3, 1), (64, 128, 32, 32, 128, 3, 1), (64, 128, 16, 16, 128, 3, 1), (64, 128, 8, 8, 128, 3, 1),
          (384, 128, 8, 8, 128, 3, 1), (128, 128, 8, 8, 128, 3, 1), (128, 128, 16, 16, 128, 3, 1), (320, 128, 32, 32, 128, 3, 1)]
# the stride-2 layers of the headline (folded ConvMeanPool / UpsampleConv, 4x4 stride 2) at the row counts of the step
S2 = [(192, 128, 32, 32, 128, 4, 2), (128, 128, 32, 32, 128, 4, 2), (64, 128, 32, 32, 128, 4, 2), (320, 128, 32, 32, 128, 4, 2),
      (192, 128, 16, 16, 128, 4, 2), (128, 128, 16, 16, 128, 4, 2), (64, 128, 16, 16, 128, 4, 2), (320, 128, 16, 16, 128, 4, 2),
      (128, 128, 8, 8, 128, 4, 2), (320, 128, 8, 8, 128, 4, 2)]
# round 6: the stride-1 3x3 layers whose launches cannot fill the chip with pixel tiles (run with CTGAN_X3_HK=0 / 2 and with f32)
HK = [(64, 128, 8, 8, 128, 3, 1), (128, 128, 8, 8, 128, 3, 1), (192, 128, 8, 8, 128, 3, 1), (256, 128, 8, 8, 128, 3, 1), (384, 128, 8, 8, 128, 3, 1),
      (64, 128, 16, 16, 128, 3, 1), (128, 128, 16, 16, 128, 3, 1), (192, 128, 16, 16, 128, 3, 1)]
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
if len(sys.argv) > 2 and sys.argv[2] == 'hk':
    SHAPES = HK
if len(sys.argv) > 2 and sys.argv[2] == 'resnet':
    SHAPES = RESNET
if len(sys.argv) > 2 and sys.argv[2] == 's2':
    SHAPES = S2
if len(sys.argv) > 2 and sys.argv[2] == 'dcgan':      # the DCGAN critic's layers at the hand-scheduled step's row counts (4B = 256, B = 64)
    SHAPES = [(256, 128, 16, 16, 256, 5, 2), (256, 256, 8, 8, 512, 5, 2), (64, 128, 16, 16, 256, 5, 2), (64, 256, 8, 8, 512, 5, 2), (320, 256, 8, 8, 512, 5, 2), (320, 128, 16, 16, 256, 5, 2)]
if dt == 'f32':
    dt = None


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


print('dbg16=%s dtype=%s' % (os.environ.get('CTGAN_DBG16', '0'), dt))
print('%-36s %10s %10s %10s   (TFLOP/s; us)' % ('(N,C,H,W,K,R,stride)', 'fwd', 'dgrad', 'wgrad'))
for N, C, H, W, Ko, R, st in SHAPES:
    g = K.ConvGeom(C, H, W, Ko, R, R, st, False)
    x = K.empty_cl(N, C, H, W, 'cuda').normal_()
    w = (torch.randn(R, R, C, Ko, device='cuda') * 0.02)
    K._STABLE_PTRS.add(w.data_ptr())          # packed once, as a parameter would be
    gy = K.empty_cl(N, Ko, g.P, g.Q, 'cuda').normal_()
    fl = 2.0 * N * g.P * g.Q * Ko * R * R * C
    with K.mma_dtype(dt):
        tf = timed(lambda: K.conv_fwd(x, w, None, g)); kf = K.last_kernel()
        td = timed(lambda: K.conv_dgrad(gy, w, g, N)); kd = K.last_kernel()
        tw, kw = 1.0, '-'
        if not (len(sys.argv) > 2 and sys.argv[2] == 'hk'):
            tw = timed(lambda: K.conv_wgrad(x, gy, g)); kw = K.last_kernel()
    print('%-36s %5.0f %4.0fus %5.0f %4.0fus %5.0f %4.0fus   %s | %s | %s' % (str((N, C, H, W, Ko, R, st)), fl / tf / 1e12, tf * 1e6, fl / td / 1e12, td * 1e6,
                                                                 fl / tw / 1e12, tw * 1e6, kf, kd, kw))
