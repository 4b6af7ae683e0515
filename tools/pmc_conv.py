#!/usr/bin/env python
"""Run ONE conv shape/op repeatedly (for rocprofv3 --pmc passes).  usage: tools/pmc_conv.py op N H reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
op, N, H, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = K.ConvGeom(128, H, H, 128, 3, 3, 1, False)
x = K.empty_cl(N, 128, H, H, 'cuda').normal_()
w = torch.randn(3, 3, 128, 128, device='cuda') * 0.05
gy = K.empty_cl(N, 128, H, H, 'cuda').normal_()
for _ in range(reps):
    if op == 'fwd': K.conv_fwd(x, w, None, g)
    elif op == 'dgrad': K.conv_dgrad(gy, w, g, N)
    else: K.conv_wgrad(x, gy, g)
torch.cuda.synchronize()
print(K.last_kernel())
