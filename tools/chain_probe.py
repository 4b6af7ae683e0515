#!/usr/bin/env python
"""Launch ctgan_conv2d16_chain8x8 (four convs, the backward program of critic_schedule) repeatedly on N images, for rocprofv3 kernel durations
(tools/chain_probe.sh).  usage: python tools/chain_probe.py [N]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C, H = 128, 8
ws = [torch.randn(3, 3, C, C, device='cuda') * 0.03 for _ in range(4)]
for w in ws:
    K._STABLE_PTRS.add(w.data_ptr())
x = K.empty_cl(N, C, H, H, 'cuda').normal_()
masks = [K.empty_cl(N, C, H, H, 'cuda').normal_() for _ in range(4)]
ctr = torch.tensor([7], dtype=torch.int64, device='cuda')
with K.mma_dtype('f32x3'):
    for _ in range(30):
        K.conv_chain8x8(x, [{'save': 1}, {'w': ws[0], 'op': 1, 'mask': masks[0], 'out': True},
                            {'w': ws[1], 'op': 1, 'mask': masks[1], 'resid': 1, 'drop': 1, 'save': 2, 'out': True},
                            {'w': ws[2], 'op': 1, 'mask': masks[2], 'out': True},
                            {'w': ws[3], 'op': 1, 'mask': masks[3], 'resid': 2, 'drop': 2, 'out': True}],
                        drops=[(0.5, 3, 4, N * 3 // 4), (0.5, 5, 6, N * 3 // 4)], seed=9, ctr=ctr)
torch.cuda.synchronize()
