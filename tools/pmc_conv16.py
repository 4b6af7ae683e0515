#!/usr/bin/env python
"""Run ONE conv geometry / operator of the 16-bit family repeatedly (for rocprofv3 --pmc passes).
usage: tools/pmc_conv16.py op N C H K R stride reps [dtype]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
K.X3_HYBRID = False      # families are compared explicitly here: 'f32' means the fp32 MFMA family on every layer
op = sys.argv[1]
N, C, H, Ko, R, st, reps = (int(v) for v in sys.argv[2:9])
dt = sys.argv[9] if len(sys.argv) > 9 else 'f16'
g = K.ConvGeom(C, H, H, Ko, R, R, st, False)
x = K.empty_cl(N, C, H, H, 'cuda').normal_()
w = torch.randn(R, R, C, Ko, device='cuda') * 0.02
K._STABLE_PTRS.add(w.data_ptr())
gy = K.empty_cl(N, Ko, g.P, g.Q, 'cuda').normal_()
with K.mma_dtype(dt if dt != 'f32' else None):
    for _ in range(reps):
        if op == 'fwd':
            K.conv_fwd(x, w, None, g)
        elif op == 'dgrad':
            K.conv_dgrad(gy, w, g, N)
        else:
            K.conv_wgrad(x, gy, g)
torch.cuda.synchronize()
print(K.last_kernel())
