#!/usr/bin/env python
"""Run the headline's split-mode (f32x3) kernels - one geometry per device symbol - repeatedly, for rocprofv3 --pmc passes
(tools/pmc_x3.sh).  Prints one JSON line: symbol -> geometry, FLOPs and algorithmic bytes per launch."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
K.X3_HYBRID = True
info = {}


def case(op, N, C, H, Ko, R, st, relu=False):
    g = K.ConvGeom(C, H, H, Ko, R, R, st, False)
    x = K.empty_cl(N, C, H, H, 'cuda').normal_()
    w = torch.randn(R, R, C, Ko, device='cuda') * 0.02
    K._STABLE_PTRS.add(w.data_ptr())
    gy = K.empty_cl(N, Ko, g.P, g.Q, 'cuda').normal_()
    with K.mma_dtype('f32x3'):
        for _ in range(reps):
            if op == 'fwd':
                K.conv_fwd(x, w, None, g, relu_in=relu)
            elif op == 'dgrad':
                K.conv_dgrad(gy, w, g, N)
            else:
                K.conv_wgrad(x, gy, g, relu_x=relu)
    torch.cuda.synchronize()
    sym = K.last_symbol() if op != 'wgrad' else None
    flops = 2.0 * N * g.P * g.Q * Ko * R * R * C
    xb, yb = 4 * N * C * H * H, 4 * N * Ko * g.P * g.Q
    if op == 'wgrad':
        # (the last launch of a weight gradient is its split-K reduction: name the GEMM kernel explicitly)
        sym = 'wgrad16_kernel<3, 2, 2, %s>' % ('true' if relu else 'false')
        alg = xb + yb + 4 * R * R * C * Ko
    else:
        alg = xb + yb + 6 * R * R * C * Ko             # (dy or x) + (dx or y) + three bf16 planes of the packed filter
    info[sym] = {'geometry': '%s (N,C,H,W,K,R,stride) = (%d,%d,%d,%d,%d,%d,%d)%s' % (op, N, C, H, H, Ko, R, st, ', relu on load' if relu else ''),
                 'flops_per_launch': flops, 'algorithmic_bytes_per_launch': alg}


case('fwd', 192, 128, 32, 128, 3, 1, relu=False)      # conv16x3hf_kernel<false, 4>
case('fwd', 192, 128, 16, 128, 3, 1, relu=True)       # conv16x3hf_kernel<true, 2>: 16x16 images on 64-pixel tiles (the critic's relu-on-load convs)
case('fwd', 384, 128, 8, 128, 3, 1, relu=True)        # conv16x3hf_kernel<true, 1>: the 384-row shared tail forward on 32-pixel tiles
case('dgrad', 192, 128, 8, 128, 3, 1)                 # conv16x3hf_kernel<false, 1>: the main pass's 8x8 data gradients
case('dgrad', 128, 128, 32, 128, 4, 2)                # conv16x3p_kernel<2> (round 4): four-phase data gradient of the folded ConvMeanPool, 512 workgroups of eight waves
case('dgrad', 128, 128, 16, 128, 4, 2)                # conv16x3p_kernel<1>: the same on an 8x8 dy grid, 32-position tiles
case('fwd', 192, 128, 32, 128, 4, 2, relu=True)       # conv16_kernel<3, 2, 1, 32, true, false>: its forward at 192 rows on 128-kout x 64-pixel tiles
# (weight gradients: tools/pmc_wgrad_col.sh on the step's grouped job table)
print(json.dumps(info))
