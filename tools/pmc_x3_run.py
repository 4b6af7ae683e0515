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


def case(op, N, C, H, Ko, R, st, relu=False, mode='f32x3', wide_only=False):
    """mode None: the hybrid fp32 routing (the launches it leaves on the fp32 MFMA family / the few-channel kernels)."""
    g = K.ConvGeom(C, H, H, Ko, R, R, st, False)
    x = K.empty_cl(N, C, H, H, 'cuda').normal_() if C > 4 else torch.randn(N, C, H, H, device='cuda')
    w = torch.randn(R, R, C, Ko, device='cuda') * 0.02
    K._STABLE_PTRS.add(w.data_ptr())
    gy = K.empty_cl(N, Ko, g.P, g.Q, 'cuda').normal_() if Ko > 4 else torch.randn(N, Ko, g.P, g.Q, device='cuda')
    with K.mma_dtype(mode):
        for _ in range(reps):
            if op == 'fwd':
                K.conv_fwd(x, w, None, g, relu_in=relu)
            elif op == 'dgrad':
                K.conv_dgrad(gy, w, g, N, out_strides=tuple(x.stride()) if C <= 4 else None)
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
    elif wide_only:
        alg = max(xb, yb) + min(xb, yb) + 4 * R * R * C * Ko      # few-channel kernels: the wide tensor once (+ the few-channel image, the filter)
    elif mode is None and not sym.startswith('conv16x3'):
        alg = xb + yb + 4 * R * R * C * Ko             # fp32 family: the fp32 filter
    else:
        alg = xb + yb + 6 * R * R * C * Ko             # (dy or x) + (dx or y) + three bf16 planes of the packed filter
    info[sym] = {'geometry': '%s (N,C,H,W,K,R,stride) = (%d,%d,%d,%d,%d,%d,%d)%s' % (op, N, C, H, H, Ko, R, st, ', relu on load' if relu else ''),
                 'flops_per_launch': flops, 'algorithmic_bytes_per_launch': alg}


case('fwd', 192, 128, 32, 128, 3, 1, relu=False)      # conv16x3hf_kernel<false, 4>
case('fwd', 192, 128, 16, 128, 3, 1, relu=True)       # conv16x3hf_kernel<true, 2>: 16x16 images on 64-pixel tiles (the critic's relu-on-load convs)
case('fwd', 384, 128, 8, 128, 3, 1, relu=True)        # conv16x3hf_kernel<true, 1>: the 384-row shared tail forward on 32-pixel tiles
case('dgrad', 128, 128, 32, 128, 4, 2)                # conv16x3sf_kernel<false, 2> (round 5; round 4: conv16x3p_kernel<2>): four-phase data gradient of the folded ConvMeanPool, one phase per workgroup
case('dgrad', 128, 128, 16, 128, 4, 2)                # conv16x3p_kernel<1>: the same on an 8x8 dy grid, 32-position tiles
case('fwd', 192, 128, 32, 128, 4, 2, relu=True)       # conv16x3sf_kernel<true, 2> (round 5): its forward at 192 rows, filter fragments from L2, 64-position tiles
# round 5: the merged backward's 8x8 data gradients (4B = 256 rows: dropout-pass rows + penalty rows), the 16x16 data gradients at 3B rows,
# the penalty's double-backward convs that stay on the fp32 family (64 rows of 8x8), the one-pixel-per-lane many -> few kernel
case('dgrad', 256, 128, 8, 128, 3, 1)                 # conv16x3hf_kernel<false, 1> at the merged backward's row count (overrides the 192-row entry)
case('dgrad', 192, 128, 16, 128, 3, 1)                # conv16x3hf_kernel<false, 2>
case('fwd', 64, 128, 8, 128, 3, 1, mode=None)         # round 6: conv16x3hk_kernel<false> (one channel chunk per wave) - until round 5 igemm_fwd_pipe_kernel<1, 2, 4, ...>
case('fwd', 64, 128, 16, 128, 4, 2, mode=None)        # igemm_fwd_pipe_kernel<1, 2, 4, ...>: what the fp32 pipe keeps of the double backward (the folded 4x4 stride-2 conv, 16x16 -> 8x8, 64 rows)
case('dgrad', 64, 3, 32, 128, 3, 1, mode=None, wide_only=True)      # m2f_px_kernel<3>: data gradient of the first critic conv on the penalty rows
case('fwd', 192, 3, 32, 128, 3, 1, mode=None, wide_only=True)       # f2m_kernel<3, 3, 3>: the first critic conv on [real ; fake ; x_hat]
# (weight gradients: tools/pmc_wgrad_col.sh on the step's grouped job table)
print(json.dumps(info))
