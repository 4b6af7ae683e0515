#!/usr/bin/env python
"""Distance to an fp64 reference of the three conv families on the headline's layer shapes (generic fp32 operands): the fp32 MFMA family,
the split mode 'f32x3' (three bf16 terms per operand, six bf16 MFMAs per product) and plain bf16 - relative L2 and max-abs / max errors
of forward, data gradient and weight gradient.  usage: python tools/x3_precision.py > profiles/rNN_x3_precision.json   (GPU box)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctgan_amd.kernels as K
K.X3_HYBRID = False      # families are compared explicitly here: 'f32' means the fp32 MFMA family on every layer

SHAPES = [(16, 128, 32, 32, 128, 3, 1), (48, 128, 16, 16, 128, 3, 1), (16, 128, 32, 32, 128, 4, 2), (64, 128, 8, 8, 128, 3, 1)]


def ref_conv(x, w, st):
    """fp64 SAME conv on the GPU (torch, double): x NCHW, w HWIO."""
    R = w.shape[0]
    H = x.shape[2]
    P = -(-H // st)
    pad = max((P - 1) * st + R - H, 0)
    lo, hi = pad // 2, pad - pad // 2
    xp = torch.nn.functional.pad(x, (lo, hi, lo, hi))
    return torch.nn.functional.conv2d(xp, w.permute(3, 2, 0, 1), stride=st)


out = []
for N, C, H, W, Ko, R, st in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(7)
    geom = K.ConvGeom(C, H, W, Ko, R, R, st, False)
    x = K.empty_cl(N, C, H, W, 'cuda').normal_(generator=g)
    w = torch.randn(R, R, C, Ko, device='cuda', generator=g) / np.sqrt(R * R * C)
    gy = K.empty_cl(N, Ko, geom.P, geom.Q, 'cuda').normal_(generator=g)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = ref_conv(xd, wd, st)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [xd, wd], gy.double())
    row = {'shape': [N, C, H, W, Ko, R, st]}
    for mode in (None, 'f32x3', 'bf16'):
        with K.mma_dtype(mode):
            y = K.conv_fwd(x, w, None, geom); kf = K.last_kernel()
            gx = K.conv_dgrad(gy, w, geom, N); kd = K.last_kernel()
            gw = K.conv_wgrad(x, gy, geom); kw = K.last_kernel()
        e = {}
        for name, got, want in (('fwd', y, y_ref), ('dgrad', gx, gx_ref), ('wgrad', gw, gw_ref)):
            d = got.double() - want
            e[name] = {'rel_l2': float(d.norm() / want.norm()), 'max_over_max': float(d.abs().max() / want.abs().max())}
        e['kernels'] = [kf, kd, kw]
        row[mode or 'f32'] = e
    out.append(row)
print(json.dumps(out, indent=1))
