#!/usr/bin/env python
"""The grouped weight-gradient launch of one critic step on synthetic operands, repeatedly - for rocprofv3 --pmc passes (tools/pmc_x3.sh).
Default (hybrid fp32 mode, kernels.X3_WGRAD_GROUP = 2): every queued weight gradient of the headline's D step at B = 64 rides
wgrad16_group_kernel<3, 2, 2> (split mode, 128x128 tiles, functional._flush_groups); with CTGAN_X3_WGRAD_GROUP=0 the queued ones of the
round-2 routing ride igemm_wgrad_pipe_group_kernel<1, 4, 2, 1> (fp32 MFMA).  Prints one JSON line: symbol -> geometry, FLOPs and
algorithmic bytes per launch."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
# (C, H, K, R, stride, [rows of the queued segments]): Discriminator.2.Conv2 (folded ConvMeanPool) and .2.Shortcut (pool + 1x1 as 2x2 stride 2)
# over the main pass (128 rows) and the GP double backward (64), the four 8x8 convs over 192 + 64, .1.Conv2 and .2.Conv1 over the main pass
# (192) and the GP double backward (64).  (fp32 group of the round-2 routing: the last two with their GP segments only.)
X3 = K.X3_WGRAD_GROUP >= 2 and K.X3_HYBRID
PROBLEMS = [(128, 16, 128, 4, 2, [128, 64]), (128, 16, 128, 2, 2, [128, 64]),
            (128, 8, 128, 3, 1, [192, 64]), (128, 8, 128, 3, 1, [192, 64]), (128, 8, 128, 3, 1, [192, 64]), (128, 8, 128, 3, 1, [192, 64]),
            (128, 32, 128, 4, 2, [192, 64] if X3 else [64]), (128, 16, 128, 3, 1, [192, 64] if X3 else [64])]
groups, flops, alg = [], 0.0, 0
for C, H, Ko, R, st, rows in PROBLEMS:
    g = K.ConvGeom(C, H, H, Ko, R, R, st, False)
    segs = []
    for n in rows:
        x = K.empty_cl(n, C, H, H, 'cuda').normal_()
        gy = K.empty_cl(n, Ko, g.P, g.Q, 'cuda').normal_()
        segs.append((x, gy, True, True))
        flops += 2.0 * n * g.P * g.Q * Ko * R * R * C
        alg += 4 * (x.numel() + gy.numel())
    dw = torch.empty(R, R, C, Ko, device='cuda')
    db = torch.empty(Ko, device='cuda')
    alg += 4 * (dw.numel() + db.numel())
    groups.append((segs, g, dw, db))
for _ in range(reps):
    K.conv_wgrad_group(groups)
torch.cuda.synchronize()
if X3:
    print(json.dumps({'wgrad16_group_kernel<3, 2, 2>': {
        'geometry': 'the 8 weight gradients of one critic step at B = 64 (16 segments: main pass + GP double backward each), split mode 128x128 tiles, one launch',
        'flops_per_launch': flops, 'algorithmic_bytes_per_launch': alg}}))
else:
    print(json.dumps({'igemm_wgrad_pipe_group_kernel<1, 4, 2, 1>': {
        'geometry': 'the 8 queued weight gradients of one critic step at B = 64 (2 + 2 + 4x2 + 1 + 1 segments), fp32 MFMA 64x128 tiles',
        'flops_per_launch': flops, 'algorithmic_bytes_per_launch': alg, 'mfma_flop': 4096, 'mfma_cycles': 64}}))
