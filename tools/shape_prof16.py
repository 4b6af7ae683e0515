#!/usr/bin/env python
"""Per (kernel variant, conv geometry) table of one eager iteration of an alternative bench configuration (bench.py
ALT_CONFIGS): time, TFLOP/s and the bytes-per-second each launch would need if it read its operands and wrote its result
exactly once (fp32 activations, 16-bit or fp32 filters) - which roofline the launch sits closer to.
usage: python tools/shape_prof16.py <config> [top]   (GPU box)"""
import importlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import ctgan_amd.kernels as K
import ctgan_amd.tflib as lib
from ctgan_amd.dcgan_step import DCGANTrainer

name = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
modname, dtype, _ = bench.ALT_CONFIGS[name]
M = importlib.import_module('ctgan_amd.' + modname)
lib.delete_all_params(); lib.set_seed(0); M.configure()
B = M.cfg.BATCH_SIZE
if hasattr(M, 'build_params'):
    M.build_params('cuda')
else:
    with torch.no_grad():
        M.Discriminator(M.Generator(2, noise=torch.zeros(2, 128, device='cuda')), u=[torch.ones(2, *s, device='cuda') for s in M.feat_shapes()])
K.set_mma_dtype(dtype)
tr = DCGANTrainer(M, seed=1)
batch = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, M.cfg.OUTPUT_DIM), dtype=np.int32)).cuda()
tr.train_iteration(1, lambda: batch); torch.cuda.synchronize()
K.PROFILE = []; K.PROFILE_REPS = 4
tr.train_iteration(1, lambda: batch); torch.cuda.synchronize()
prof, K.PROFILE, K.PROFILE_REPS = K.PROFILE, None, 1
agg = {}
for kname, fl, e0, e1, reps, shp, _sym in prof:
    a = agg.setdefault((kname, shp), [0, 0.0, 0.0])
    a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3 / reps
tot = sum(a[2] for a in agg.values())
print('%s: total conv time %.3f ms, %d launches, %.1f TF average' % (name, tot * 1e3, len(prof), sum(a[1] for a in agg.values()) / tot / 1e12))
print('%-28s %-36s %4s %9s %8s %7s %8s' % ('kernel', '(N,C,H,W,K,R,stride,up)', 'n', 'us/launch', 'ms', 'TF', 'min GB/s'))
for (kname, shp), (n, fl, t) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:top]:
    N, C, H, W, Ko, R, st, up = shp
    P, Q = -(-H // st), -(-W // st)
    wbytes = R * R * C * Ko * (2 if '16' in kname else 4)
    nbytes = N * C * H * W * 4 + N * Ko * P * Q * 4 + wbytes
    print('%-28s %-36s %4d %9.1f %8.3f %7.1f %8.0f' % (kname[:28], str(shp), n, t / n * 1e6, t * 1e3, fl / t / 1e12, nbytes / (t / n) / 1e9))
