#!/usr/bin/env python
"""Per (kernel variant, conv geometry) table of one eager iteration of the headline (CIFAR ResNet CT-WGAN step, batch 64): launches,
time and TFLOP/s - which layers a conv family / mode spends the step on.
usage: python tools/shape_prof_resnet.py [f32 (fp32 MFMA family only) | hybrid (the default routing) | f32x3 | bf16] [top]   (GPU box)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.kernels as K
import ctgan_amd.tflib as lib
from ctgan_amd.engine import GraphedTrainer

dt = sys.argv[1] if len(sys.argv) > 1 else 'f32'
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib.delete_all_params(); lib.set_seed(0); R.configure(); R.build_params('cuda')
K.X3_HYBRID = dt == 'hybrid'
K.set_mma_dtype(None if dt in ('f32', 'hybrid') else dt)
tr = R.Trainer(seed=1)
B = R.cfg.BATCH_SIZE
rng = np.random.default_rng(0)
batch = (torch.from_numpy(rng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(), torch.from_numpy(rng.integers(0, 10, (B,), dtype=np.int32)).cuda())
eng = GraphedTrainer(tr, use_graphs=False)
for it in (1, 2):
    eng.train_iteration(it, lambda: batch)
torch.cuda.synchronize()
K.PROFILE = []; K.PROFILE_REPS = 4
eng.train_iteration(3, lambda: batch); torch.cuda.synchronize()
prof, K.PROFILE, K.PROFILE_REPS = K.PROFILE, None, 1
agg = {}
for kname, fl, e0, e1, reps, shp, _sym in prof:
    a = agg.setdefault((kname, shp), [0, 0.0, 0.0])
    a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3 / reps
tot = sum(a[2] for a in agg.values())
print('%s: total conv time %.3f ms, %d launches, %.1f TF average' % (dt, tot * 1e3, len(prof), sum(a[1] for a in agg.values()) / tot / 1e12))
print('%-34s %-36s %4s %9s %8s %7s' % ('kernel', '(N,C,H,W,K,R,stride,up)', 'n', 'us/launch', 'ms', 'TF'))
for (kname, shp), (n, fl, t) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:top]:
    print('%-34s %-36s %4d %9.1f %8.3f %7.1f' % (kname[:34], str(shp), n, t / n * 1e6, t * 1e3, fl / t / 1e12))
