#!/usr/bin/env python
"""Per (kernel variant, conv geometry) time table of one eager training iteration (4 launches per event bracket).
usage: python tools/shape_prof.py [top]   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib

top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib.set_seed(1); R.configure(); R.build_params('cuda')
tr = R.Trainer(seed=1)
B = R.cfg.BATCH_SIZE
g = torch.Generator().manual_seed(0)
batch = (torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32).cuda(), torch.randint(0, 10, (B,), generator=g, dtype=torch.int32).cuda())
nb = lambda: batch
tr.train_iteration(1, nb); torch.cuda.synchronize()
K.PROFILE = []; K.PROFILE_REPS = 4
tr.train_iteration(1, nb); torch.cuda.synchronize()
prof, K.PROFILE, K.PROFILE_REPS = K.PROFILE, None, 1
agg = {}
for name, fl, e0, e1, reps, shp, _sym in prof:
    a = agg.setdefault((name, shp), [0, 0.0, 0.0])
    a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3 / reps
tot = sum(a[2] for a in agg.values())
print('total conv time %.3f ms, %d launches, %.1f TF average' % (tot * 1e3, len(prof), sum(a[1] for a in agg.values()) / tot / 1e12))
print('%-44s %-34s %5s %9s %8s %7s %9s' % ('kernel', '(N,C,H,W,K,R,stride,up)', 'n', 'us/launch', 'ms', 'TF', 'lost ms'))
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
for (name, shp), (n, fl, t) in rows[:top]:
    lost = t - fl / 120e12
    print('%-44s %-34s %5d %9.1f %8.3f %7.1f %9.3f' % (name[:44], str(shp), n, t / n * 1e6, t * 1e3, fl / t / 1e12, lost * 1e3))
