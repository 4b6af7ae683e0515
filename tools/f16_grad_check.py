#!/usr/bin/env python
"""config[4] at the reference widths, B = 64: parameter gradients of one critic step with the convs on the fp16 / bf16 matrix
cores against the fp32 kernels on the same inputs and Philox streams (is fp16's exponent range a problem at batch 64?)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctgan_amd.gan_lsun128 as M
import ctgan_amd.kernels as K
import ctgan_amd.tflib as lib
from ctgan_amd.dcgan_step import DCGANTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib.delete_all_params(); lib.set_seed(0); M.configure(BATCH_SIZE=B)
M.build_params('cuda')
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, M.cfg.OUTPUT_DIM), dtype=np.int32)).cuda()
res = {}
for dt in (None, 'f16', 'bf16'):
    K.set_mma_dtype(dt)
    lib.bump_epoch()
    tr = DCGANTrainer(M, seed=1) if dt is None else tr
    tr.rng.ctr.zero_()
    tr.rng.begin_step()
    out = tr.d_losses(x)
    grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
    res[dt] = ({k: out[k].item() for k in ('cost', 'wgan_only', 'ct', 'gp')}, [g.clone() if g is not None else None for g in grads])
    print(dt, res[dt][0], flush=True)
names = [n for n, _ in tr.d_named]
for dt in ('f16', 'bf16'):
    rows = []
    for n, a, b in zip(names, res[dt][1], res[None][1]):
        if b is None or b.abs().max() < 1e-12:
            continue
        a = a.double().reshape(-1); b = b.double().reshape(-1)
        rows.append((((a - b).norm() / b.norm()).item(), torch.nn.functional.cosine_similarity(a.view(1, -1), b.view(1, -1)).item(), n, b.abs().max().item()))
    rows.sort(reverse=True)
    print(dt, 'worst', [(n, round(e, 4), round(c, 5), '%.1e' % m) for e, c, n, m in rows[:5]], 'median', round(sorted(r[0] for r in rows)[len(rows) // 2], 4))
