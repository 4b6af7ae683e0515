#!/usr/bin/env python
"""Time the three captured graphs of an iteration in isolation (back-to-back replays) and the full iteration: the difference
is what the host-side glue between replays (input copies, lr upload, graph launch gaps) costs on the GPU timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib
from ctgan_amd.engine import GraphedTrainer

lib.set_seed(0); R.configure(); R.build_params('cuda')
tr = R.Trainer(); eng = GraphedTrainer(tr)
B = 64
nrng = np.random.default_rng(1)
batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(),
            torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).cuda()) for _ in range(4)]
cur = [0]


def nb():
    cur[0] = (cur[0] + 1) % 4
    return batches[cur[0]]


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for it in range(1, 4):
    eng.train_iteration(it, nb)
tr.d_opt.set_lr(0.0); tr.g_opt.set_lr(0.0)
td = timed(eng.d_graph.replay, 20); tg = timed(eng.g_graph.replay, 20); tf = timed(eng.f_graph.replay, 20)
it = [4]


def full():
    eng.train_iteration(it[0], nb); it[0] += 1


tt = timed(full, 20)
print('d_graph %.3f ms, g_graph %.3f ms, f_graph %.3f ms: 5 D + G + F = %.3f ms; full iteration %.3f ms (glue %.3f ms)'
      % (td, tg, tf, 5 * td + tg + tf, tt, tt - 5 * td - tg - tf))
