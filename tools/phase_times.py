#!/usr/bin/env python
"""Per-phase time of the headline iteration under hipGraph replay: generator step, the batched fake draw, one critic step (each graph
replayed alone, back to back), and the whole-iteration graph.   usage: python tools/phase_times.py   (GPU box)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib
from ctgan_amd.engine import GraphedTrainer

lib.delete_all_params(); lib.set_seed(0); R.configure(); R.build_params(torch.device('cuda', 0))
tr = R.Trainer(seed=2024)
B = R.cfg.BATCH_SIZE
nrng = np.random.default_rng(1234)
batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(), torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).cuda())
           for _ in range(8)]
k = [0]


def nb():
    k[0] = (k[0] + 1) % 8
    return batches[k[0]]


eng = GraphedTrainer(tr)
assert eng.graphed and eng.it_graph is not None, (eng.graph_error, eng.it_graph_error)
for it in range(1, 6):
    eng.train_iteration(it, nb)
tr.set_lr(0.0)          # replays below must not move the weights far


def timed(graph, reps=20):
    graph.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g, f, d, it = timed(eng.g_graph), timed(eng.f_graph), timed(eng.d_graph), timed(eng.it_graph)
print('generator step %.3f ms | fake draw (5 x 64 rows) %.3f ms | critic step %.3f ms (x %d = %.3f) | sum %.3f | iteration graph %.3f ms'
      % (g, f, d, R.cfg.N_CRITIC, d * R.cfg.N_CRITIC, g + f + d * R.cfg.N_CRITIC, it))
