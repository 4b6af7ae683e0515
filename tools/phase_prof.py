#!/usr/bin/env python
"""Replay ONE of the iteration's graphs (g = generator step, f = batched fake draw, d = critic step) N times, for a rocprofv3 kernel trace of
that phase alone:   rocprofv3 --kernel-trace --stats -d out -o t -- python3 tools/phase_prof.py g 40"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib
from ctgan_amd.engine import GraphedTrainer

which, reps = sys.argv[1], int(sys.argv[2])
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
for it in range(1, 4):
    eng.train_iteration(it, nb)
tr.set_lr(0.0)
graph = {'g': eng.g_graph, 'f': eng.f_graph, 'd': eng.d_graph}[which]
torch.cuda.synchronize()
import time
time.sleep(0.1)          # a pause tools/prof_tail.py finds: the replays below are what it tabulates
for _ in range(reps):
    graph.replay()
torch.cuda.synchronize()
