#!/usr/bin/env python
"""Host-side cost of one training iteration under hipGraph replay: perf_counter around every replay / eager call of
GraphedTrainer.train_iteration (no device syncs inside the loop), next to the GPU time of the iteration.  Answers
"is the host ahead of the GPU?" - idle gaps at graph boundaries in the rocprofv3 trace mean it is not."""
import time
import numpy as np
import torch
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib
from ctgan_amd.engine import GraphedTrainer

lib.set_seed(0)
R.configure()
R.build_params('cuda')
tr = R.Trainer()
eng = GraphedTrainer(tr)
B = 64
nrng = np.random.default_rng(1)
batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(),
            torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).cuda()) for _ in range(4)]
cur = [0]


def nb():
    cur[0] = (cur[0] + 1) % 4
    return batches[cur[0]]


log = []
for name in ('d_graph', 'g_graph', 'f_graph'):
    g = getattr(eng, name)
    orig = g.replay

    def rep(orig=orig, name=name):
        t0 = time.perf_counter()
        orig()
        log.append((name, time.perf_counter() - t0))
    g.replay = rep
for it in range(1, 4):
    eng.train_iteration(it, nb)
torch.cuda.synchronize()
del log[:]
N = 10
t0 = time.perf_counter()
for it in range(4, 4 + N):
    eng.train_iteration(it, nb)
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('host time to issue %d iterations: %.2f ms / iteration; wall incl. GPU drain: %.2f ms / iteration' % (N, 1e3 * t_cpu / N, 1e3 * t_all / N))
agg = {}
for n, t in log:
    agg.setdefault(n, []).append(t)
for n, ts in agg.items():
    print('%s.replay(): %d calls, mean %.1f us, max %.1f us' % (n, len(ts), 1e6 * sum(ts) / len(ts), 1e6 * max(ts)))
