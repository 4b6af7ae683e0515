#!/usr/bin/env python
"""Which large stride-1 3x3 launches of one eager headline iteration qualify for the split-mode hybrid routing, and what keeps the
others on the fp32 family.  usage: CTGAN_X3_HYBRID=1 CTGAN_X3_LOG=1 python tools/x3_route_log.py | sort | uniq -c   (GPU box)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib
from ctgan_amd.engine import GraphedTrainer

lib.delete_all_params(); lib.set_seed(0); R.configure(); R.build_params('cuda')
tr = R.Trainer(seed=1)
B = R.cfg.BATCH_SIZE
rng = np.random.default_rng(0)
batch = (torch.from_numpy(rng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(), torch.from_numpy(rng.integers(0, 10, (B,), dtype=np.int32)).cuda())
eng = GraphedTrainer(tr, use_graphs=False)
eng.train_iteration(1, lambda: batch)
torch.cuda.synchronize()
print('x3-log ---- second iteration', flush=True)
eng.train_iteration(2, lambda: batch)
torch.cuda.synchronize()
