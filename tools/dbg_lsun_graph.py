#!/usr/bin/env python
"""Diagnosis: full-width lsun128 (or any --config module) graphed vs eager trainer, loss terms per iteration.  usage: dbg_lsun_graph.py [f16|bf16|none] [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctgan_amd.kernels as K
import ctgan_amd.tflib as lib
from ctgan_amd.dcgan_step import DCGANTrainer
from ctgan_amd.engine import GraphedDCGANTrainer
import ctgan_amd.gan_lsun128 as M
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dt = None if dt == 'none' else dt
nrng = np.random.default_rng(5)
def run(graphs):
    lib.delete_all_params(); lib.set_device(None); lib.set_seed(3)
    M.configure(BATCH_SIZE=B)
    M.build_params('cuda')
    tr = DCGANTrainer(M, seed=11)
    if dt == 'f16': tr.loss_scale = 1024.0
    eng = GraphedDCGANTrainer(tr, (B, M.cfg.OUTPUT_DIM), torch.int32, use_graphs=graphs)
    assert eng.graphed == graphs, eng.graph_error
    k = [0]
    def nb():
        k[0] += 1; return batches[k[0] % len(batches)]
    recs = []
    for it in range(iters):
        out = eng.train_iteration(it, nb)
        recs.append({n: float(out[n].item()) for n in ('cost', 'wgan_only', 'ct', 'gp')})
        print('graphs' if graphs else 'eager ', it, recs[-1], 'fake absmax', float(eng.fake_all.abs().max()) if (graphs and eng.fake_all is not None) else None, flush=True)
    return recs, tr.d_opt.theta.clone(), tr.g_opt.theta.clone()
K.set_mma_dtype(dt)
M.configure(BATCH_SIZE=B)
batches = [torch.from_numpy(nrng.integers(0, 256, (B, M.cfg.OUTPUT_DIM), dtype=np.int32)).cuda() for _ in range(4)]
g = run(True); e = run(False)
print('theta equal', torch.equal(g[1], e[1]), torch.equal(g[2], e[2]), 'max diff', float((g[1]-e[1]).abs().max()), float((g[2]-e[2]).abs().max()))
