#!/usr/bin/env python
"""Diagnosis: GraphedDCGANTrainer's fake-batch graph on lsun128 f16 at B=64: NaN?  Variations by env DBG_ORDER=f_first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
import ctgan_amd.tflib as lib
import ctgan_amd.gan_lsun128 as M
import ctgan_amd.engine as E
from ctgan_amd.dcgan_step import DCGANTrainer
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K.set_mma_dtype(None if dt == 'none' else dt)
lib.delete_all_params(); lib.set_device(None); lib.set_seed(3)
M.configure(BATCH_SIZE=B)
M.build_params('cuda')
tr = DCGANTrainer(M, seed=11)
tr.loss_scale = 1024.0 if dt == 'f16' else 1.0
# workspace trace
orig_ws = K.workspace
log = []
def ws(nbytes, device):
    b = orig_ws(nbytes, device)
    log.append((int(nbytes), b.numel(), b.data_ptr(), bool(torch.cuda.is_current_stream_capturing())))
    return b
K.workspace = ws
eng = E.GraphedDCGANTrainer(tr, (B, M.cfg.OUTPUT_DIM), torch.int32, use_graphs=True)
print('graphed', eng.graphed, eng.graph_error)
cap = [l for l in log if l[3]]
print('capture-time workspace requests:', len(cap), 'distinct buffers', len(set(l[2] for l in cap)), 'max request MB', max(l[0] for l in cap) / 1e6, 'buffer sizes MB', sorted(set(round(l[1] / 1e6, 1) for l in cap)))
eng.f_graph.replay(); torch.cuda.synchronize()
print('fake_all after f replay: nan', bool(torch.isnan(eng.fake_all).any()), 'absmax', float(eng.fake_all.abs().max()))
eng.g_graph.replay(); eng.f_graph.replay(); torch.cuda.synchronize()
print('after g then f replay: nan', bool(torch.isnan(eng.fake_all).any()))
f = tr.generate_fakes(5); torch.cuda.synchronize()
print('eager generate_fakes nan', bool(torch.isnan(f).any()), float(f.abs().max()))
