#!/usr/bin/env python
"""Diagnosis: lsun128 generator forward at n rows, eager vs captured in a hipGraph (f16 mode): which block output first differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
import ctgan_amd.functional as F
import ctgan_amd.tflib as lib
import ctgan_amd.gan_lsun128 as M
from ctgan_amd.rng import DeviceRNG
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 320
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 10
K.set_mma_dtype(None if dt == 'none' else dt)
lib.delete_all_params(); lib.set_device(None); lib.set_seed(3)
M.configure(BATCH_SIZE=64)
M.build_params('cuda')
taps = []
orig_rb, orig_norm, orig_suc = M.ResidualBlock, M.Normalize, M.ScaledUpsampleConv
def rb(name, *a, **k):
    o = orig_rb(name, *a, **k); taps.append((name, o.clone())); return o
def nm(name, *a, **k):
    o = orig_norm(name, *a, **k); taps.append((name, o.clone())); return o
def suc(name, *a, **k):
    o = orig_suc(name, *a, **k); taps.append((name, o.clone())); return o
M.ResidualBlock, M.Normalize, M.ScaledUpsampleConv = rb, nm, suc
rng = DeviceRNG(seed=5)
def fwd():
    lib.bump_epoch('Generator'); F.prepare_filters()
    rng.begin_step()
    with torch.no_grad():
        return M.Generator(n, rng=rng, groups=groups)
for _ in range(2):
    taps.clear(); y_e = fwd()
torch.cuda.synchronize()
e_taps = [(k, v.clone()) for k, v in taps]
K.reset_capture_workspaces()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    taps.clear(); fwd()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
taps.clear()
with torch.cuda.graph(g):
    y_g = fwd()
g_taps = list(taps)
g.replay(); torch.cuda.synchronize()
print('final: eager nan', bool(torch.isnan(y_e).any()), 'graph nan', bool(torch.isnan(y_g).any()), 'equal', bool(torch.equal(y_e, y_g)))
for (ke, ve), (kg, vg) in zip(e_taps, g_taps):
    print('%-28s %-18s eager absmax %.4g nan %d | graph absmax %.4g nan %d | equal %d' % (ke, tuple(ve.shape), float(ve.abs().max()), int(torch.isnan(ve).any()),
          float(vg.abs().max()), int(torch.isnan(vg).any()), int(torch.equal(ve, vg))))
