#!/usr/bin/env python
"""Micro-benchmark of the grouped split-mode weight-gradient call on the job tables the ResNet critic / generator steps queue at full
width (DIM 128, B 64): GEMM launch(es) alone (phases = 1) and GEMM + batched reduction, HIP events over `reps` calls.
usage: python tools/wgrad_group_bench.py [d|g|both] [reps]      env CTGAN_WGRAD16_COL=0: the slice kernel only (round-3 path)"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
from ctgan_amd._lib import WgradGroup, lib

# (C, H, K, k, stride, rows per use, relu flags, bias flags)
D_STEP = [(128, 8, 128, 3, 1, (192, 64), (1, 0), (1, 0))] * 4 + [
    (128, 16, 128, 2, 2, (128, 64), (0, 0), (1, 0)),
    (128, 16, 128, 4, 2, (128, 64), (1, 0), (1, 0)),
    (128, 16, 128, 3, 1, (128, 64), (1, 0), (1, 0)),
    (128, 32, 128, 4, 2, (128, 64), (1, 0), (1, 0))]
G_STEP = [(128, 32, 128, 3, 1, (128,), (0,), (1,)), (128, 32, 128, 4, 2, (128,), (0,), (0,)), (128, 16, 128, 1, 1, (128,), (0,), (1,)),
          (128, 16, 128, 3, 1, (128,), (0,), (1,)), (128, 16, 128, 4, 2, (128,), (0,), (0,)), (128, 8, 128, 1, 1, (128,), (0,), (1,)),
          (128, 8, 128, 3, 1, (128,), (0,), (1,)), (128, 8, 128, 4, 2, (128,), (0,), (0,)), (128, 4, 128, 1, 1, (128,), (0,), (1,))]


def build(table):
    groups = []
    for C, H, Ko, k, st, Ns, relus, biases in table:
        geom = K.ConvGeom(C, H, H, Ko, k, k, st, False)
        segs = []
        for n, r, b in zip(Ns, relus, biases):
            x = K.empty_cl(n, C, H, H, 'cuda').normal_()
            gy = K.empty_cl(n, Ko, geom.P, geom.Q, 'cuda').normal_()
            segs.append((x, gy, bool(r), bool(b)))
        has_b = any(sg[3] for sg in segs)
        groups.append((segs, geom, torch.empty(k, k, C, Ko, device='cuda'), torch.empty(Ko, device='cuda') if has_b else None))
    return groups


def arrays(groups):
    n = len(groups)
    arr = (WgradGroup * n)()
    for i, (segs, g, dw, db) in enumerate(groups):
        G = arr[i]
        G.d = g.desc(segs[0][0].shape[0], segs[0][0].stride(), segs[0][1].stride())
        G.nseg = len(segs)
        for k, sg in enumerate(segs):
            G.Ns[k] = sg[0].shape[0]
            G.seg_flags[k] = (2 if sg[2] else 0) | (4 if sg[3] else 0)
            G.xs[k] = sg[0].data_ptr(); G.dys[k] = sg[1].data_ptr()
        G.dw = dw.data_ptr(); G.db = db.data_ptr() if db is not None else None
    return arr


def timed(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


which = sys.argv[1] if len(sys.argv) > 1 else 'both'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print('CTGAN_WGRAD16_COL=%s CTGAN_WGRAD16_COL_CHUNK=%s' % (os.environ.get('CTGAN_WGRAD16_COL', '1'), os.environ.get('CTGAN_WGRAD16_COL_CHUNK', '-')))
for name, table in (('d', D_STEP), ('g', G_STEP)):
    if which not in (name, 'both'):
        continue
    groups = build(table)
    arr = arrays(groups)
    takes = [i for i in range(len(groups)) if lib.ctgan_conv2d16_wgrad_group_workspace_bytes(ctypes.byref(arr[i]), 1, 3) > 0]
    sub = (WgradGroup * len(takes))()
    for k, i in enumerate(takes):
        sub[k] = arr[i]
    flops = sum(2.0 * sum(sg[0].shape[0] for sg in groups[i][0]) * groups[i][1].P * groups[i][1].Q * groups[i][1].R * groups[i][1].S * groups[i][1].C * groups[i][1].K
                for i in takes)
    nb = lib.ctgan_conv2d16_wgrad_group_workspace_bytes(sub, len(takes), 3)
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream

    def call(ph):
        rc = lib.ctgan_conv2d16_wgrad_group(sub, len(takes), 3, ws.data_ptr(), nb, ph, st)
        assert rc == 0, lib.ctgan_last_error()
    t1 = timed(lambda: call(1), reps)
    kinds = lib.ctgan_debug_last_wgrad_group_kinds()
    t3 = timed(lambda: call(3), reps)
    print('%s step: %d of %d filters in the grouped split-mode call, %.2f GFLOP, slabs %.1f MB, kernels(col=1|slice=2)=%d: GEMM %.1f us = %.1f TFLOP/s (%.3f of 2500/6); '
          'GEMM + reduction %.1f us' % (name, len(takes), len(groups), flops / 1e9, nb / 1e6, kinds, t1, flops / t1 / 1e6, flops / t1 / 1e6 / (2500 / 6), t3))
