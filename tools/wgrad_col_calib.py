#!/usr/bin/env python
"""Calibration of the filter-column kernel's planner (csrc/wgrad16c.hip slice_us): time of ONE round of ~256 equal workgroups with
1 / 2 / 3 taps per column, for a forced chunk, per arithmetic mode -> microseconds per slice.  usage: wgrad_col_calib.py <f32x3|bf16|f16>"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
if len(sys.argv) < 3:
    # one child per forced chunk (the switch is read once per process)
    for chunk in (512, 2048):
        subprocess.run([sys.executable, __file__, mode, str(chunk)], env=dict(os.environ, CTGAN_WGRAD16_COL_CHUNK=str(chunk)))
    sys.exit(0)
chunk = int(sys.argv[2])
import torch
import ctgan_amd.kernels as K
from ctgan_amd._lib import WgradGroup, lib
code = {'bf16': 1, 'f16': 2, 'f32x3': 3}[mode]
spx = 32 if code == 3 else 64
# (k, stride, H): taps per column 1 (1x1), 2 (4x4 stride 2), 3 (3x3); rows chosen so that tiles * splits ~ 256
for name, k, st, H, ntap in (('1x1', 1, 1, 16, 1), ('4x4s2', 4, 2, 32, 2), ('3x3', 3, 1, 16, 3)):
    geom = K.ConvGeom(128, H, H, 128, k, k, st, False)
    cols = {1: 1, 2: 8, 3: 3}[ntap]
    splits = max(1, 256 // cols)
    rows = max(1, splits * chunk // (geom.P * geom.Q))
    x = K.empty_cl(rows, 128, H, H, 'cuda').normal_(); gy = K.empty_cl(rows, 128, geom.P, geom.Q, 'cuda').normal_()
    arr = (WgradGroup * 1)()
    G = arr[0]
    G.d = geom.desc(rows, x.stride(), gy.stride()); G.nseg = 1; G.Ns[0] = rows; G.seg_flags[0] = 0
    G.xs[0] = x.data_ptr(); G.dys[0] = gy.data_ptr()
    dw = torch.empty(k, k, 128, 128, device='cuda'); G.dw = dw.data_ptr(); G.db = None
    nb = lib.ctgan_conv2d16_wgrad_group_workspace_bytes(arr, 1, code)
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    def call():
        rc = lib.ctgan_conv2d16_wgrad_group(arr, 1, code, ws.data_ptr(), nb, 1, stream)
        assert rc == 0, lib.ctgan_last_error()
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    Kg = rows * geom.P * geom.Q
    real_splits = (Kg + chunk - 1) // chunk
    wgs = cols * real_splits
    rounds = -(-wgs // 256)
    print('%s %-6s chunk %d: %d workgroups (%d rounds), %.1f us -> %.3f us per slice of %d px (ntap %d), kinds %d' % (
        mode, name, chunk, wgs, rounds, us, us / rounds / (chunk / spx), spx, ntap, lib.ctgan_debug_last_wgrad_group_kinds()))
