#!/usr/bin/env python
"""gpurun_out/pmc_wcol/<tag>_*.csv -> per-symbol summary (means over the launches of each weight-gradient symbol, first launch dropped).
hbm_bytes = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KB -> B.  SQ_BUSY_CYCLES is summed over the 32 SQ
instances, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: busy fraction of the matrix pipes = MFMA_BUSY / (1024 * BUSY / 32)."""
import collections
import csv
import glob
import json
import os
import re
import sys

d, tag = sys.argv[1], sys.argv[2]
# the critic step's job table (tools/wgrad_group_bench.py D_STEP): algorithmic work and bytes of ONE grouped launch
D_STEP = [(128, 8, 128, 3, 1, (192, 64))] * 4 + [(128, 16, 128, 2, 2, (128, 64)), (128, 16, 128, 4, 2, (128, 64)), (128, 16, 128, 3, 1, (128, 64)),
                                                 (128, 32, 128, 4, 2, (128, 64))]
FLOPS = sum(2.0 * sum(Ns) * (H // st) ** 2 * k * k * C * Ko for C, H, Ko, k, st, Ns in D_STEP)
ALG_BYTES = sum(4.0 * (sum(Ns) * (H * H * C + (H // st) ** 2 * Ko) + k * k * C * Ko) for C, H, Ko, k, st, Ns in D_STEP)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, tag + '_*.csv')):
    base = os.path.basename(f)
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        m = re.search(r'((wgrad16c_group|wgrad16_group|reduce16_batch)_kernel(<[^>]*>)?)', k)
        if not m:
            continue
        if '_trace_' in base:
            if 'SQ_BUSY' in base:
                dur[m.group(1)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        else:
            agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for sym, c in agg.items():
    mean = lambda n: (sum(c[n][1:]) / len(c[n][1:])) if len(c.get(n, [])) > 1 else (c[n][0] if c.get(n) else None)
    o = {'launches_seen': max(len(v) for v in c.values())}
    if 'wgrad16' in sym:
        o.update(flops_per_launch=FLOPS, algorithmic_bytes_per_launch=ALG_BYTES,
                 geometry='critic-step job table: 8 filters x 2 uses, 67.65 GFLOP (tools/wgrad_group_bench.py d)')
    if dur.get(sym):
        v = dur[sym][1:] or dur[sym]
        o['us_under_pmc'] = round(sum(v) / len(v), 1)
    fs, ws = mean('FETCH_SIZE'), mean('WRITE_SIZE')
    if fs is not None and ws is not None:
        o.update(fetch_size_kb=fs, write_size_kb=ws, hbm_bytes_per_launch=2 * fs * 1024 + ws * 1024)
        if 'algorithmic_bytes_per_launch' in o:
            o['traffic_over_algorithmic'] = round(o['hbm_bytes_per_launch'] / o['algorithmic_bytes_per_launch'], 2)
    hit, miss = mean('TCC_HIT_sum'), mean('TCC_MISS_sum')
    if hit is not None and miss is not None and hit + miss > 0:
        o['l2_hit_rate'] = round(hit / (hit + miss), 4)
    busy, mf, wave = mean('SQ_BUSY_CYCLES'), mean('SQ_VALU_MFMA_BUSY_CYCLES'), mean('SQ_WAVE_CYCLES')
    if busy:
        busy = 1024.0 * busy / 32.0
        o['shader_cycles_per_launch'] = round(busy / 1024.0)
        if o.get('us_under_pmc'):
            o['shader_clock_ghz'] = round(o['shader_cycles_per_launch'] / o['us_under_pmc'] / 1e3, 3)
    for key, num, den in (('mfma_busy_frac', mf, busy), ('valu_active_frac', mean('SQ_ACTIVE_INST_VALU'), wave),
                          ('lds_wait_frac', mean('SQ_WAIT_INST_LDS'), wave), ('wave_parked_frac', mean('SQ_WAIT_ANY'), wave),
                          ('issue_stall_frac', mean('SQ_WAIT_INST_ANY'), wave), ('inst_active_frac', mean('SQ_ACTIVE_INST_ANY'), wave)):
        if num is not None and den:
            o[key] = round(num / den, 4)
    bc, ia = mean('SQ_LDS_BANK_CONFLICT'), mean('SQ_LDS_IDX_ACTIVE')
    if bc is not None and ia:
        o['lds_bank_conflict_frac'] = round(bc / ia, 4)
    o['raw_means'] = {n: mean(n) for n in sorted(c)}
    out[sym] = o
print(json.dumps(out, indent=1))
