#!/usr/bin/env python
"""gpurun_out/pmc_x3/counters_*.csv + info.json -> the JSON bench.py reads (profiles/r04_pmc_traffic_x3.json; round 3: r03_...), keyed by device symbol.
HBM bytes per launch = 2 x FETCH_SIZE (the guide's gfx950 correction: rocprofv3 tallies 128-B requests of wide coalesced reads at
64 B) + WRITE_SIZE, both reported in KB by rocprofv3; SQ counters are summed over all shader engines."""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
info = json.load(open(os.path.join(d, 'info.json')))
if os.path.exists(os.path.join(d, 'info_group.json')):
    info.update(json.load(open(os.path.join(d, 'info_group.json'))))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, 'counters_*.csv')):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        m = re.search(r'((conv16x3hf|conv16x3hk|conv16x3sf|conv16x3h|conv16x3p|conv16|wgrad16_group|wgrad16|igemm_wgrad_pipe_group|igemm_fwd_pipe|m2f_px|f2m)_kernel<[^>]*>)', k)
        if m:
            agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
out = {'_how': 'tools/pmc_x3.sh on one MI355X: separate rocprofv3 --kernel-trace --pmc passes per counter group; means over the launches '
               'of each symbol (first launch of a geometry dropped: cold caches); hbm_bytes = 2 x FETCH_SIZE (gfx950 correction, '
               'MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KB -> B'}
for sym, rec in info.items():
    c = agg.get(sym)
    if not c:
        out[sym] = dict(rec, error='no counters collected for this symbol')
        continue
    mean = lambda n: (sum(c[n][1:]) / len(c[n][1:])) if len(c.get(n, [])) > 1 else (c[n][0] if c.get(n) else None)
    o = dict(rec)
    fs, ws = mean('FETCH_SIZE'), mean('WRITE_SIZE')
    if fs is not None and ws is not None:
        o.update(fetch_size_kb=fs, write_size_kb=ws, hbm_bytes_per_launch=2 * fs * 1024 + ws * 1024)
        o['traffic_over_algorithmic'] = round(o['hbm_bytes_per_launch'] / rec['algorithmic_bytes_per_launch'], 2)
    hit, miss = mean('TCC_HIT_sum'), mean('TCC_MISS_sum')
    if hit is not None and miss is not None and hit + miss > 0:
        o['l2_hit_rate'] = round(hit / (hit + miss), 4)
    busy, mf, wave = mean('SQ_BUSY_CYCLES'), mean('SQ_VALU_MFMA_BUSY_CYCLES'), mean('SQ_WAVE_CYCLES')
    # SQ_BUSY_CYCLES is summed over the 32 SQ instances (shader engines), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (checked in
    # round 1 against a kernel of known MFMA count: profiles/history/r01_pmc_sq_fwd_n128_h32.json): busy fraction of the matrix pipes =
    # MFMA_BUSY / (1024 * BUSY / 32)
    if busy:
        busy = 1024.0 * busy / 32.0
        o['shader_cycles_per_launch'] = round(busy / 1024.0)
        # split mode: six bf16 32x32x16 MFMAs (32,768 FLOP, 32 cycles each) per fp32 product; fp32 family: v_mfma_f32_32x32x2 (4,096 FLOP, 64 cycles)
        mflop, mcyc = rec.get('mfma_flop'), rec.get('mfma_cycles', 32)
        n_mfma = rec['flops_per_launch'] / mflop if mflop else rec['flops_per_launch'] * 6 / 32768.0
        o['mfma_busy_check'] = '%.4g MFMAs x %d cycles = %.4g vs SQ_VALU_MFMA_BUSY_CYCLES %.4g' % (n_mfma, mcyc, n_mfma * mcyc, mf or 0)
    for key, num, den in (('mfma_busy_frac', mf, busy), ('valu_active_frac', mean('SQ_ACTIVE_INST_VALU'), wave),
                          ('lds_wait_frac', mean('SQ_WAIT_INST_LDS'), wave), ('wave_parked_frac', mean('SQ_WAIT_ANY'), wave),
                          ('issue_stall_frac', mean('SQ_WAIT_INST_ANY'), wave), ('inst_active_frac', mean('SQ_ACTIVE_INST_ANY'), wave)):
        if num is not None and den:
            o[key] = round(num / den, 4)
    o['raw_means'] = {n: mean(n) for n in sorted(c)}
    out[sym] = o
print(json.dumps(out, indent=1))
