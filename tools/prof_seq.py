#!/usr/bin/env python
"""Ordered dispatch list of ONE replay (the last) after the last idle gap of >= gap_ms in a rocprofv3 kernel trace (rocpd SQLite): name, grid,
workgroup, LDS, duration and the gap to the previous dispatch.   usage: python tools/prof_seq.py results.db reps [gap_ms=20]"""
import sqlite3
import sys

path, reps = sys.argv[1], int(sys.argv[2])
gap_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
# reps == 0: a bench.py trace - the last training iteration, delimited by the Adam launches (6 per iteration), as tools/prof_gaps.py does
cur = sqlite3.connect(path).cursor()
sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
namecol = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
gx = 'd.grid_size_x' if 'grid_size_x' in dcols else ('d.grid_x' if 'grid_x' in dcols else '0')
wx = 'd.workgroup_size_x' if 'workgroup_size_x' in dcols else ('d.workgroup_x' if 'workgroup_x' in dcols else '0')
lds = 'd.lds_block_size' if 'lds_block_size' in dcols else ('d.group_segment_size' if 'group_segment_size' in dcols else '0')
rows = cur.execute("select s.%s, d.start, d.end, %s, %s, %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % (namecol, gx, wx, lds)).fetchall()
if reps == 0:
    adam = [i for i, r in enumerate(rows) if 'adam_packed_kernel' in r[0] or 'adam_kernel' in r[0]]
    last = rows[adam[-7] + 1:adam[-1] + 1]
    per = len(last)
    while last and 'step_advance' in rows[adam[-7] + 1][0] and 'step_advance' in last[0][0]:
        last = last[1:]
else:
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] >= gap_ms * 1e6:
            cut = i
    win = rows[cut:]
    per = len(win) // reps
    last = win[len(win) - per:]
print('# %s: %d dispatches per replay; the last replay, %.3f ms' % (path, per, (last[-1][2] - last[0][1]) / 1e6))
print('%4s %9s %7s %10s %5s %7s  %s' % ('#', 't(us)', 'dur', 'grid', 'wg', 'lds', 'kernel'))
t0 = last[0][1]
for i, (n, s, e, g, w, l) in enumerate(last):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    print('%4d %9.1f %7.1f %10d %5d %7d  %s' % (i, (s - t0) / 1e3, (e - s) / 1e3, g, w, l, n[:110]))
