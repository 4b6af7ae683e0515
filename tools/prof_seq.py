#!/usr/bin/env python
"""Ordered launch sequence of the LAST unit in a rocprofv3 kernel trace (rocpd SQLite), the unit being delimited by a
marker kernel (default gp_mean_kernel: one per `bench.py --gp-unit-only` replay), with per-launch durations averaged
over the last `reps` units.  usage: python tools/prof_seq.py results.db [marker=gp_mean_kernel] [reps=10] [back=0]
(back = k: show the unit k markers before the last one instead, e.g. marker adam_kernel, reps 1, back 5 = the generator
step of the last iteration of a `bench.py --no-roofline` trace)"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', n)
    return (m.group(1) if m else n)[:60]


def main(path, marker='gp_mean_kernel', reps=10, back=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    namecol = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
    dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gx = 'd.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x' if 'grid_size_x' in dcols else '0,0,0,0'
    rows = cur.execute("select s.%s, d.start, d.end, %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start" % (namecol, gx)).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if back:
        marks = marks[:-back]
    if len(marks) < reps + 1:
        print('not enough markers', len(marks)); return
    L = marks[-1] - marks[-2]
    units = [rows[marks[-k - 1] + 1: marks[-k] + 1] for k in range(1, reps + 1)]
    units = [u for u in units if len(u) == L and [r[0] for r in u] == [r[0] for r in units[0]]]
    print('# %s: %d launches / unit, %d matching units' % (path, L, len(units)))
    span = sum(u[-1][2] - u[0][1] for u in units) / len(units) / 1e3
    busy = sum(sum(r[2] - r[1] for r in u) for u in units) / len(units) / 1e3
    print('# span %.1f us, busy %.1f us' % (span, busy))
    t0 = 0.0
    for i in range(L):
        r = units[0][i]
        dur = sum(u[i][2] - u[i][1] for u in units) / len(units) / 1e3
        gap = sum((u[i + 1][1] - u[i][2]) for u in units) / len(units) / 1e3 if i + 1 < L else 0.0
        wg = r[6] or 1
        print('%4d %-60s grid %6d x%3d x%3d wg %4d  %8.2f us  gap %5.2f  t=%8.1f' % (
            i, short(r[0]), (r[3] or 0) // wg, r[4] or 0, r[5] or 0, wg, dur, gap, t0))
        t0 += dur + gap


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'gp_mean_kernel', int(sys.argv[3]) if len(sys.argv) > 3 else 10,
         int(sys.argv[4]) if len(sys.argv) > 4 else 0)
