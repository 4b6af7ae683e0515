#!/usr/bin/env python
"""Per-kernel table of the dispatches AFTER the last idle gap of >= gap_ms in a rocprofv3 kernel trace (rocpd SQLite): the replays that
tools/phase_prof.py issues after a pause.   usage: python tools/prof_tail.py results.db reps [gap_ms=20] [top=60]"""
import sqlite3
import sys
from collections import defaultdict

path, reps = sys.argv[1], int(sys.argv[2])
gap_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
cur = sqlite3.connect(path).cursor()
sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
namecol = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
rows = cur.execute("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % namecol).fetchall()
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - rows[i - 1][2] >= gap_ms * 1e6:
        cut = i
win = rows[cut:]
span = win[-1][2] - win[0][1]
busy = sum(r[2] - r[1] for r in win)
print('# %s: %d dispatches after the last %.0f ms pause = %.1f per replay (%d replays), span %.3f ms / replay, busy %.1f %%' % (
    path, len(win), gap_ms, len(win) / reps, reps, span / 1e6 / reps, 100.0 * busy / span))
agg = defaultdict(lambda: [0, 0])
for n, s, e in win:
    agg[n][0] += 1; agg[n][1] += e - s
print('%-100s %8s %10s %9s %6s' % ('Name', 'Calls/rep', 'us/rep', 'Avg(us)', '%span'))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%-100s %8.1f %10.1f %9.2f %6.2f' % (n[:100], c / reps, t / 1e3 / reps, t / c / 1e3, 100.0 * t / span))
