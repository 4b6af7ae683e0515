#!/usr/bin/env python
"""Steady-state analysis of a rocprofv3 kernel trace (rocpd SQLite) of `bench.py --no-roofline --no-cpu-baseline`:
the window of the last `iters` training iterations (delimited by the Adam update launches: 6 per iteration),
its busy time, idle gaps, and the per-kernel table inside that window.
usage: python tools/prof_gaps.py results.db [iters=8] [top=40]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, iters=8, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    namecol = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
    rows = cur.execute("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start" % namecol).fetchall()
    adam = [i for i, r in enumerate(rows) if any(k in r[0] for k in ('adam_kernel', 'adam_packed_kernel', 'adam_end_kernel'))]
    need = 6 * iters
    if len(adam) < need + 1:
        print('not enough adam launches', len(adam)); return
    lo, hi = adam[-need - 1] + 1, adam[-1] + 1
    win = rows[lo:hi]
    span = win[-1][2] - win[0][1]
    busy = sum(r[2] - r[1] for r in win)
    print('# %s' % path)
    print('# window: last %d iterations, %d dispatches (%.0f / iteration), span %.3f ms (%.3f ms / iteration), busy %.3f ms (%.1f %%)'
          % (iters, len(win), len(win) / iters, span / 1e6, span / 1e6 / iters, busy / 1e6, 100.0 * busy / span))
    gaps = defaultdict(lambda: [0, 0])
    hist = defaultdict(int)
    for a, b in zip(win[:-1], win[1:]):
        gp = max(0, b[1] - a[2])
        gaps[a[0]][0] += 1; gaps[a[0]][1] += gp
        hist[min(int(gp / 1000), 20)] += 1
    print('# gap histogram (us: count):', ' '.join('%d:%d' % (k, hist[k]) for k in sorted(hist)))
    agg = defaultdict(lambda: [0, 0])
    for n, s, e in win:
        agg[n][0] += 1; agg[n][1] += e - s
    print('%-100s %8s %10s %9s %6s %10s' % ('Name', 'Calls/it', 'us/it', 'Avg(us)', '%span', 'gap-after us/it'))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-100s %8.1f %10.1f %9.2f %6.2f %10.1f' % (n[:100], c / iters, t / 1e3 / iters, t / c / 1e3, 100.0 * t / span, gaps[n][1] / 1e3 / iters))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 40)
