#!/usr/bin/env python
"""Grid-size histogram of the kernels whose name contains a pattern, inside the steady-state window of a rocprofv3
kernel trace (see tools/prof_gaps.py): identifies WHICH tensors the torch plumbing kernels (add / fill / copy) touch.
usage: python tools/prof_grids.py results.db pattern [iters=8]"""
import sqlite3, sys
from collections import Counter

db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cur = db.cursor()
sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
namecol = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
gx = 'grid_size_x' if 'grid_size_x' in dcols else ('grid_x' if 'grid_x' in dcols else None)
wx = 'workgroup_size_x' if 'workgroup_size_x' in dcols else None
rows = cur.execute("select s.%s, d.start, d.end, d.%s, d.%s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"
                   % (namecol, gx, wx)).fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
lo, hi = adam[-6 * iters - 1] + 1, adam[-1] + 1
c = Counter()
for n, s, e, g, w in rows[lo:hi]:
    if pat in n:
        c[(g, w)] += 1
for (g, w), k in sorted(c.items(), key=lambda kv: -kv[1]):
    print('grid %9d threads (wg %4d): %6.1f launches / iteration' % (g, w, k / iters))
