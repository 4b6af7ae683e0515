#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table
(the same columns as rocprofv3's kernel_stats.csv: calls, total, average, percentage)."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    namecol = 'display_name' if 'display_name' in sym_cols else ('kernel_name' if 'kernel_name' in sym_cols else sym_cols[-1])
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.%s order by 3 desc" % (namecol, namecol))
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print('# %s' % path)
    print('# kernels: %d distinct, %d dispatches, busy %.3f ms over a %.3f ms span' % (
        len(rows), sum(r[1] for r in rows), total / 1e6, (span[1] - span[0]) / 1e6))
    print('%-110s %8s %12s %10s %10s %10s %6s' % ('Name', 'Calls', 'Total(us)', 'Avg(us)', 'Min(us)', 'Max(us)', '%'))
    for name, n, tot, mn, mx in rows[:top]:
        print('%-110s %8d %12.1f %10.2f %10.2f %10.2f %6.2f' % (name[:110], n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
