#!/usr/bin/env python
"""bench.py's roofline.by_kernel (HIP-event timings of an instrumented eager iteration, keyed by device symbol) beside the rocprofv3
steady-state table of the graph-replayed loop (tools/prof_gaps.py output): same symbols? launches per iteration? microseconds within 5 %?
usage: python tools/roofline_crosscheck.py <bench.json> <steady_state.txt>"""
import json
import re
import sys


def main(bench_path, steady_path):
    rec = json.loads([ln for ln in open(bench_path).read().splitlines() if ln.startswith('{')][-1])
    bk = rec['roofline']['by_kernel']
    prof = {}
    for ln in open(steady_path):
        m = re.match(r'(.{100})\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', ln)
        if m:
            prof[m.group(1).strip()] = (float(m.group(2)), float(m.group(4)))
    print('%-56s %9s %9s %9s %9s %7s' % ('device symbol', 'n bench', 'n rocprof', 'us bench', 'us rocprof', 'diff %'))
    worst, missing = 0.0, []
    for sym, v in bk.items():
        hits = [(name, c) for name, c in prof.items() if sym in name]
        if not hits:
            missing.append(sym)
            print('%-56s %9d %9s %9.2f %9s' % (sym[:56], v['launches'], '-', v['avg_launch_us'], 'NOT FOUND'))
            continue
        calls, us = hits[0][1]
        diff = 100.0 * (v['avg_launch_us'] - us) / us
        flag = ''
        if abs(calls - v['launches']) < 0.51:            # same launch mix: the averages are comparable
            worst = max(worst, abs(diff))
        else:
            flag = '  (launch counts differ: averages over different mixes)'
        print('%-56s %9d %9.1f %9.2f %9.2f %+7.1f%s' % (sym[:56], v['launches'], calls, v['avg_launch_us'], us, diff, flag))
    print('# symbols not found in the rocprof table: %d; worst |diff| among symbols with equal launch counts: %.1f %%' % (len(missing), worst))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
