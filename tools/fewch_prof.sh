#!/bin/bash
# kernel-only durations of the few-channel kernels per shape (tools/fewch_bench.py under rocprofv3 --kernel-trace): gpurun_out/fewch_prof.txt
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/fewch_prof; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/raw -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/fewch_bench.py > $out/run.log 2>&1
f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
head -2 "$f" > $out/header.txt
python3 - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/fewch_prof.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# group consecutive launches of the same kernel with the same grid size: one group per (kernel, shape) of the benchmark
groups = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    if not any(k in name for k in ('f2m', 'm2f', 'fw_wgrad', 'fw_reduce')):
        continue
    key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', ''))
    groups.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (name, grid, wg), ts in groups.items():
    ts = sorted(ts)
    print('%-40s grid %-8s wg %-4s n=%3d  median %7.1f us  min %7.1f' % (name[:40], grid, wg, len(ts), ts[len(ts) // 2], ts[0]))
PY
rm -rf $out/raw
cat $GRAFT_REPO_ROOT/gpurun_out/fewch_prof.txt
