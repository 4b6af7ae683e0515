#!/bin/bash
# kernel-only durations (rocprofv3 --kernel-trace) of the stride-1 3x3 layers that cannot fill the chip with pixel tiles, on conv16x3hk_kernel
# (CTGAN_X3_HK=2), on the pixel-tiled split-mode kernels (=0) and on the fp32 pipe: gpurun_out/hk_prof.txt.  The Python loop of
# tools/conv16_bench.py is launch-bound below ~20 us per call - only the trace resolves these kernels.   usage: bash tools/hk_prof.sh
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/hk_prof; mkdir -p $out
: > $GRAFT_REPO_ROOT/gpurun_out/hk_prof.txt
for leg in "0 f32x3" "2 f32x3" "0 f32"; do
  set -- $leg
  export CTGAN_X3_HK=$1
  rm -rf $out/raw
  rocprofv3 --kernel-trace -d $out/raw -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/conv16_bench.py $2 hk > $out/run_$1_$2.log 2>&1
  f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
  echo "== CTGAN_X3_HK=$1 mode $2" >> $GRAFT_REPO_ROOT/gpurun_out/hk_prof.txt
  python3 - "$f" <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/hk_prof.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
groups = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    if not any(k in name for k in ('conv16x3', 'igemm_fwd', 'splitk')):
        continue
    key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '')))
    groups.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (name, grid), ts in groups.items():
    ts = sorted(ts)
    print('%-58s grid %-8s n=%3d  median %7.1f us  min %7.1f' % (name[:58], grid, len(ts), ts[len(ts) // 2], ts[0]))
PY
done
rm -rf $out/raw
cat $GRAFT_REPO_ROOT/gpurun_out/hk_prof.txt
