#!/bin/bash
# kernel-only duration of chain8x8_kernel (256 and 64 images): gpurun_out/chain_probe.txt.  profiles/r06_chain_probe.txt was taken with a diagnosis build whose
# kernel read CTGAN_CHAIN_DBG (1 no MFMAs, 2 the filter stream re-reads one tap, 4 no LDS fragment reads, 8 no conv phase); those branches cost 40 us per
# launch by themselves and were removed from the product kernel - this script now times the product kernel only
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/chain_probe; mkdir -p $out
: > $GRAFT_REPO_ROOT/gpurun_out/chain_probe.txt
for n in 256 64; do for dbg in 0; do
  export CTGAN_CHAIN_DBG=$dbg
  rm -rf $out/raw
  rocprofv3 --kernel-trace -d $out/raw -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/chain_probe.py $n > $out/run.log 2>&1
  f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $n $dbg <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/chain_probe.txt
import csv, sys
ts = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if 'chain8x8' in r['Kernel_Name'])
print('images %4s dbg %s: n=%d median %7.1f us min %7.1f' % (sys.argv[2], sys.argv[3], len(ts), ts[len(ts) // 2] if ts else -1, ts[0] if ts else -1))
PY
done; done
rm -rf $out/raw
cat $GRAFT_REPO_ROOT/gpurun_out/chain_probe.txt
