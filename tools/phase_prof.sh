#!/bin/bash
# per-phase kernel tables: tools/phase_prof.sh <tag>  ->  gpurun_out/phase_<tag>/{g,f,d}_steady.txt  (last 20 replays of each graph)
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/phase_$tag; mkdir -p $out
for ph in g f d; do
  rocprofv3 --kernel-trace --stats -d $out/raw_$ph -o t -- python3 $GRAFT_REPO_ROOT/tools/phase_prof.py $ph 30 > $out/run_$ph.log 2>&1
  db=$(find $out/raw_$ph -name '*.db' | head -1)
  if [ -n "$db" ]; then (cd $GRAFT_REPO_ROOT && python tools/prof_tail.py $db 30 20 70 > $out/${ph}_steady.txt 2>&1; python tools/prof_seq.py $db 30 20 > $out/${ph}_sequence.txt 2>&1); fi
  rm -rf $out/raw_$ph
done
head -40 $out/g_steady.txt | cut -c1-170
