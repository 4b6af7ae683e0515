#!/bin/bash
# rocprofv3 kernel trace of the critic-forward + GP-backward unit (bench.py --gp-unit-only): per-kernel table of the graph replays.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_gp_unit; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/raw -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --gp-unit-only > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT
db=$(find $out/raw -name '*.db' | head -1)
python tools/prof_stats.py $db 40 > $out/kernel_stats.txt
rm -rf $out/raw
cat $out/bench.json; head -45 $out/kernel_stats.txt | cut -c1-175
