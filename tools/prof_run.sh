#!/bin/bash
# rocprofv3 kernel-trace of one bench configuration -> per-kernel stats table under gpurun_out/prof_<tag>/.
#   tools/prof_run.sh <tag> <bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/raw -o trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-roofline > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT
db=$(find $out/raw -name '*.db' | head -1)
if [ -n "$db" ]; then python tools/prof_stats.py $db 60 > $out/kernel_stats.txt; python tools/prof_gaps.py $db 8 90 > $out/steady_state.txt 2>/dev/null; python tools/prof_seq.py $db 0 > $out/sequence.txt 2>/dev/null; fi
find $out/raw -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv 2>/dev/null
rm -rf $out/raw
head -30 $out/kernel_stats.txt
