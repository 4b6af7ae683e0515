#!/bin/bash
# One gpurun call: kernel trace of the headline bench -> ordered launch sequence of every step of the last iteration (the adam_kernel
# launch ends a step), plus the steady-state table.   usage: gpurun -- 'bash tools/seq_call.sh <tag> [pytest -k expr]'
tag=${1:-seq}; o=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $o
if [ -n "$2" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$2" > $o/tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/tests.log; fi
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 250 $o/bench.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $o/raw -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline > $o/prof_bench.json 2> $o/prof_bench.err
cd $GRAFT_REPO_ROOT
db=$(find $o/raw -name '*.db' | head -1)
python tools/prof_gaps.py $db 8 90 > $o/steady_state.txt 2>/dev/null
for b in 0 1 2 5 6; do python tools/prof_seq.py $db adam_packed_kernel 1 $b > $o/seq_back$b.txt 2>&1; done
rm -rf $o/raw
head -4 $o/steady_state.txt
