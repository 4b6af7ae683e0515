#!/bin/bash
# mid-round check: full GPU suite with durations, the headline bench, a steady-state rocprof summary
tag=${1:-mid1}; o=gpurun_out/$tag; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -q --durations=40 > $o/tests_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -60 $o/tests_all.log
python bench.py --steps 30 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 400 $o/bench.json; echo
bash tools/prof_run.sh $tag --steps 20 --warmup 5 > $o/prof_run.log 2>&1
head -45 gpurun_out/prof_$tag/steady_state.txt
