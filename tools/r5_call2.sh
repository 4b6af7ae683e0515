#!/bin/bash
# round 5: full GPU suite + bench (+ A/B of the m2f kernel) + profile
o=gpurun_out/r5b; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $o/tests_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -20 $o/tests_all.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 300 $o/bench.json; echo
CTGAN_M2F_PX=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --feed device > $o/bench_ring.json 2>/dev/null; head -c 200 $o/bench_ring.json; echo
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --feed device > $o/bench2.json 2>/dev/null; head -c 200 $o/bench2.json; echo
bash tools/prof_run.sh r5b --steps 20 --warmup 5 --feed device > $o/prof_run.log 2>&1
