#!/bin/bash
# round 5, first GPU call: the hand-scheduled critic step - its tests, the step / loop suites, and the A/B bench
o=gpurun_out/r5a; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_resnet_step.py tests/test_gpu_graph_loop.py -m gpu -x -q --durations=8 > $o/tests_step.log 2>&1; echo "step tests rc=$?"; tail -15 $o/tests_step.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "adam or dropout" > $o/tests_k.log 2>&1; echo "kernel subset rc=$?"; tail -3 $o/tests_k.log
python bench.py --steps 30 --warmup 5 > $o/bench_merged.json 2> $o/bench_merged.err; echo "bench rc=$?"; head -c 400 $o/bench_merged.json; echo
CTGAN_MERGED_BWD=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $o/bench_autograd.json 2> $o/bench_autograd.err; echo "bench autograd rc=$?"; head -c 300 $o/bench_autograd.json; echo
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --feed device > $o/bench_merged2.json 2>/dev/null; head -c 200 $o/bench_merged2.json; echo
CTGAN_MERGED_BWD=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --feed device > $o/bench_autograd2.json 2>/dev/null; head -c 200 $o/bench_autograd2.json; echo
