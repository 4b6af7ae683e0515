#!/bin/bash
o=gpurun_out/r03j; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -q -k "f32x3" > $o/tests_k16.log 2>&1; echo "k16 tests rc=$?"; tail -3 $o/tests_k16.log
for m in 0 1 0 1; do
  CTGAN_GP_STREAM=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_gps$m.json 2> $o/bench_gps$m.err; echo "gp stream $m: rc=$?"; head -c 230 $o/bench_gps$m.json | tail -c 90; echo
done
CTGAN_GP_STREAM=1 timeout 1200 python -m pytest tests -m gpu -x -q -k "graph_replay_loop_equals or whole_iteration_graph or teacher_forced" > $o/tests_gps.log 2>&1; echo "gps tests rc=$?"; tail -4 $o/tests_gps.log
