#!/bin/bash
o=gpurun_out/r03d; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -q > $o/tests_k16.log 2>&1; echo "k16 tests rc=$?"; tail -6 $o/tests_k16.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 300 $o/bench.json; echo
CTGAN_X3_HALO_V=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_v1.json 2> $o/bench_v1.err; head -c 300 $o/bench_v1.json; echo
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_b.json 2> $o/bench_b.err; head -c 300 $o/bench_b.json; echo
python bench.py --gp-unit-only > $o/gp_unit.json 2> $o/gp_unit.err; cat $o/gp_unit.json | head -c 600; echo
python tools/op_sources.py > $o/op_sources.txt 2>&1; tail -60 $o/op_sources.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "teacher_forced or graph_replay_loop_equals or whole_iteration_graph or fused_into_conv_epilogues or fused_critic_heads or grouped_wgrad or deferred or checkpoint or test_gpu_golden" > $o/tests_step.log 2>&1; echo "step tests rc=$?"; tail -4 $o/tests_step.log
