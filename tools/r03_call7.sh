#!/bin/bash
o=gpurun_out/r03g; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -q -k "f32x3" > $o/tests_k16.log 2>&1; echo "k16 tests rc=$?"; tail -6 $o/tests_k16.log
python tools/conv16_bench.py f32x3 resnet 2>&1 | grep "8, 8," > $o/conv_bench_x3_sq64.txt; cat $o/conv_bench_x3_sq64.txt
CTGAN_X3_HF_SQ64=0 python tools/conv16_bench.py f32x3 resnet 2>&1 | grep "8, 8," > $o/conv_bench_x3_nosq64.txt; cat $o/conv_bench_x3_nosq64.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 300 $o/bench.json; echo
CTGAN_X3_HF_SQ64=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_nosq64.json 2> $o/bench_nosq64.err; head -c 300 $o/bench_nosq64.json; echo
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_b.json 2> $o/bench_b.err; head -c 300 $o/bench_b.json; echo
python bench.py --gp-unit-only > $o/gp_unit.json 2>/dev/null; head -c 200 $o/gp_unit.json; echo
CTGAN_X3_HF_SQ64=0 python bench.py --gp-unit-only > $o/gp_unit_nosq64.json 2>/dev/null; head -c 200 $o/gp_unit_nosq64.json; echo
timeout 1500 python -m pytest tests -m gpu -x -q -k "teacher_forced or graph_replay_loop_equals or whole_iteration_graph or fused_into_conv_epilogues or fused_critic_heads" > $o/tests_step.log 2>&1; echo "step tests rc=$?"; tail -4 $o/tests_step.log
