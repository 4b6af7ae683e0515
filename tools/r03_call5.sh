#!/bin/bash
o=gpurun_out/r03e; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -q -k "f32x3" > $o/tests_k16.log 2>&1; echo "k16 tests rc=$?"; tail -6 $o/tests_k16.log
python tools/conv16_bench.py f32x3 resnet 2>&1 | grep -v amdgpu.ids > $o/conv_bench_x3.txt; cat $o/conv_bench_x3.txt
CTGAN_X3_HALO_V=1 python tools/conv16_bench.py f32x3 resnet 2>&1 | grep "4, 2)" > $o/conv_bench_x3_v1_s2.txt; cat $o/conv_bench_x3_v1_s2.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 300 $o/bench.json; echo
CTGAN_X3_HALO_V=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_v1.json 2> $o/bench_v1.err; head -c 300 $o/bench_v1.json; echo
bash tools/prof_run.sh r03e --steps 20 --warmup 5 > $o/prof_run.log 2>&1; head -40 gpurun_out/prof_r03e/steady_state.txt | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -x -q -k "teacher_forced or graph_replay_loop_equals or whole_iteration_graph" > $o/tests_step.log 2>&1; echo "step tests rc=$?"; tail -4 $o/tests_step.log
