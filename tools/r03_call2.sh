#!/bin/bash
o=gpurun_out/r03b; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -x -q -k "f32x3" > $o/tests_k16.log 2>&1; echo "k16 tests rc=$?"; tail -4 $o/tests_k16.log
CTGAN_X3_HALO_V=1 python tools/conv16_bench.py f32x3 resnet > $o/conv_bench_x3_v1.txt 2>&1
CTGAN_X3_HALO_V=2 python tools/conv16_bench.py f32x3 resnet > $o/conv_bench_x3_v2.txt 2>&1
python tools/conv16_bench.py f32 resnet > $o/conv_bench_f32.txt 2>&1
cat $o/conv_bench_x3_v1.txt; cat $o/conv_bench_x3_v2.txt; cat $o/conv_bench_f32.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_hf.json 2> $o/bench_hf.err; echo "bench rc=$?"; head -c 400 $o/bench_hf.json; echo
CTGAN_X3_HALO_V=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_v1.json 2> $o/bench_v1.err; head -c 300 $o/bench_v1.json; echo
timeout 1200 python -m pytest tests -m gpu -x -q -k "teacher_forced or graph_replay_loop_equals or whole_iteration_graph or fused_into_conv_epilogues or fused_critic_heads or matches_oracle_loop or smoke" > $o/tests_step.log 2>&1; echo "step tests rc=$?"; tail -4 $o/tests_step.log
