#!/bin/bash
o=gpurun_out/r03ar; mkdir -p $o
for i in 1 2 3 4 5 6; do python tests/ar_in_graph_check.py 32 8 4 2>$o/err_$i.log | tail -1; echo "run $i rc=${PIPESTATUS[0]}"; done
python -m pytest tests/test_gpu_graph_loop.py -q -k "all_reduce_captured" 2>&1 | tail -2
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $o/bench_2rank_gloo.json 2> $o/bench_2rank_gloo.err; echo "2rank rc=$?"; head -c 400 $o/bench_2rank_gloo.json | tail -c 200; echo
