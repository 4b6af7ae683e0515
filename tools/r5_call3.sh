#!/bin/bash
o=gpurun_out/r5c; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_graph_loop.py tests/test_gpu_bench_multirank.py -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -5 $o/tests.log
python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --feed device > $o/bench_2rank_gloo.json 2> $o/bench_2rank_gloo.err; echo "2rank rc=$?"; python -c "
import json; r=json.load(open('$o/bench_2rank_gloo.json')); print(r['value'], r['ms_per_step'], json.dumps(r['config']['collective'])[:1500])"
tail -3 $o/bench_2rank_gloo.err
