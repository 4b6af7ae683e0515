#!/bin/bash
tag=${1:-mid5}; o=gpurun_out/$tag; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_dcgan_step.py tests/test_gpu_kernels.py tests/test_gpu_checkpoint.py -m gpu -q --durations=3 > $o/tests_some.log 2>&1; echo "tests rc=$?"; tail -6 $o/tests_some.log
for cfg in lsun128_f16 cifar_dcgan_bf16 cifar_dcgan_f32; do
python bench.py --config $cfg --steps 10 --warmup 3 > $o/bench_$cfg.json 2> $o/bench_$cfg.err; echo "$cfg rc=$?"; tail -1 $o/bench_$cfg.err; python -c "
import json; r=json.load(open('$o/bench_$cfg.json')); print(r['value'], r['ms_per_step'], r['config'].get('last_d_terms'))"
done
