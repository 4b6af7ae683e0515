#!/bin/bash
tag=${1:-mid7}; o=gpurun_out/$tag; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_dcgan_step.py tests/test_gpu_kernels16.py tests/test_lsun128.py tests/test_gpu_resnet_step.py -m gpu -q --durations=3 > $o/tests_some.log 2>&1; echo "tests rc=$?"; tail -6 $o/tests_some.log
for cfg in cifar_dcgan_bf16 cifar_dcgan_f32 lsun128_f16; do
python bench.py --config $cfg --steps 10 --warmup 3 > $o/bench_$cfg.json 2> $o/bench_$cfg.err; echo "$cfg rc=$?"; python -c "
import json; r=json.load(open('$o/bench_$cfg.json')); print(r['value'], r['ms_per_step'], r['config'].get('last_d_terms'))"
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench.json 2> $o/bench.err; python -c "
import json; r=json.load(open('$o/bench.json')); print('headline', r['value'], r['ms_per_step'])"
