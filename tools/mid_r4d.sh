#!/bin/bash
tag=${1:-mid4}; o=gpurun_out/$tag; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_dcgan_step.py tests/test_lsun128.py tests/test_gan_64x64.py tests/test_gpu_kernels16.py -m gpu -q --durations=5 > $o/tests_some.log 2>&1; echo "tests rc=$?"; tail -12 $o/tests_some.log
for cfg in lsun128_f16 lsun128_f32; do
python bench.py --config $cfg --steps 6 --warmup 2 > $o/b_$cfg.json 2> $o/b_$cfg.err; echo "$cfg rc=$?"; tail -2 $o/b_$cfg.err; python -c "
import json; r=json.load(open('$o/b_$cfg.json')); print(r['value'], r['ms_per_step'], r['config'].get('last_d_terms'))"
done
python bench.py --config lsun128_f16 --steps 6 --warmup 2 --no-graph > $o/b_nograph.json 2> $o/b_nograph.err; echo "lsun f16 eager rc=$?"; tail -2 $o/b_nograph.err; python -c "
import json; r=json.load(open('$o/b_nograph.json')); print(r['value'], r['ms_per_step'], r['config'].get('last_d_terms'))"
