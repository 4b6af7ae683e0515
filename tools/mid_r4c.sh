#!/bin/bash
tag=${1:-mid3}; o=gpurun_out/$tag; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_dcgan_step.py tests/test_lsun128.py tests/test_gan_64x64.py tests/test_gpu_kernels16.py tests/test_gpu_graph_loop.py -m gpu -q -x --durations=8 > $o/tests_some.log 2>&1; echo "tests rc=$?"; tail -14 $o/tests_some.log
for env in "" "CTGAN_LRELU_DROP=0" "CTGAN_DCGAN_BATCH_FAKES=0" "CTGAN_LRELU_DROP=0 CTGAN_DCGAN_BATCH_FAKES=0 CTGAN_DEFER_16BIT=0"; do
  env $env python bench.py --config cifar_dcgan_bf16 --steps 20 --warmup 5 > $o/b.json 2> $o/b.err; echo "dcgan_bf16 [$env] rc=$?"; python -c "
import json; r=json.load(open('$o/b.json')); print(r['value'], r['ms_per_step'], r['config'].get('dispatches_per_step'))"
done
cp $o/b.json $o/bench_cifar_dcgan_bf16_alloff.json
python bench.py --config cifar_dcgan_bf16 --steps 20 --warmup 5 > $o/bench_cifar_dcgan_bf16.json 2> $o/b.err
for env in "" "CTGAN_DCGAN_BATCH_FAKES=0"; do
  env $env python bench.py --config lsun128_f16 --steps 8 --warmup 2 > $o/b2.json 2> $o/b2.err; echo "lsun128_f16 [$env] rc=$?"; python -c "
import json; r=json.load(open('$o/b2.json')); print(r['value'], r['ms_per_step'])"
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench.json 2> $o/bench.err; python -c "
import json; r=json.load(open('$o/bench.json')); print('headline', r['value'], r['ms_per_step'], r['config'].get('host_feed'))"
