#!/bin/bash
tag=${1:-mid2}; o=gpurun_out/$tag; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_graph_loop.py tests/test_lsun128.py tests/test_gan_64x64.py tests/test_gpu_dcgan_step.py tests/test_gpu_kernels16.py tests/test_gpu_bench_multirank.py -m gpu -q --durations=12 > $o/tests_some.log 2>&1; echo "tests rc=$?"; tail -22 $o/tests_some.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; python - <<PY
import json
r=json.load(open('$o/bench.json')); print(r['value'], r['ms_per_step'], r['config'].get('host_feed'), r['roofline']['kernel'], r['roofline']['frac'], r.get('gp_unit',{}).get('ms'))
PY
for cfg in cifar_dcgan_bf16 lsun128_f16; do
python bench.py --config $cfg --steps 10 --warmup 3 > $o/bench_$cfg.json 2> $o/bench_$cfg.err; echo "$cfg rc=$?"; head -c 220 $o/bench_$cfg.json; echo
CTGAN_DEFER_16BIT=0 python bench.py --config $cfg --steps 10 --warmup 3 > $o/bench_${cfg}_nodefer.json 2> $o/bench_${cfg}_nodefer.err; echo "$cfg nodefer rc=$?"; head -c 220 $o/bench_${cfg}_nodefer.json; echo
done
