#!/bin/bash
# round 5, call 7: tiled single forward pack, lane-parallel split-K reduction
o=gpurun_out/r5o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels16.py -m gpu -x -q > $o/tests_k.log 2>&1; echo "k rc=$?"; tail -4 $o/tests_k.log
timeout 900 python -m pytest tests/test_gpu_dcgan_step.py tests/test_lsun128.py -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -4 $o/tests.log
for i in 1 2; do python bench.py --config cifar_dcgan_bf16 --steps 30 --warmup 5 --no-roofline > $o/dcgan_bf16_$i.json 2>/dev/null; head -c 220 $o/dcgan_bf16_$i.json; echo; done
python bench.py --config lsun128_f16 --steps 8 --warmup 2 --no-roofline > $o/lsun128_f16.json 2>/dev/null; head -c 220 $o/lsun128_f16.json; echo
python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > $o/resnet.json 2>/dev/null; head -c 220 $o/resnet.json; echo
bash tools/prof_run.sh r5o_dcgan_bf16 --config cifar_dcgan_bf16 --steps 20 --warmup 5 > $o/prof.log 2>&1
