#!/bin/bash
# round 5, call 6: band im2col / per-pixel col2im, channels-last features in the DCGAN schedule, no fp32 repacked filters in the 16-bit modes
o=gpurun_out/r5j; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "im2col" > $o/tests_k.log 2>&1; echo "k rc=$?"; tail -5 $o/tests_k.log
timeout 900 python -m pytest tests/test_gpu_dcgan_step.py tests/test_gan_64x64.py tests/test_lsun128.py -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -5 $o/tests.log
for i in 1 2; do python bench.py --config cifar_dcgan_bf16 --steps 30 --warmup 5 --no-roofline > $o/dcgan_bf16_$i.json 2>/dev/null; head -c 220 $o/dcgan_bf16_$i.json; echo; done
python bench.py --config cifar_dcgan_f32 --steps 30 --warmup 5 --no-roofline > $o/dcgan_f32.json 2>/dev/null; head -c 220 $o/dcgan_f32.json; echo
python bench.py --config lsun128_f16 --steps 8 --warmup 2 --no-roofline > $o/lsun128_f16.json 2>/dev/null; head -c 220 $o/lsun128_f16.json; echo
bash tools/prof_run.sh r5j_dcgan_bf16 --config cifar_dcgan_bf16 --steps 20 --warmup 5 > $o/prof.log 2>&1
