#!/bin/bash
o=gpurun_out/r5h; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_dcgan_step.py tests/test_gan_64x64.py -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -5 $o/tests.log
for i in 1 2; do python bench.py --config cifar_dcgan_bf16 --steps 30 --warmup 5 --no-roofline > $o/dcgan_bf16_sched_$i.json 2>/dev/null; head -c 220 $o/dcgan_bf16_sched_$i.json; echo; done
CTGAN_DCGAN_MERGED_BWD=0 python bench.py --config cifar_dcgan_bf16 --steps 30 --warmup 5 --no-roofline > $o/dcgan_bf16_autograd.json 2>/dev/null; head -c 220 $o/dcgan_bf16_autograd.json; echo
python bench.py --config cifar_dcgan_f32 --steps 30 --warmup 5 --no-roofline > $o/dcgan_f32_sched.json 2>/dev/null; head -c 220 $o/dcgan_f32_sched.json; echo
CTGAN_DCGAN_MERGED_BWD=0 python bench.py --config cifar_dcgan_f32 --steps 30 --warmup 5 --no-roofline > $o/dcgan_f32_autograd.json 2>/dev/null; head -c 220 $o/dcgan_f32_autograd.json; echo
