#!/bin/bash
# round 5, call 5: the LeakyReLU + dropout pair in the 16-bit slice kernels' epilogue - bit test, the DCGAN step tests, config[1] A/B
o=gpurun_out/r5i; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -m gpu -x -q -k "fused_lrelu_dropout or conv16_fwd_dgrad" > $o/tests_k16.log 2>&1; echo "k16 rc=$?"; tail -5 $o/tests_k16.log
timeout 900 python -m pytest tests/test_gpu_dcgan_step.py -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -5 $o/tests.log
for i in 1 2; do python bench.py --config cifar_dcgan_bf16 --steps 30 --warmup 5 --no-roofline > $o/dcgan_bf16_act_$i.json 2>/dev/null; head -c 220 $o/dcgan_bf16_act_$i.json; echo; done
CTGAN_ACT_EPILOGUE=0 python bench.py --config cifar_dcgan_bf16 --steps 30 --warmup 5 --no-roofline > $o/dcgan_bf16_noact.json 2>/dev/null; head -c 220 $o/dcgan_bf16_noact.json; echo
python bench.py --steps 20 --warmup 5 --no-roofline > $o/resnet.json 2>/dev/null; head -c 220 $o/resnet.json; echo
