#!/bin/bash
o=gpurun_out/r5p; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_dcgan_step.py -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -4 $o/tests.log
for i in 1 2; do python bench.py --config cifar_dcgan_bf16 --steps 30 --warmup 5 --no-roofline > $o/dcgan_bf16_$i.json 2>/dev/null; head -c 220 $o/dcgan_bf16_$i.json; echo; done
bash tools/prof_run.sh r5p_dcgan_bf16 --config cifar_dcgan_bf16 --steps 20 --warmup 5 > $o/prof.log 2>&1
grep -n "wgrad\|reduce" gpurun_out/prof_r5p_dcgan_bf16/steady_state.txt | cut -c1-150
