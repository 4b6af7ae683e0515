#!/bin/bash
o=gpurun_out/r5r; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_kernels16.py tests/test_gpu_resnet_step.py tests/test_gpu_graph_loop.py tests/test_lsun128.py tests/test_gan_64x64.py -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc=$?"; tail -4 $o/tests.log
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline > $o/resnet_$i.json 2>/dev/null; head -c 200 $o/resnet_$i.json; echo; done
CTGAN_X3_S2FWD=0 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline > $o/resnet_slice.json 2>/dev/null; head -c 200 $o/resnet_slice.json; echo
python bench.py --config lsun128_f32 --steps 5 --warmup 2 --no-roofline > $o/lsun128_f32.json 2>/dev/null; head -c 200 $o/lsun128_f32.json; echo
python bench.py --config cifar_dcgan_f32 --steps 20 --warmup 5 --no-roofline > $o/dcgan_f32.json 2>/dev/null; head -c 200 $o/dcgan_f32.json; echo
python tools/phase_times.py 2>/dev/null | grep -v amdgpu.ids
bash tools/phase_prof.sh r5r > $o/phase.log 2>&1
