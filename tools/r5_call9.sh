#!/bin/bash
# round 5, call 9: conv16x3sf_kernel (strided forward, filter fragments from L2)
o=gpurun_out/r5q; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -m gpu -x -q -k "strided_forward or split_mode_is_as_accurate or batched_filter_pack" > $o/tests_k16.log 2>&1; echo "k16 rc=$?"; tail -5 $o/tests_k16.log
python tools/conv16_bench.py f32x3 s2 2>&1 | grep -v amdgpu | cut -c1-100 > $o/s2_new.txt; cat $o/s2_new.txt
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > $o/resnet_$i.json 2>/dev/null; head -c 200 $o/resnet_$i.json; echo; done
CTGAN_X3_S2FWD=0 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > $o/resnet_slice.json 2>/dev/null; head -c 200 $o/resnet_slice.json; echo
