#!/bin/bash
o=gpurun_out/r5w; mkdir -p $o
python tools/sf_dgrad_check.py 2>&1 | grep -v amdgpu | tee $o/sf_dgrad.txt
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline > $o/resnet_$i.json 2>/dev/null; head -c 200 $o/resnet_$i.json; echo; done
for i in 1 2; do CTGAN_X3_S2DGRAD_SF=1 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline > $o/resnet_sfd_$i.json 2>/dev/null; head -c 200 $o/resnet_sfd_$i.json; echo; done
