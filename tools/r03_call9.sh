#!/bin/bash
o=gpurun_out/r03i; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -q -k "f32x3" > $o/tests_k16.log 2>&1; echo "k16 tests rc=$?"; tail -6 $o/tests_k16.log
for m in 1 2 0; do
  CTGAN_X3_8X8=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_mode$m.json 2> $o/bench_mode$m.err; echo "mode $m:"; head -c 230 $o/bench_mode$m.json | tail -c 90; echo
done
CTGAN_X3_8X8=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_mode1b.json 2> /dev/null; echo "mode 1 again:"; head -c 230 $o/bench_mode1b.json | tail -c 90; echo
CTGAN_X3_8X8=2 python tools/conv16_bench.py f32x3 resnet 2>&1 | grep "8, 8," | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$14}'
