#!/bin/bash
o=gpurun_out/r03h; mkdir -p $o
for t in 32 64 128; do
  echo "== CTGAN_X3_HF_TILE=$t SQ64=0"; CTGAN_X3_HF_SQ64=0 CTGAN_X3_HF_TILE=$t CTGAN_X3_HALO=2 python tools/conv16_bench.py f32x3 resnet 2>&1 | grep ", 3, 1)" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$14}' | tee $o/sweep_tile$t.txt
done
echo "== default"; python tools/conv16_bench.py f32x3 resnet 2>&1 | grep ", 3, 1)" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$14}' | tee $o/sweep_default.txt
