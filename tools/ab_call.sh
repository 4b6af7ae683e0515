#!/bin/bash
# A/B of two trees on ONE box (box-to-box variance is +-3 %): usage: gpurun -- 'bash tools/ab_call.sh <tag> <other tree dir> [bench args]'
tag=$1; other=$2; shift; shift
o=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $o
for r in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/$other && python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'])") | tee -a $o/ab.txt
  (cd $GRAFT_REPO_ROOT && python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])") | tee -a $o/ab.txt
done
