#!/bin/bash
# A/B of two trees (and of env settings of this tree) on ONE box (box-to-box variance is +-3 %).
#   usage: gpurun -- 'bash tools/ab_call.sh <tag> <other tree dir or -> ["ENV=val ..." ...]'
tag=$1; other=$2; shift; shift
o=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $o
run() { (cd $1 && env $3 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])") | tee -a $o/ab.txt; }
for r in 1 2 3; do
  [ "$other" != "-" ] && run $GRAFT_REPO_ROOT/$other old ""
  run $GRAFT_REPO_ROOT new ""
  for e in "$@"; do run $GRAFT_REPO_ROOT "new[$e]" "$e"; done
done
