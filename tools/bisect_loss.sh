#!/bin/bash
# Runs tools/loss_trace.py under a matrix of fusion switches; one log per configuration under gpurun_out/bisect/.
out=gpurun_out/bisect; mkdir -p $out
run() { name=$1; shift; echo "== $name"; ( env "$@" timeout 300 python tools/loss_trace.py --tag $name $EXTRA > $out/$name.log 2>&1 ); tail -n 3 $out/$name.log | cut -c1-250; }
run default X=1
EXTRA=--no-graph run eager X=1
run tail0 CTGAN_TAIL_SHARE=0
run trunk0 CTGAN_TRUNK_SHARE=0
run fewch0 CTGAN_FEWCH_DEFER=0
run head0 CTGAN_HEAD_FUSION=0
run grouped0 CTGAN_WGRAD_GROUPED=0
run defer0 CTGAN_DEFER_WGRADS=0
run prep0 CTGAN_PREP_FUSION=0
run batchfakes0 CTGAN_BATCH_FAKES=0
run alloff CTGAN_TAIL_SHARE=0 CTGAN_TRUNK_SHARE=0 CTGAN_FEWCH_DEFER=0 CTGAN_HEAD_FUSION=0 CTGAN_WGRAD_GROUPED=0 CTGAN_PREP_FUSION=0 CTGAN_DROP_FUSION=0
