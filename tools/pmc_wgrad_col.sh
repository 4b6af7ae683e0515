#!/bin/bash
# Pipe counters and HBM-side traffic of the grouped weight-gradient launch on the critic step's job table (tools/wgrad_group_bench.py d):
# one rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass; --pmc never together with a sys/hip trace), the
# program directly after `--`.  Writes gpurun_out/pmc_wcol/<tag>_*.csv and a summary JSON (tools/pmc_wgrad_col_parse.py).
# usage: tools/pmc_wgrad_col.sh <tag> [d|g]      (env CTGAN_WGRAD16_COL etc. pass through)
tag=${1:-col}; which=${2:-d}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_wcol; mkdir -p $out
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp -d $out/raw_$name -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/wgrad_group_bench.py $which 3 > $out/run_${tag}_$name.log 2>&1
  f=$(find $out/raw_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $out/${tag}_$name.csv; else echo "($name: no data)"; tail -3 $out/run_${tag}_$name.log; fi
  f=$(find $out/raw_$name -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $out/${tag}_trace_$name.csv; fi
  rm -rf $out/raw_$name
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_wgrad_col_parse.py gpurun_out/pmc_wcol $tag | tee gpurun_out/pmc_wcol/${tag}_summary.json
