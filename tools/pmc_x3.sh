#!/bin/bash
# HBM-side traffic and pipe counters of the split-mode forward / data-gradient kernels the headline's time is in (the grouped weight
# gradient has its own script: tools/pmc_wgrad_col.sh).  One rocprofv3 pass per
# counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass; --pmc never together with a sys/hip trace), the program directly
# after `--`.  Writes gpurun_out/pmc_x3/{counters_*.csv,info.json}; tools/pmc_x3_parse.py turns them into r06_pmc_traffic_x3.json (copied to profiles/).
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_x3; mkdir -p $out
python3 $GRAFT_REPO_ROOT/tools/pmc_x3_run.py 2 2>/dev/null | grep '^{' | tail -1 > $out/info.json
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp -d $out/raw_$name -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_x3_run.py 6 > $out/run_$name.log 2>&1
  f=$(find $out/raw_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $out/counters_$name.csv; else echo "($name: no data)"; tail -3 $out/run_$name.log; fi
  rm -rf $out/raw_$name
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_x3_parse.py gpurun_out/pmc_x3 > gpurun_out/pmc_x3/r06_pmc_traffic_x3.json && head -c 1500 gpurun_out/pmc_x3/r06_pmc_traffic_x3.json
