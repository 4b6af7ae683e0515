#!/bin/bash
# second PMC set: where do the waves of a 16-bit conv kernel wait (separate passes per group; unknown counters just fail their pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag; mkdir -p $out
for grp in "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp -d $out/raw_$name -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_conv16.py "$@" > $out/run_$name.log 2>&1
  f=$(find $out/raw_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    k = r.get('Kernel_Name', '')
    if 'conv16' not in k and 'wgrad16' not in k: continue
    agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in agg.items():
    print('%-40s n=%d mean=%.5g' % (c, len(v), sum(v) / len(v)))
PY
  else echo "($name: no data)"; fi
  rm -rf $out/raw_$name
done
