#!/bin/bash
# short PMC set for the split-mode kernels: MFMA pipe occupancy and where the waves wait (one pass per pair of counters)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag; mkdir -p $out
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
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
