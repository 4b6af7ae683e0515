#!/bin/bash
# PMC passes (separate runs per counter group, kernel-trace only) of one 16-bit conv launch: L2 hit rate and HBM-side bytes.
#   tools/pmc_run16.sh <tag> <pmc_conv16.py args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag; mkdir -p $out
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp -d $out/raw_$name -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_conv16.py "$@" > $out/run_$name.log 2>&1
  f=$(find $out/raw_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$name" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get('Kernel_Name', '')
    if '16_kernel' not in k and 'igemm' not in k: continue
    agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    for c, v in d.items():
        print('%-62s %-32s n=%d mean=%.4g' % (k, c, len(v), sum(v) / len(v)))
PY
  fi
  rm -rf $out/raw_$name
done
