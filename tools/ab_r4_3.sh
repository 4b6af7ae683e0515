mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_wgrad_col.py -x -q 2>&1 | tail -3
for pp in 0 1; do
  CTGAN_WGRAD16_COL_PP=$pp timeout 120 python tools/wgrad_group_bench.py both 40 2>&1 | grep step | sed "s/^/pp=$pp /" | cut -c1-20,95-260
done 2>&1 | tee gpurun_out/r4/ab3.log
timeout 600 tools/pmc_wgrad_col.sh col3 d > gpurun_out/r4/pmc_col3.log 2>&1
CTGAN_WGRAD16_COL_PP=1 timeout 600 tools/pmc_wgrad_col.sh col3pp d > gpurun_out/r4/pmc_col3pp.log 2>&1
python - <<'PY'
import json
for t in ('col3','col3pp'):
    d=json.load(open('gpurun_out/pmc_wcol/%s_summary.json'%t))
    for k,v in d.items():
        if 'wgrad' in k:
            v=dict(v); rm=v.pop('raw_means'); print(t,k,json.dumps(v)); print({a:round(b) for a,b in rm.items() if 'INSTS' in a})
PY
