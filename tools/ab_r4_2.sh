mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_wgrad_col.py -x -q 2>&1 | tail -3
for pp in 0 1; do for ch in 0 1536; do
  if [ $ch = 0 ]; then unset CTGAN_WGRAD16_COL_CHUNK; else export CTGAN_WGRAD16_COL_CHUNK=$ch; fi
  CTGAN_WGRAD16_COL_PP=$pp timeout 120 python tools/wgrad_group_bench.py both 40 2>&1 | grep step | sed "s/^/pp=$pp chunk=$ch /" | cut -c1-30,100-260
done; done 2>&1 | tee gpurun_out/r4/ab2.log
unset CTGAN_WGRAD16_COL_CHUNK
timeout 600 tools/pmc_wgrad_col.sh col2 d > gpurun_out/r4/pmc_col2.log 2>&1
CTGAN_WGRAD16_COL_PP=1 timeout 600 tools/pmc_wgrad_col.sh col2pp d > gpurun_out/r4/pmc_col2pp.log 2>&1
python - <<'PY'
import json
for t in ('col2','col2pp'):
    d=json.load(open('gpurun_out/pmc_wcol/%s_summary.json'%t))
    for k,v in d.items():
        if 'wgrad' in k:
            v=dict(v); v.pop('raw_means'); print(t,k,json.dumps(v))
PY
