mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_wgrad_col.py -x -q 2>&1 | tail -2
for v in "" _wcdbg4; do
  for fl in 0 2 1 3; do
  CTGAN_LIB=$PWD/ctgan_amd/libctgan_hip$v.so CTGAN_WGRAD16_COL_PP=$fl timeout 120 python tools/wgrad_group_bench.py both 40 2>&1 | grep step | sed "s/^/lib=$v flags=$fl /" | cut -c1-32,105-260
  done
done 2>&1 | tee gpurun_out/r4/ab5.log
timeout 600 tools/pmc_wgrad_col.sh col5 d > gpurun_out/r4/pmc_col5.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_wcol/col5_summary.json'))
for k,v in d.items():
    if 'wgrad' in k:
        v=dict(v); rm=v.pop('raw_means'); print(k,json.dumps(v))
PY
