timeout 900 python -m pytest tests/test_gpu_wgrad_col.py tests/test_gpu_kernels16.py tests/test_gpu_dcgan_step.py tests/test_lsun128.py -m gpu -q --durations=3 2>&1 | tail -14
for cfg in cifar_dcgan_bf16 lsun128_f16; do
python bench.py --config $cfg --steps 10 --warmup 3 > gpurun_out/b_$cfg.json 2> gpurun_out/b_$cfg.err; echo "$cfg rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/b_$cfg.json')); print(r['value'], r['ms_per_step'], r['config'].get('last_d_terms'), r['roofline']['kernel'], r['roofline']['frac'])"
CTGAN_WGRAD16_COL=0 python bench.py --config $cfg --steps 10 --warmup 3 --no-roofline > gpurun_out/b2_$cfg.json 2> gpurun_out/b2_$cfg.err; echo "$cfg COL=0 rc=$?"; python -c "
import json; r=json.load(open('gpurun_out/b2_$cfg.json')); print(r['value'], r['ms_per_step'])"
done
