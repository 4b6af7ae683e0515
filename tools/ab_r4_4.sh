mkdir -p gpurun_out/r4
for v in "" _wcdbg16 _wcdbg2 _wcdbg10 _wcdbg8 _wcdbg4; do
  for pp in 0 1; do
  CTGAN_LIB=$PWD/ctgan_amd/libctgan_hip$v.so CTGAN_WGRAD16_COL_PP=$pp timeout 120 python tools/wgrad_group_bench.py d 40 2>&1 | grep step | sed "s/^/lib=$v pp=$pp /" | cut -c1-30,105-260
  done
done 2>&1 | tee gpurun_out/r4/ab4.log
CTGAN_LIB=$PWD/ctgan_amd/libctgan_hip_wcdbg16.so timeout 600 python -m pytest tests/test_gpu_wgrad_col.py -x -q 2>&1 | tail -3
