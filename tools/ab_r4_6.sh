mkdir -p gpurun_out/r4
for v in "" _wcdbg10 _wcdbg32 _wcdbg64 _wcdbg128 _wcdbg234; do
  for fl in 0 1; do
  CTGAN_LIB=$PWD/ctgan_amd/libctgan_hip$v.so CTGAN_WGRAD16_COL_PP=$fl timeout 120 python tools/wgrad_group_bench.py d 40 2>&1 | grep step | sed "s/^/lib=$v flags=$fl /" | cut -c1-32,105-260
  done
done 2>&1 | tee gpurun_out/r4/ab6.log
