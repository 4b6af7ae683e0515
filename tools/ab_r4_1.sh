mkdir -p gpurun_out/r4
for pp in 0 1; do for ch in 0 512 1024 1536 2048 3072; do
  if [ $ch = 0 ]; then unset CTGAN_WGRAD16_COL_CHUNK; else export CTGAN_WGRAD16_COL_CHUNK=$ch; fi
  CTGAN_WGRAD16_COL_PP=$pp timeout 120 python tools/wgrad_group_bench.py both 30 2>&1 | grep step | sed "s/^/pp=$pp chunk=$ch /" | cut -c1-260
done; done 2>&1 | tee gpurun_out/r4/ab1.log
unset CTGAN_WGRAD16_COL_CHUNK
timeout 600 tools/pmc_wgrad_col.sh col d > gpurun_out/r4/pmc_col.log 2>&1
CTGAN_WGRAD16_COL_PP=1 timeout 600 tools/pmc_wgrad_col.sh colpp d > gpurun_out/r4/pmc_colpp.log 2>&1
tail -5 gpurun_out/r4/pmc_col.log
