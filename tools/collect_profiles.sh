#!/bin/bash
# Copy the summaries of one tools/round_end_call.sh run (gpurun_out/<tag>, gpurun_out/prof_<tag>*, gpurun_out/pmc_wcol) into profiles/r04_*
# (the tracked, judged copies).  usage: tools/collect_profiles.sh <tag>
set -e
tag=${1:?tag}; o=gpurun_out/$tag; p=profiles; r=r04
cp $o/tests_all.log $p/${r}_gpu_tests.log
cp $o/bench.json $p/${r}_bench_resnet.json
cp $o/bench_defaults.json $p/${r}_bench_resnet_defaults.json
cp $o/bench_2rank_gloo.json $p/${r}_bench_resnet_2rank_gloo_one_gpu.json
cp $o/gp_unit.json $p/${r}_gp_unit.json
cp gpurun_out/prof_gp_unit/kernel_stats.txt $p/${r}_gp_unit_kernel_stats.txt
cp $o/crosscheck.txt $p/${r}_roofline_crosscheck.txt
cp gpurun_out/prof_$tag/kernel_stats.txt $p/${r}_kernel_stats_resnet.txt
cp gpurun_out/prof_$tag/steady_state.txt $p/${r}_steady_state_resnet.txt
cp gpurun_out/prof_$tag/bench.json $p/${r}_bench_resnet_under_rocprof.json
for c in dcgan_bf16 lsun128_f16; do
  cp gpurun_out/prof_${tag}_$c/kernel_stats.txt $p/${r}_kernel_stats_$c.txt
  cp gpurun_out/prof_${tag}_$c/steady_state.txt $p/${r}_steady_state_$c.txt
done
cp $o/bench_dcgan_bf16.json $p/${r}_bench_cifar_dcgan_bf16.json
cp $o/bench_dcgan_bf16_round3_switches.json $p/${r}_bench_cifar_dcgan_bf16_round3_switches.json
cp $o/bench_dcgan_f32.json $p/${r}_bench_cifar_dcgan_f32.json
cp $o/bench_lsun128_f16.json $p/${r}_bench_lsun128_f16.json
cp $o/bench_lsun128_f16_slice_wgrad.json $p/${r}_bench_lsun128_f16_slice_wgrad.json
cp $o/bench_lsun128_f32.json $p/${r}_bench_lsun128_f32.json
cp $o/wgrad_group_bench_col.txt $p/${r}_wgrad_group_bench_col.txt
cp $o/wgrad_group_bench_slice.txt $p/${r}_wgrad_group_bench_slice.txt
cp $o/wgrad_group_bench_col_splitmajor.txt $p/${r}_wgrad_group_bench_col_splitmajor.txt
python3 - <<PY
import json
out = {}
for t in ('col', 'slice'):
    out.update(json.load(open('gpurun_out/pmc_wcol/%s_summary.json' % t)))
json.dump(out, open('$p/${r}_pmc_wgrad_col.json', 'w'), indent=1)
print(sorted(out))
PY
ls $p | grep -c "^${r}_"
