#!/bin/bash
# Copy the summaries of one tools/round_end_call.sh run (gpurun_out/<tag>, gpurun_out/prof_<tag>*, gpurun_out/phase_<tag>, gpurun_out/pmc_*) into
# profiles/r06_* (the tracked, judged copies).  usage: tools/collect_profiles.sh <tag>
set -e
tag=${1:?tag}; o=gpurun_out/$tag; p=profiles; r=r06
cp $o/tests_all.log $p/${r}_gpu_tests.log
cp $o/bench.json $p/${r}_bench_resnet.json
cp $o/bench_defaults.json $p/${r}_bench_resnet_defaults.json
cp $o/bench_driver_args.json $p/${r}_bench_resnet_driver_args.json
for v in WGRAD_OVERLAP PREP_ASYNC CHAIN8X8 X3_HK0; do cp $o/bench_$v.json $p/${r}_bench_resnet_$v.json; done
cp gpurun_out/hk_prof.txt $p/${r}_hk_prof.txt
cp gpurun_out/chain_probe.txt $p/${r}_chain_kernel_times.txt
cp gpurun_out/winograd_probe.txt $p/${r}_winograd_probe.txt; cp gpurun_out/mfma_loop_probe.txt $p/${r}_mfma_loop_probe.txt
cp $o/bench_autograd_critic.json $p/${r}_bench_resnet_autograd_critic_step.json
cp $o/bench_2rank_gloo.json $p/${r}_bench_resnet_2rank_gloo_one_gpu.json
cp $o/gp_unit.json $p/${r}_gp_unit.json
cp $o/phase_times.txt $p/${r}_phase_times.txt
cp gpurun_out/prof_gp_unit/kernel_stats.txt $p/${r}_gp_unit_kernel_stats.txt
cp $o/crosscheck.txt $p/${r}_roofline_crosscheck.txt
cp gpurun_out/prof_$tag/kernel_stats.txt $p/${r}_kernel_stats_resnet.txt
cp gpurun_out/prof_$tag/steady_state.txt $p/${r}_steady_state_resnet.txt
cp gpurun_out/prof_$tag/bench.json $p/${r}_bench_resnet_under_rocprof.json
for ph in g f d; do cp gpurun_out/phase_$tag/${ph}_steady.txt $p/${r}_phase_${ph}_kernels.txt; done
for c in dcgan_bf16 lsun128_f16; do
  cp gpurun_out/prof_${tag}_$c/kernel_stats.txt $p/${r}_kernel_stats_$c.txt
  cp gpurun_out/prof_${tag}_$c/steady_state.txt $p/${r}_steady_state_$c.txt
done
cp $o/bench_dcgan_bf16.json $p/${r}_bench_cifar_dcgan_bf16.json
cp $o/bench_dcgan_f32.json $p/${r}_bench_cifar_dcgan_f32.json
cp $o/bench_lsun128_f16.json $p/${r}_bench_lsun128_f16.json
cp $o/bench_lsun128_f32.json $p/${r}_bench_lsun128_f32.json
cp $o/wgrad_group_bench_col.txt $p/${r}_wgrad_group_bench_col.txt
cp gpurun_out/fewch_prof.txt $p/${r}_fewch_kernel_times.txt
cp gpurun_out/pmc_x3/r06_pmc_traffic_x3.json $p/${r}_pmc_traffic_x3.json
cp gpurun_out/pmc_wcol/col_summary.json $p/${r}_pmc_wgrad_col.json
cp $o/bench_2rank_gloo.err $p/${r}_bench_resnet_2rank_gloo_one_gpu_legs.txt 2>/dev/null || true
for f in lsun128_f16_B64_vs_fixture.json cifar_dcgan_B64_vs_fixture.json lsun128_f16_B64_gstep_vs_fixture.json cifar_dcgan_B64_gstep_vs_fixture.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f $p/${r}_$f; done
ls $p | grep -c "^${r}_"
