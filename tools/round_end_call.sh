#!/bin/bash
# One gpurun call that reproduces the round's evidence: full GPU suite (with durations), the headline bench (+ A/B of the hand-scheduled
# critic step, launcher / multi-rank-on-one-GPU paths, the other BASELINE configs), rocprofv3 summaries (iteration + per phase), the GP unit,
# PMC passes.  usage: gpurun -- 'bash tools/round_end_call.sh <tag>'
tag=${1:-final}; o=gpurun_out/$tag; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $o/tests_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -22 $o/tests_all.log
python bench.py --steps 30 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 300 $o/bench.json; echo
python bench.py > $o/bench_defaults.json 2> $o/bench_defaults.err; echo "bench (default 100/20) rc=$?"; head -c 300 $o/bench_defaults.json; echo
CTGAN_MERGED_BWD=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $o/bench_autograd_critic.json 2> /dev/null; head -c 200 $o/bench_autograd_critic.json; echo
python bench.py --steps 20 --warmup 5 > $o/bench_driver_args.json 2> /dev/null; echo "bench (driver's 20/5) rc=$?"
for v in WGRAD_OVERLAP PREP_ASYNC CHAIN8X8; do env CTGAN_$v=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --feed device > $o/bench_$v.json 2> /dev/null; done
CTGAN_X3_HK=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --feed device > $o/bench_X3_HK0.json 2> /dev/null
bash tools/hk_prof.sh > $o/hk_prof.log 2>&1
bash tools/chain_probe.sh > $o/chain_probe.log 2>&1
(hipcc --offload-arch=gfx950 -O3 tools/winograd_probe.hip -o /tmp/wp 2>/dev/null && timeout 120 /tmp/wp > gpurun_out/winograd_probe.txt; hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip -o /tmp/mp 2>/dev/null && timeout 120 /tmp/mp > gpurun_out/mfma_loop_probe.txt)
python bench.py --gp-unit-only > $o/gp_unit.json 2> $o/gp_unit.err
python tools/phase_times.py 2>/dev/null | grep -v amdgpu.ids > $o/phase_times.txt; cat $o/phase_times.txt
bash tools/prof_run.sh $tag --steps 20 --warmup 5 > $o/prof_run.log 2>&1
bash tools/phase_prof.sh $tag > $o/phase_prof.log 2>&1
bash tools/prof_gp_unit.sh > $o/gp_unit_prof.log 2>&1
bash tools/pmc_x3.sh > $o/pmc_x3.log 2>&1
bash tools/pmc_wgrad_col.sh col d > $o/pmc_wgrad_col.log 2>&1
timeout 120 python tools/wgrad_group_bench.py both 40 > $o/wgrad_group_bench_col.txt 2>&1
bash tools/fewch_prof.sh > $o/fewch_prof.log 2>&1
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --feed device > $o/bench_2rank_gloo.json 2> $o/bench_2rank_gloo.err; grep '^bench: leg' $o/bench_2rank_gloo.err; echo "2rank rc=$?"
python bench.py --config lsun128_f16 --steps 8 --warmup 2 > $o/bench_lsun128_f16.json 2> $o/bench_lsun128_f16.err; echo "lsun rc=$?"; head -c 200 $o/bench_lsun128_f16.json; echo
python bench.py --config cifar_dcgan_bf16 --steps 20 --warmup 5 > $o/bench_dcgan_bf16.json 2> $o/bench_dcgan_bf16.err; echo "dcgan rc=$?"; head -c 200 $o/bench_dcgan_bf16.json; echo
python bench.py --config cifar_dcgan_f32 --steps 20 --warmup 5 --no-roofline > $o/bench_dcgan_f32.json 2> /dev/null
python bench.py --config lsun128_f32 --steps 5 --warmup 2 --no-roofline > $o/bench_lsun128_f32.json 2> /dev/null
bash tools/prof_run.sh ${tag}_dcgan_bf16 --config cifar_dcgan_bf16 --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/prof_run.sh ${tag}_lsun128_f16 --config lsun128_f16 --steps 6 --warmup 2 > /dev/null 2>&1
python tools/roofline_crosscheck.py $o/bench.json gpurun_out/prof_$tag/steady_state.txt > $o/crosscheck.txt 2>&1; tail -3 $o/crosscheck.txt
