#!/bin/bash
# One gpurun call that reproduces the round's evidence: full GPU suite, the headline bench (+ launcher / multi-rank-on-one-GPU paths, the
# other BASELINE configs), rocprofv3 summaries, the GP unit, PMC passes.  usage: gpurun -- 'bash tools/round_end_call.sh <tag>'
tag=${1:-final}; o=gpurun_out/$tag; mkdir -p $o
timeout 2700 python -m pytest tests -m gpu -q > $o/tests_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -6 $o/tests_all.log
python bench.py --steps 30 --warmup 5 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 300 $o/bench.json; echo
python bench.py > $o/bench_defaults.json 2> $o/bench_defaults.err; echo "bench (default 100/20) rc=$?"; head -c 300 $o/bench_defaults.json; echo
python bench.py --gp-unit-only > $o/gp_unit.json 2> $o/gp_unit.err
bash tools/prof_run.sh $tag --steps 20 --warmup 5 > $o/prof_run.log 2>&1
bash tools/prof_gp_unit.sh > $o/gp_unit_prof.log 2>&1
bash tools/pmc_x3.sh > $o/pmc_x3.log 2>&1
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $o/bench_2rank_gloo.json 2> $o/bench_2rank_gloo.err; echo "2rank rc=$?"
python bench.py --gpus 1 --launcher always --steps 30 --warmup 5 --no-roofline --no-cpu-baseline > $o/bench_launcher_1rank.json 2> $o/bench_launcher_1rank.err; echo "launcher rc=$?"
python bench.py --config lsun128_f16 --steps 5 --warmup 2 > $o/bench_lsun128_f16.json 2> $o/bench_lsun128_f16.err; echo "lsun rc=$?"; head -c 200 $o/bench_lsun128_f16.json; echo
python bench.py --config cifar_dcgan_bf16 --steps 20 --warmup 5 > $o/bench_dcgan_bf16.json 2> $o/bench_dcgan_bf16.err; echo "dcgan rc=$?"; head -c 200 $o/bench_dcgan_bf16.json; echo
python tools/roofline_crosscheck.py $o/bench.json gpurun_out/prof_$tag/steady_state.txt > $o/crosscheck.txt 2>&1; tail -3 $o/crosscheck.txt
