#!/bin/bash
# round-3 baseline call: bench (default + launcher paths), GP-unit profile, steady-state profile, PMC passes, new / changed GPU tests
o=gpurun_out/r03a; mkdir -p $o
python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $o/bench_2rank_gloo.json 2> $o/bench_2rank_gloo.err; echo "2rank rc=$?"
python bench.py --gpus 1 --launcher always --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > $o/bench_spawn1.json 2> $o/bench_spawn1.err; echo "spawn1 rc=$?"
python bench.py --gpus 2 > $o/bench_refuse.json 2> $o/bench_refuse.err; echo "refuse rc=$? (expect 2)"
bash tools/prof_gp_unit.sh > $o/gp_unit.log 2>&1
bash tools/prof_run.sh r03a --steps 20 --warmup 5 > $o/prof_run.log 2>&1
bash tools/pmc_x3.sh > $o/pmc_x3.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "sample_tensors or full_width_16bit or 64x64_on_gpu or graph_replay_loop_equals or teacher_forced or whole_iteration_graph or resumes_bit_exactly" > $o/tests.log 2>&1; echo "tests rc=$?"
tail -5 $o/tests.log
head -c 600 $o/bench_default.json; echo; cat $o/bench_2rank_gloo.json | head -c 900; echo; head -c 400 $o/bench_spawn1.json; echo; cat $o/bench_refuse.err | tail -2
