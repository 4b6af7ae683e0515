#!/bin/bash
o=gpurun_out/r03c; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels16.py -q -k "f32x3" > $o/tests_k16.log 2>&1; echo "k16 tests rc=$?"; tail -6 $o/tests_k16.log
CTGAN_X3_HALO_V=2 python tools/conv16_bench.py f32x3 resnet > $o/conv_bench_x3_v2.txt 2>&1; cat $o/conv_bench_x3_v2.txt
CTGAN_WGRAD_XCD=0 python tools/conv16_bench.py f32x3 resnet 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$12,$13}' > $o/conv_bench_x3_noxcd.txt; cat $o/conv_bench_x3_noxcd.txt
python tools/conv16_bench.py f32 resnet 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$12,$13}' > $o/conv_bench_f32_xcd.txt; cat $o/conv_bench_f32_xcd.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; head -c 300 $o/bench.json; echo
CTGAN_WGRAD_XCD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_noxcd.json 2> $o/bench_noxcd.err; head -c 300 $o/bench_noxcd.json; echo
# ordered launch lists of one critic step and of the generator step (steady state)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$o/raw -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$o/prof_bench.json 2> $GRAFT_REPO_ROOT/$o/prof_bench.err
cd $GRAFT_REPO_ROOT
db=$(find $o/raw -name '*.db' | head -1)
python tools/prof_seq.py $db adam_kernel 1 0 > $o/seq_d_step.txt 2>&1
python tools/prof_seq.py $db adam_kernel 1 5 > $o/seq_g_step.txt 2>&1
python tools/prof_gaps.py $db > $o/steady_state.txt 2>&1
rm -rf $o/raw
head -5 $o/steady_state.txt
