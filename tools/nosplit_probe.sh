#!/bin/bash
# Upper bound of "split operands stored by the producer" (VERDICT r4 #2): build the library with the in-kernel fp32 -> 3 x bf16 split replaced
# by one pack (-DCTGAN_SPLIT_NONE on igemm16.hip and wgrad16c.hip: results wrong by design) and time the split-mode kernels in both builds.
# step 1 (build container): tools/nosplit_probe.sh build      step 2 (GPU box): tools/nosplit_probe.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd ctgan_amd/csrc; make -j8 > /dev/null
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -I../../include -DCTGAN_SPLIT_NONE"
  /opt/rocm/bin/hipcc $F -c igemm16.hip -o /tmp/nosplit_igemm16.o &
  /opt/rocm/bin/hipcc $F -c wgrad16c.hip -o /tmp/nosplit_wgrad16c.o &
  wait
  objs=$(ls *.o | grep -v "^igemm16.o$" | grep -v "^wgrad16c.o$" | tr '\n' ' ')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/nosplit_igemm16.o /tmp/nosplit_wgrad16c.o -o ../libctgan_hip_nosplit.so
  echo built ctgan_amd/libctgan_hip_nosplit.so
  exit 0
fi
o=gpurun_out/nosplit; mkdir -p $o
for v in default nosplit; do
  if [ $v = nosplit ]; then export CTGAN_LIB=$PWD/ctgan_amd/libctgan_hip_nosplit.so; fi
  python tools/conv16_bench.py f32x3 s2 > $o/s2_$v.txt 2>&1
  python tools/conv16_bench.py f32x3 resnet > $o/resnet_$v.txt 2>&1
  timeout 120 python tools/wgrad_group_bench.py d 40 > $o/wgrad_$v.txt 2>&1
done
for f in s2 resnet wgrad; do echo "== $f"; paste -d'\n' $o/${f}_default.txt $o/${f}_nosplit.txt | grep -v amdgpu.ids | cut -c1-150; done
