#!/bin/bash
# A/B build of the library with extra defines on ONE source: tools/build_variant.sh <name> <source.hip> "<-Dfoo=1 ...>"
# -> ctgan_amd/libctgan_hip_<name>.so (load with CTGAN_LIB=...; tools only).  The other objects come from the normal build.
set -e
name=$1; src=$2; defs=$3
cd "$(dirname "$0")/../ctgan_amd/csrc"
make -j8 >/dev/null
obj=/tmp/variant_${name}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -I../../include $defs -c $src -o $obj
objs=$(ls *.o | grep -v "^${src%.hip}.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $obj -o ../libctgan_hip_${name}.so
echo built ctgan_amd/libctgan_hip_${name}.so
