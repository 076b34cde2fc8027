#!/bin/bash
# Build an A/B variant of one kernel translation unit: tools/build_variant.sh <tag> "<extra hipcc flags>" [unit]
# -> smol_amd/exp/libsmolmc_<tag>.so (load with SMOLMC_LIB=...).  Only <unit>.hip (default lean_n2) is
# rebuilt; the other objects are reused from the normal build (run make first).
set -e
tag=$1; flags=$2; unit=${3:-lean_n2}
cd "$(dirname "$0")/../smol_amd/csrc"
mkdir -p ../exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flags -c -o /tmp/${unit}_$tag.o $unit.hip
others=$(ls *.o | grep -v "^$unit.o\$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libsmolmc_$tag.so $others /tmp/${unit}_$tag.o
echo built smol_amd/exp/libsmolmc_$tag.so
