#!/bin/bash
# Build an A/B variant of the lean kernels: tools/build_variant.sh <tag> "<extra hipcc flags>"
# -> smol_amd/exp/libsmolmc_<tag>.so (load with SMOLMC_LIB=...).  Only lean_n2 is rebuilt; the
# other objects are reused from the normal build (run make first).
set -e
tag=$1; flags=$2
cd "$(dirname "$0")/../smol_amd/csrc"
mkdir -p ../exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flags -c -o /tmp/lean_n2_$tag.o lean_n2.hip
others=$(ls *.o | grep -v '^lean_n2.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libsmolmc_$tag.so $others /tmp/lean_n2_$tag.o
echo built smol_amd/exp/libsmolmc_$tag.so
