#!/bin/bash
# Another build of libieache.so whose blind-rotation kernels are compiled with extra flags (compiler-scheduling A/B):
#   scripts/build_alt.sh <name> <extra hipcc flags...>   ->  ie-ache_amd/csrc/build/alt_<name>/libieache.so
# Run a script against it with IEACHE_LIBRARY=<that path>.
set -e
cd "$(dirname "$0")/../ie-ache_amd/csrc"
NAME=$1; shift
make -s -j4 > /dev/null
D=build/alt_$NAME; mkdir -p $D
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=on "$@" -Rpass-analysis=kernel-resource-usage \
    -c blind_rotate_w64.hip -o $D/blind_rotate_w64.o 2> $D/remarks.txt
OBJS=$(ls build/*.o | grep -v blind_rotate_w64.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fopenmp -o $D/libieache.so $OBJS $D/blind_rotate_w64.o
grep -A 12 "Function Name: .*18k_blind_rotate_w1bILi3ELi7ELi2ELb0E" $D/remarks.txt | grep -E "VGPRs:|VGPRs Spill|ScratchSize" | sed 's/.*remark: //' | tr '\n' ' '; echo " <- k_blind_rotate_w1b<3,7,2> in $D"
