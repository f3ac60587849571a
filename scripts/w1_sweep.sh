#!/bin/bash
# one-limb kernel: sub-variant sweep (development aid)
cd "$(dirname "$0")/.."
out=gpurun_out/w1_sweep.log; : > $out
for v in ${VARIANTS:-13 14 15 16 17}; do
  BR_VARIANT=$v BR_WIDE_MAX=0 timeout -k 10 200 python scripts/br_bench.py ${COUNTS:-2048 8192 16384} >> $out 2>&1 || exit 1
done
grep -v amdgpu.ids $out
