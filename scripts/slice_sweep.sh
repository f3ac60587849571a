#!/bin/bash
# CMux steps per launch for the one-wave kernel (development aid)
cd "$(dirname "$0")/.."
out=gpurun_out/slice_sweep.log; : > $out
for s in 16 21 24 32 40 42 48 63 16; do
  BR_SLICE=$s timeout -k 10 200 python scripts/br_bench.py 8192 16384 4096 >> $out 2>&1 || exit 1
done
grep -v amdgpu.ids $out
