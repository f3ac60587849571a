#!/bin/bash
# mid-size launches: two-wave one-limb kernel against the others (development aid)
cd "$(dirname "$0")/.."
out=gpurun_out/w2s_sweep.log; : > $out
C="257 384 512 768 1024 1216 1536 2048"
BR_VARIANT=20 timeout -k 10 200 python scripts/br_bench.py $C >> $out 2>&1 || exit 1
BR_VARIANT=21 timeout -k 10 200 python scripts/br_bench.py 512 1024 >> $out 2>&1 || exit 1
BR_VARIANT=13 timeout -k 10 200 python scripts/br_bench.py $C >> $out 2>&1 || exit 1
BR_VARIANT=7 timeout -k 10 200 python scripts/br_bench.py 128 256 257 384 512 >> $out 2>&1 || exit 1
EXACT_FFT=1 BR_WIDE_MAX=0 timeout -k 10 200 python scripts/br_bench.py 257 512 1024 >> $out 2>&1 || exit 1
timeout -k 10 200 python scripts/br_bench.py 200 300 1000 1100 3000 8192 >> $out 2>&1 || exit 1
grep -v amdgpu.ids $out
