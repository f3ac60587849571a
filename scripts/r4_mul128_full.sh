#!/bin/bash
# configs[4]'s whole per-GPU share in one timed pass: 128-bit MUL x 1024 expressions = 124 092 416 bootstraps (about 600 s).
# A ticker keeps the box's hang detector fed while the one long pass runs.
mkdir -p gpurun_out/r4
( while sleep 60; do echo "tick $(date +%H:%M:%S)"; done ) &
T=$!
timeout -k 10 1100 python bench.py --batch 64 --steps 1 --warmup 1 --legs mul128 --mul128-batch 1024 --time-box 5000 \
    --exact-leg off --cpu-seconds 3 > gpurun_out/r4/mul128x1024_bench.json 2> gpurun_out/r4/mul128x1024_bench.err
rc=$?
kill $T
tail -5 gpurun_out/r4/mul128x1024_bench.err
exit $rc
