#!/bin/bash
# configs[4]'s whole per-GPU share in one timed pass: 128-bit MUL x 1024 expressions = 124 092 416 bootstraps (about 600 s).
# A ticker keeps the box's hang detector fed while the one long pass runs.
mkdir -p gpurun_out/r5_final
( while sleep 60; do echo "tick $(date +%H:%M:%S)"; done ) &
T=$!
timeout -k 10 1100 python bench.py --batch 64 --steps 1 --warmup 1 --legs mul128 --mul128-batch 1024 --time-box 5000 \
    --exact-leg off --cpu-seconds 3 --details gpurun_out/r5_final/mul128x1024_full_share_details.json > gpurun_out/r5_final/mul128x1024_full_share.json 2> gpurun_out/r5_final/mul128x1024_full_share.err
rc=$?
kill $T
tail -5 gpurun_out/r5_final/mul128x1024_full_share.err
exit $rc
