#!/bin/bash
# Other workloads, latencies and serving on one MI355X (run through gpurun from the repo root)
cd "$(dirname "$0")/.."
O=gpurun_out/r2_work; mkdir -p $O
[ -n "$SKIP_SWEEP" ] || { VARIANTS="13 19 18" COUNTS="8192 16384" bash scripts/w1_sweep.sh > $O/w1_variants.txt 2>&1 || exit 1; }
echo sweep done
timeout -k 10 300 python bench.py --workload muladd64 --steps 1 --warmup 0 --no-cpu-baseline --mul32-leg off > $O/muladd64x128.json 2> $O/muladd64x128.err || exit 1
echo muladd64 done
timeout -k 10 300 python bench.py --workload mul128 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --mul32-leg off > $O/mul128x16.json 2> $O/mul128x16.err || exit 1
echo mul128x16 done
timeout -k 10 400 python bench.py --workload mul128 --batch 128 --steps 1 --warmup 0 --no-cpu-baseline --mul32-leg off > $O/mul128x128.json 2> $O/mul128x128.err || exit 1
echo mul128x128 done
timeout -k 10 400 python scripts/latency.py > $O/latency.txt 2>&1 || exit 1
echo latency done
timeout -k 10 400 python scripts/serve_bench.py 64 > $O/serving.txt 2>&1 || exit 1
echo serving done
