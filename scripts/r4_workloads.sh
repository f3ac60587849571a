#!/bin/bash
# Other workloads, latencies and serving on one MI355X (run through gpurun from the repo root)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_work; mkdir -p $O
timeout -k 10 300 python bench.py --workload mul128 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --legs none > $O/mul128x16.json 2> $O/mul128x16.err || exit 1
echo mul128x16 done
timeout -k 10 400 python scripts/latency.py > $O/latency.txt 2>&1 || exit 1
echo latency done
timeout -k 10 400 python scripts/serve_bench.py 64 > $O/serving.txt 2>&1 || exit 1
echo serving done
timeout -k 10 300 python scripts/br_bench.py 1 37 128 256 300 512 768 1024 1216 1536 2048 4096 8192 16384 > $O/kernels_by_launch_size.txt 2>&1 || exit 1
echo kernels done
