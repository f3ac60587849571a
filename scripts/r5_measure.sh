#!/bin/bash
# Round-5 measurement set on one MI355X (run through gpurun from the repo root):
#   tests : the whole -m gpu suite
#   driver: the driver's bench command (all legs, product defaults: two streams) -> the JSON line + bench_details.json
#   stats : rocprofv3 --kernel-trace --stats of the primary leg with every launch on ONE stream (IEACHE_OVERLAP=0: the mode
#           per-kernel durations are quoted in), and of the primary + exact legs in the same mode
#   sizes : blind rotation alone by launch size (scripts/br_bench.py, one stream)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_meas
mkdir -p $OUT
for stage in "$@"; do
case $stage in
tests)
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gputests.txt 2>&1 || { tail -30 $OUT/gputests.txt; exit 1; }
  tail -2 $OUT/gputests.txt ;;
driver)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
  cp bench_details.json $OUT/bench_driver_details.json
  echo "driver command done"; wc -c $OUT/bench_driver_cmd.json; tail -4 $OUT/bench_driver_cmd.err ;;
stats)
  rm -rf $OUT/stats_add16 $OUT/stats_exact
  IEACHE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_add16 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs none --exact-leg off > $OUT/add16_rocprof.json 2> $OUT/add16_rocprof.err
  find $OUT/stats_add16 -name "*kernel_stats.csv" | head -3
  IEACHE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_exact -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --legs none > $OUT/exact_rocprof.json 2> $OUT/exact_rocprof.err
  find $OUT/stats_exact -name "*kernel_stats.csv" | head -3 ;;
sizes)
  IEACHE_OVERLAP=0 timeout -k 10 300 python scripts/br_bench.py 1 37 128 256 300 512 768 1024 1216 1400 1536 1792 2048 4096 8192 16384 > $OUT/kernels_by_launch_size.txt 2>&1
  echo "# the product's defaults (streams on: rotation of roles at 1 025 .. 1 792 and 2 049 .. 2 688 gates, level halves from 4 096):" >> $OUT/kernels_by_launch_size.txt
  timeout -k 10 300 python scripts/br_bench.py 1024 1100 1216 1280 1365 1400 1536 1600 1700 1792 1900 2048 2100 2304 2560 2688 2816 3072 4096 8192 16384 >> $OUT/kernels_by_launch_size.txt 2>&1
  tail -32 $OUT/kernels_by_launch_size.txt ;;
esac
done
