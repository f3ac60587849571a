#!/bin/bash
# Round-4 measurement set on one MI355X (run through gpurun from the repo root; stages keep each call under gpurun's limit):
#   driver: the driver's bench command (all legs) -> the JSON line
#   stats:  the primary leg alone, then the whole default command, under rocprofv3 --kernel-trace --stats
#   pmc / pmc_x1: PMC passes over the blind-rotation / key-switch microbenchmark (one-limb default / exact_fft)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_meas
mkdir -p $OUT
for stage in "$@"; do
case $stage in
driver)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
  echo "driver command done"; tail -c 300 $OUT/bench_driver_cmd.json; echo ;;
stats)
  rm -rf $OUT/stats_add16 $OUT/stats_all
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_add16 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs none --exact-leg off > $OUT/add16_rocprof.json 2> $OUT/add16_rocprof.err
  find $OUT/stats_add16 -name "*kernel_stats.csv" | head -3
  rm -rf $OUT/stats_exact   # the exact leg (two-limb kernels) beside it: primary leg + exact leg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_exact -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --legs none > $OUT/exact_rocprof.json 2> $OUT/exact_rocprof.err
  find $OUT/stats_exact -name "*kernel_stats.csv" | head -3
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_all -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/all_rocprof.json 2> $OUT/all_rocprof.err
  find $OUT/stats_all -name "*kernel_stats.csv" | head -3 ;;
pmc)
  rm -rf $OUT/pmc
  bash scripts/pmc_passes.sh $OUT/pmc python3 scripts/br_bench.py 8192 > $OUT/pmc.log 2>&1 || echo "pmc failed"
  tail -50 $OUT/pmc/summary.txt || true ;;
pmc_x1)   # the same passes over the two-limb one-wave-per-gate kernel (exact_fft)
  rm -rf $OUT/pmc_x1
  EXACT_FFT=1 bash scripts/pmc_passes.sh $OUT/pmc_x1 python3 scripts/br_bench.py 8192 > $OUT/pmc_x1.log 2>&1 || echo "pmc_x1 failed"
  tail -50 $OUT/pmc_x1/summary.txt || true ;;
esac
done
