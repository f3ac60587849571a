#!/bin/bash
# Round-2 measurement set on one MI355X (run through gpurun from the repo root):
#   1. the driver's bench command under rocprofv3 --kernel-trace --stats (shortened primary leg, full mul32 leg)
#   2. PMC passes over the blind-rotation / key-switch microbenchmark
#   3. kernel-variant and slice sweeps
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_meas
rm -rf $OUT/stats $OUT/pmc; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_rocprof.json 2> $OUT/bench_rocprof.err
echo "bench under rocprof done"; tail -c 600 $OUT/bench_rocprof.json
find $OUT/stats -name "*kernel_stats.csv" | head -3
bash scripts/pmc_passes.sh $OUT/pmc python3 scripts/br_bench.py 8192 > $OUT/pmc.log 2>&1 || echo "pmc failed"
tail -30 $OUT/pmc/summary.txt || true
for v in 13 14 15; do BR_VARIANT=$v python3 scripts/br_bench.py 8192 16384; done > $OUT/variants.txt 2>&1
EXACT_FFT=1 python3 scripts/br_bench.py 8192 16384 >> $OUT/variants.txt 2>&1
for s in 8 32; do BR_SLICE=$s python3 scripts/br_bench.py 8192; done >> $OUT/variants.txt 2>&1
cat $OUT/variants.txt
