#!/bin/bash
# rocprofv3 --kernel-trace --stats of the whole default bench command in the product's default mode (two streams): which kernels the
# GPU time goes to.  (Per-kernel DURATIONS under two streams describe kernels that share the chip: profiles/r5_add16_kernel_stats.csv,
# collected with IEACHE_OVERLAP=0, is the one-stream summary the roofline cites.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_final
mkdir -p $O
rm -rf $O/stats_all
( while sleep 60; do echo tick; done ) &
T=$!
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_all -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --details $O/all_rocprof_details.json > $O/all_rocprof.json 2> $O/all_rocprof.err
rc=$?
kill $T
f=$(find $O/stats_all -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/all_kernel_stats.csv
rm -rf $O/stats_all
head -8 $O/all_kernel_stats.csv | cut -c1-100
exit $rc
