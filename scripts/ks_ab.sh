#!/bin/bash
# key-switch time of the default library against build/alt_<name> (scripts/ks_mfma_check.py columns: walk, mfma (auto split), mfma/1..8)
cd "$(dirname "$0")/.."
for r in 1 2; do
  for n in default "$@"; do
    if [ "$n" = default ]; then L=""; else L=$PWD/ie-ache_amd/csrc/build/alt_$n/libieache.so; fi
    echo "## build $n (round $r)"
    IEACHE_LIBRARY=$L timeout -k 10 300 python scripts/ks_mfma_check.py 1024 4096 8192 16384 2>&1 | grep count
  done
done
