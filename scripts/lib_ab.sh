#!/bin/bash
# Same measurement against several builds of the library, interleaved (scripts/build_alt.sh): lib_ab.sh <rounds> <name...> -- <variant_check args>
R=$1; shift
NAMES=()
while [ "$1" != "--" ]; do NAMES+=("$1"); shift; done
shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $R); do
  for n in "${NAMES[@]}"; do
    if [ "$n" = default ]; then L=""; else L=$PWD/ie-ache_amd/csrc/build/alt_$n/libieache.so; fi
    echo "## build $n (round $r)"
    IEACHE_LIBRARY=$L REPS=5 timeout -k 10 300 python scripts/variant_check.py "$@" 2>&1 | grep variant
  done
done
