#!/bin/bash
# per-gate key switch on tiny launches: split walk vs one workgroup per gate (development aid)
cd "$(dirname "$0")/.."
for m in 1 16; do KS_SPLIT_MAX=$m timeout -k 10 200 python scripts/br_bench.py 1 2 8 44 128 256 400 2>&1 | grep "^variant"; done
