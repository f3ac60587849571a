#!/bin/bash
cd "$(dirname "$0")/.."
run() { IEACHE_LEVEL_CAP=$2 timeout -k 10 200 python bench.py --workload mul32 --batch $1 --steps 1 --warmup 0 --no-cpu-baseline --mul32-leg off 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('batch $1 cap ${2:-auto}: %.0f gate ops/s, %.2f s per batch' % (d['value'], d['ms_per_step']/1e3))"; }
run 58 ""; run 64 ""; run 100 ""; run 100 1056; run 128 ""; run 128 48; run 32 ""; run 32 1056; run 200 ""; run 200 30
