#!/bin/bash
# 32-bit multiplier at small batches: ASAP levels against balanced levels of a forced width (development aid)
cd "$(dirname "$0")/.."
run() { IEACHE_LEVEL_CAP=$2 timeout -k 10 200 python bench.py --workload mul32 --batch $1 --steps 1 --warmup 0 --no-cpu-baseline --mul32-leg off 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('batch $1 cap ${2:-asap}: %.0f gate ops/s, %.2f s per batch, levels %s' % (d['value'], d['ms_per_step']/1e3, d['config']['levels']))"; }
for b in ${BATCHES:-58 64 128 256}; do
  run $b ""
  for c in $(python -c "
b=$b
s=set()
for R in (1024,2048,4096):
    c=R//b
    if 8<=c<=70: s.add(c)
print(' '.join(str(c) for c in sorted(s)))"); do run $b $c; done
done
