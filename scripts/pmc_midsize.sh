#!/bin/bash
# SQ counters of the mid-size / latency blind-rotation kernels, one launch size per kernel (three rocprofv3 --pmc passes each;
# scripts/pmc_passes.sh is the full set for the wide-launch kernel).  Prints per-launch averages and, per gate-step, what a
# wave spends its cycles on.   usage: scripts/pmc_midsize.sh <outdir> [sizes...]      (default 256 512 1024 8192)
set -e
OUT=$1; shift
SIZES=${@:-256 512 1024 8192}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for N in $SIZES; do
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/n$N/pass$i -- python3 scripts/br_bench.py $N > $OUT/n$N.pass$i.log 2>&1 || echo "size $N pass $i failed"
  done
done
python3 - "$OUT" $SIZES <<'PY'
import csv, glob, sys, collections
out, sizes = sys.argv[1], [int(s) for s in sys.argv[2:]]
with open(out + "/summary.txt", "w") as fo:
    def emit(line):
        print(line); fo.write(line + "\n")
    for n in sizes:
        agg = collections.defaultdict(list)
        name = None
        for f in glob.glob("%s/n%d/pass*/*/*counter_collection.csv" % (out, n)):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "k_blind_rotate" in k:
                    name = k.split("(")[0]
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if not agg:
            emit("size %d: no counters" % n); continue
        a = {c: sum(v) / len(v) for c, v in agg.items()}
        launches = len(next(iter(agg.values())))
        steps = 630.0 * 3 / launches  # br_bench.py: three repetitions of one whole rotation (n = 630) per size
        emit("## %d gates per launch: %s (%d launches per pass, %.2f CMux steps each)" % (n, name, launches, steps))
        for c in sorted(a):
            emit("  %-24s avg=%.5g" % (c, a[c]))
        waves = a.get("SQ_WAVES", 0)
        if waves:
            ws = waves * steps  # wave-steps per launch
            g = lambda c: a.get(c, 0.0) / ws
            emit("  per wave and CMux step: %.0f cycles resident = %.0f issuing VALU + %.0f LDS + %.0f VMEM + %.0f scalar/misc; waiting on a counter %.0f (LDS part %.0f); "
                 "VALU instructions %.0f, LDS %.0f, VMEM reads %.0f, SALU %.0f"
                 % (g("SQ_WAVE_CYCLES") * 4, g("SQ_ACTIVE_INST_VALU") * 4, g("SQ_ACTIVE_INST_LDS") * 4, g("SQ_ACTIVE_INST_VMEM") * 4,
                    (g("SQ_ACTIVE_INST_SCA") + g("SQ_ACTIVE_INST_MISC")) * 4, g("SQ_WAIT_INST_ANY") * 4, g("SQ_WAIT_INST_LDS") * 4,
                    g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_SALU")))
PY
