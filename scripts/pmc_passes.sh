#!/bin/bash
# Collects PMC counters for the blind-rotation / key-switch kernels in separate passes
# (rocprofv3 slot limits: SQ 8, TCC 4 with FETCH_SIZE=3, WRITE_SIZE=2).
# usage: scripts/pmc_passes.sh <outdir> <command...>
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = "BR" if "blind_rotate" in k else ("KS" if ("keyswitch" in k or "k_ksm_gemm" in k) else None)
        if short: agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            line = "%s %-24s n=%d avg=%.4g" % (k, c, len(v), sum(v) / len(v))
            print(line); fo.write(line + "\n")
PY
