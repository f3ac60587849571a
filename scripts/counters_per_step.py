#!/usr/bin/env python3
"""Per wave and CMux step: what a wave of a blind-rotation kernel spends its resident cycles on, from a scripts/pmc_passes.sh
summary (SQ counters count quad-cycles: x 4).   usage: counters_per_step.py <summary.txt> <gates per launch> <steps per launch> <waves per gate> [label]"""
import re
import sys

vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"BR (\S+)\s+n=\d+ avg=([0-9.e+-]+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
gates, steps, wpg = float(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
label = sys.argv[5] if len(sys.argv) > 5 else sys.argv[1]
ws = gates * steps * wpg  # wave-steps per launch
c = lambda k: 4.0 * vals[k] / ws
print("%s  (%g gates x %g steps per launch, %g wave(s) per gate)" % (label, gates, steps, wpg))
print("  resident cycles per wave and step        %8.0f" % c("SQ_WAVE_CYCLES"))
print("    issuing a vector instruction           %8.0f   (%d instructions per gate-step, %.2f cycles each)"
      % (c("SQ_ACTIVE_INST_VALU"), vals["SQ_INSTS_VALU"] / (gates * steps), c("SQ_ACTIVE_INST_VALU") / (vals["SQ_INSTS_VALU"] / ws)))
print("    an LDS instruction executing           %8.0f   (%d per gate-step)" % (c("SQ_ACTIVE_INST_LDS"), vals["SQ_INSTS_LDS"] / (gates * steps)))
print("    vector memory                          %8.0f   (%d reads per gate-step)" % (c("SQ_ACTIVE_INST_VMEM"), vals["SQ_INSTS_VMEM_RD"] / (gates * steps)))
print("    ready but not issuing (SQ_WAIT_INST_ANY) %6.0f   (of which waiting for the LDS queue %.0f)" % (c("SQ_WAIT_INST_ANY"), c("SQ_WAIT_INST_LDS")))
print("    on s_waitcnt (SQ_WAIT_ANY)             %8.0f" % c("SQ_WAIT_ANY"))
waves_per_simd = 2.0
print("  vector pipe busy per SIMD (two waves)    %8.1f %%" % (100.0 * waves_per_simd * vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"]))
print("  LDS busy per CU (eight waves)            %8.1f %%" % (100.0 * 8 * vals["SQ_ACTIVE_INST_LDS"] / vals["SQ_WAVE_CYCLES"]))
if "GRBM_GUI_ACTIVE" in vals:
    print("  shader cycles per launch (GRBM_GUI_ACTIVE / 8)  %.3e" % (vals["GRBM_GUI_ACTIVE"] / 8))
if "TCC_HIT_sum" in vals:
    print("  L2 hit rate %.3f, HBM bytes per gate-step %.0f (FETCH_SIZE x 2 + WRITE_SIZE, KB)"
          % (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / (gates * steps)))
