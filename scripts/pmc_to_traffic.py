#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.txt (made by scripts/pmc_passes.sh over `scripts/br_bench.py 8192`) ->
profiles/traffic.json, the committed counter evidence bench.py's roofline reads.

HBM bytes as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) is doubled
(wide coalesced reads are tallied at 64 B per 128-B request), WRITE_SIZE (KB) taken as is, separate passes.

  python scripts/pmc_to_traffic.py profiles/r2_f_pmc_summary.txt [gates_per_launch=8192] [cmux_steps_per_launch=16] [kernel key]

The kernel key is what ieache_ctx_kernel_variant() reports ("w1x64-radix8-onelimb", the default, or
"x1x64-radix8-twolimb" with exact_fft); the other kernel's entry in traffic.json is kept.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary = sys.argv[1]
gates = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
key = sys.argv[4] if len(sys.argv) > 4 else "w1x64-radix8-onelimb"
# FP64 instructions per gate and CMux step, counted in the kernels' ISA (scripts/isa_count.py)
# (round 4, twist folded into the radix-8 passes: profiles/r4_isa_mix.txt)
KERNELS = {"w1x64-radix8-onelimb": ("k_blind_rotate_w1b<3,7,guard on 1 coefficient in 4>", 2425), "x1x64-radix8-twolimb": ("k_blind_rotate_x1<3,7>", 3280)}
vals = {}
for line in open(summary):
    m = re.match(r"(BR|KS) (\S+)\s+n=(\d+) avg=([0-9.e+-]+)", line)
    if m:
        vals[(m.group(1), m.group(2))] = float(m.group(4))
rel = os.path.relpath(os.path.abspath(summary), ROOT)


def hbm(k):
    return (2.0 * vals[(k, "FETCH_SIZE")] + vals[(k, "WRITE_SIZE")]) * 1024.0


path = os.path.join(ROOT, "profiles", "traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out.update({
    key: {
        "kernel": KERNELS[key][0], "fp64_insts_per_gate_step": KERNELS[key][1], "pmc_summary": rel,
        "source": rel + " (scripts/pmc_passes.sh: separate rocprofv3 --pmc passes over scripts/br_bench.py %d)" % gates,
        "gates_per_launch": gates, "cmux_steps_per_launch": steps,
        "FETCH_SIZE_KB_avg": vals[("BR", "FETCH_SIZE")], "WRITE_SIZE_KB_avg": vals[("BR", "WRITE_SIZE")],
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is; x1024 B",
        "hbm_bytes_per_launch": hbm("BR"), "hbm_bytes_per_gate_step": hbm("BR") / (gates * steps),
        "l2_hit_rate": vals[("BR", "TCC_HIT_sum")] / (vals[("BR", "TCC_HIT_sum")] + vals[("BR", "TCC_MISS_sum")]),
        "SQ_INSTS_VALU_per_launch": vals[("BR", "SQ_INSTS_VALU")],
        "valu_insts_per_gate_step": vals[("BR", "SQ_INSTS_VALU")] / (gates * steps),
        "SQ_INSTS_LDS_per_gate_step": vals[("BR", "SQ_INSTS_LDS")] / (gates * steps),
        "valu_busy_of_wave_cycles_x_waves_per_simd": 2 * vals[("BR", "SQ_ACTIVE_INST_VALU")] / vals[("BR", "SQ_WAVE_CYCLES")],
        "shader_cycles_per_gate_step": vals[("BR", "GRBM_GUI_ACTIVE")] / 8 / (gates * steps),
        "effective_clock_GHz_note": "GRBM_GUI_ACTIVE / 8 XCDs = %.3e shader cycles per launch (MI355X_MICROARCH.md, DVFS give-back: effective clock = that / kernel wall time)" % (vals[("BR", "GRBM_GUI_ACTIVE")] / 8),
    },
    "keyswitch": {
        "kernel": "k_ksm_gemm (int8 MFMA product)", "pmc_summary": rel, "gates_per_launch": gates,
        "FETCH_SIZE_KB_avg": vals[("KS", "FETCH_SIZE")], "WRITE_SIZE_KB_avg": vals[("KS", "WRITE_SIZE")],
        "hbm_bytes_per_launch": hbm("KS"),
        "l2_hit_rate": vals[("KS", "TCC_HIT_sum")] / (vals[("KS", "TCC_HIT_sum")] + vals[("KS", "TCC_MISS_sum")]),
    },
})
# Issue bound of the kernel's own instruction stream: its vector instructions priced at the wall time a stream of such
# instructions takes per wave-instruction and SIMD with two waves per SIMD (scripts/ubench/valu_rates.hip,
# profiles/r3_issue_costs.txt: FP64 arithmetic / conversions ~5.3 'cycles at 2.4 GHz' = 2.21 ns, the other vector
# instructions ~4.5 = 1.875 ns; already at the clock the chip holds under FP64 load)
if key in KERNELS:
    e = out[key]
    fp64, other = e["fp64_insts_per_gate_step"], e["valu_insts_per_gate_step"] - e["fp64_insts_per_gate_step"]
    t = fp64 * 2.21e-9 + other * 1.875e-9
    e["issue_model"] = {"ns_per_fp64_slot": 2.21, "ns_per_other_vector_slot": 1.875, "gate_step_ns_of_one_simd": t * 1e9,
                        "issue_bound_gates_per_s": 256 * 4 / (t * 630), "source": "profiles/r3_issue_costs.txt (two waves per SIMD)"}
# the rocprofv3 --kernel-trace --stats average of the same kernel (primary bench leg alone), if the caller names the file
for a in sys.argv[5:]:
    if a.startswith("stats="):
        import csv
        f = a[6:]
        for row in csv.DictReader(open(f)):
            if KERNELS[key][0].split("<")[0] in row["Name"] and "prologue" not in row["Name"]:
                out[key]["rocprof_avg_launch_ms"] = float(row["AverageNs"]) * 1e-6
                out[key]["rocprof_calls"] = int(row["Calls"])
                out[key]["rocprof_stats"] = os.path.relpath(os.path.abspath(f), ROOT) + " (rocprofv3 --kernel-trace --stats -- python3 bench.py %s)" % ("--steps 3 --warmup 1 --no-cpu-baseline --legs none --exact-leg off" if key.startswith("w1") else "--steps 1 --warmup 1 --no-cpu-baseline --legs none: the primary leg, then the exact leg's two passes")
                break
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[key], indent=1))
