#!/usr/bin/env python3
"""Benchmark of the hot path: bootstrapped gate ops/s on level-batched circuits.

  python bench.py --gpus N --steps K --warmup W [--workload add16|mul32|...] [--batch B]

One "step" = one pass of a whole circuit (every level, every gate) over one
batch of expressions whose ciphertexts are already resident in HBM.  The timed
K steps run the primary workload = BASELINE.json configs[1]: 16-bit ADD, batch
4096 per GPU.  The default invocation then adds, on the same resident key, one
full timed pass each of
  * configs[2]: 32-bit shift-add MUL, batch 1024 per GPU   -> `mul32`   (gate ops/s, encrypted 32-bit MUL/s)
  * configs[3]'s per-GPU share: 64-bit a*b+c, batch 128     -> `muladd64`
  * configs[4]'s circuit: 128-bit MUL, a time-boxed sub-batch of 128 of its 1024 per GPU -> `mul128`
every product decrypt-checked, each with its own roofline object.  (--extras adds the two opt-in,
decrypt-identical-only multipliers: constant-folded and carry-save.)

N > 1: `python bench.py --gpus N ...` starts its own ranks (python -m torch.distributed.run, one
child process per GPU, before this process touches the GPU) unless it already runs under
torch.distributed.run (WORLD_SIZE set).  Expressions shard across ranks with no data-path
collective (weak scaling: the per-GPU batch is fixed), after a one-time RCCL broadcast of the
bootstrapping / key-switch key.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6          # 256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
LANES, FLOP_PER_INST = 64, 2          # one wave-instruction = 64 lanes; an FMA is 2 flop, like the peak
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4  # wave-instructions/s the chip can issue: 256 CUs x 4 SIMDs, one per 4 cycles at 2.4 GHz


def algorithmic_flops_per_gate(p, limbs=1):
    """per CMux step (k+1)l forward + (k+1) inverse negacyclic transforms of N/2 complex points
    (5 M log2 M flop each) + (k+1)^2 l M complex multiply-accumulates (8 flop each); times n steps.
    n=630, N=1024, k=1, l=3: 233 472 flop per step, 1.47e8 per gate.  limbs=2 (the provably exact product: BK in two
    16-bit limbs): the forward transforms are shared, inverse transforms and multiply-accumulates double -- 328 704 per step."""
    M = p.N // 2
    logM = M.bit_length() - 1
    per_step = ((p.k + 1) * p.l + limbs * (p.k + 1)) * 5 * M * logM + limbs * (p.k + 1) ** 2 * p.l * M * 8
    return per_step * p.n


WORKLOADS = {
    # name: (circuit kind, bits, default per-GPU batch, BASELINE.json config)
    "add16": (1, 16, 4096, "16-bit ADD, batch 4096 ciphertext pairs, 1xMI355X (level-batched bootstraps)"),
    "add32": (1, 32, 1024, "32-bit ADD"),
    "mul32": (4, 32, 1024, "32-bit shift-add MUL, batch 1024, 1xMI355X"),
    "muladd32": (5, 32, 1024, "32-bit 3-operand a*b+c (AC058.pdf Fig. 7 A+B*C)"),
    "muladd64": (5, 64, 128, "64-bit 3-operand a*b+c, batch 1024, sharded across 8xMI355X"),
    "mul128": (4, 128, 1024, "128-bit multi-precision MUL, batch 8192, 8xMI355X"),
}


def algorithmic_bytes(p):
    """SURVEY.md 8(d) / BASELINE.md 3: BK n(k+1)l(k+1)N*4, expected KSK rows N*t*(1-2^-basebit)*(n+1)*4, LWE I/O 3(n+1)*4"""
    bk = p.n * (p.k + 1) * p.l * (p.k + 1) * p.N * 4
    ksk = p.k * p.N * p.ks_t * (1.0 - 2.0 ** -p.ks_basebit) * (p.n + 1) * 4
    io = 3 * (p.n + 1) * 4
    return bk, ksk, io


def pmc_counters(kernel_variant):
    """Committed PMC evidence for the dominant kernel (rocprofv3 cannot run inside this process):
    profiles/traffic.json names the launch geometry and the summary file the counters come from
    (scripts/pmc_passes.sh); SQ_INSTS_VALU is parsed from that summary."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(kernel_variant)
        if not tj:
            return None
        out = {"kernel": tj.get("kernel"), "fp64_insts_per_gate_step": tj.get("fp64_insts_per_gate_step", 0),
               "hbm_bytes_per_gate_step": tj["hbm_bytes_per_gate_step"], "l2_hit_rate": tj.get("l2_hit_rate"),
               "source": tj.get("pmc_summary", tj.get("source")), "shader_cycles_per_gate_step": tj.get("shader_cycles_per_gate_step"),
               "gates_per_launch": tj.get("gates_per_launch"), "cmux_steps_per_launch": tj.get("cmux_steps_per_launch"),
               "rocprof_avg_launch_ms": tj.get("rocprof_avg_launch_ms"), "rocprof_stats": tj.get("rocprof_stats"),
               "issue_model": tj.get("issue_model")}
        summ = tj.get("pmc_summary")
        if summ:
            for line in open(os.path.join(ROOT, summ)):
                m = re.match(r"BR SQ_INSTS_VALU\s+n=\d+ avg=([0-9.e+]+)", line)
                if m:
                    out["valu_insts_per_gate_step"] = float(m.group(1)) / (tj["gates_per_launch"] * tj["cmux_steps_per_launch"])
        return out
    except (OSError, ValueError, KeyError):
        return None


def roofline(p, stats, gate_rate, pmc, launch_kernel=None, limbs=1):
    """The record for one timed leg, for its dominant kernel (the blind rotation).

    bound fp64_valu: BK is shared by every gate of a launch out of L2 (hbm_model.reuse_factor), so what limits the
    kernel is the FP64 vector pipe.  `achieved` = ALGORITHMIC flops of one launch (233 472 flop per
    gate and CMux step x the gate-steps the launch processes) / its average duration (HIP events on the evaluator's
    own stream, around the kernel's launches only); `frac` = that / the dense FP64 vector peak.  `valu_issue` says how
    full the vector ISSUE slots are (every instruction counted once, whatever it does: FP64 FMAs and adds, index
    arithmetic, cross-lane moves) -- a utilisation of the pipe, not a flop rate.
    hbm_model is SURVEY 8(d)'s streaming-model figure, priced end to end on the leg's own rate.
    launch_kernel: the kernel this leg's launch size selects (Context.kernel_for_launch); the committed PMC / rocprof
    evidence is attached only when it was collected on that kernel."""
    bk_b, ksk_b, io_b = algorithmic_bytes(p)  # the streaming model counts the key as libtfhe holds it, whatever form the kernel reads
    per_gate = bk_b + ksk_b + io_b
    br_avg_ms = stats.blind_rotate_ms / max(1, stats.blind_rotate_launches)
    ks_avg_ms = stats.keyswitch_ms / max(1, stats.keyswitch_launches)
    gates_per_launch = stats.bootstraps / max(1, stats.chunks)
    launches_per_gate = stats.blind_rotate_launches / max(1, stats.chunks)
    steps_per_launch = p.n / max(1.0, launches_per_gate)
    br_gate_rate = stats.bootstraps / max(1e-9, stats.blind_rotate_ms * 1e-3)   # gates/s of the blind rotation alone
    ks_gate_rate = stats.bootstraps / max(1e-9, stats.keyswitch_ms * 1e-3)
    pmc_other = None
    if pmc and launch_kernel and (pmc.get("kernel") or "").split("<")[0] != launch_kernel.split("<")[0]:
        pmc_other, pmc = pmc, None  # counters of another kernel say nothing about this one
    traffic = pmc["hbm_bytes_per_gate_step"] * gates_per_launch * steps_per_launch if pmc else None
    alg_flop = algorithmic_flops_per_gate(p, limbs)
    achieved = br_gate_rate * alg_flop * 1e-12
    out = {"bound": "fp64_valu", "unit": "TFLOP/s", "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS,
           "frac": achieved / FP64_VALU_PEAK_TFLOPS, "frac_algorithmic_flops": achieved / FP64_VALU_PEAK_TFLOPS,
           "traffic": traffic,
           "kernel": (pmc or {}).get("kernel") or launch_kernel or "k_blind_rotate", "avg_launch_ms": br_avg_ms,
           "gates_per_launch": gates_per_launch, "cmux_steps_per_launch": steps_per_launch,
           "algorithmic_flops_per_gate": alg_flop,
           "algorithmic_flops_per_launch": alg_flop * gates_per_launch * steps_per_launch / p.n,
           "rocprof_avg_launch_ms": (pmc or {}).get("rocprof_avg_launch_ms"), "rocprof_stats": (pmc or {}).get("rocprof_stats"),
           "blind_rotate_share": stats.blind_rotate_ms / max(1e-9, stats.total_ms),
           "keyswitch_share": stats.keyswitch_ms / max(1e-9, stats.total_ms),
           "note": "achieved = SURVEY 8(d) algorithmic flops (%d transforms x 5 M log2 M + %d M complex MACs per CMux step) of a launch / its "
                   "HIP-event duration; 100 %% = %.0f gates/s per GPU.  rocprof_avg_launch_ms: the same kernel's average in the committed "
                   "rocprofv3 --kernel-trace --stats summary (rocprof_stats), at the geometry named there"
                   % ((p.k + 1) * p.l + limbs * (p.k + 1), limbs * (p.k + 1) ** 2 * p.l, FP64_VALU_PEAK_TFLOPS * 1e12 / alg_flop)}
    if pmc_other:
        out["pmc_note"] = ("launches of %.0f gates take %s; the committed counter passes (%s) were collected on %s, so no traffic / "
                           "vector-issue figures are attached to this leg" % (gates_per_launch, launch_kernel, pmc_other.get("source"), pmc_other.get("kernel")))
    insts = pmc.get("valu_insts_per_gate_step") if pmc else None
    if insts:
        vi = {"insts_per_gate_step": insts, "fp64_insts_per_gate_step": pmc.get("fp64_insts_per_gate_step", 0),
              "utilisation": br_gate_rate * insts * p.n / VALU_ISSUE_PEAK,
              "utilisation_fp64_insts_only": br_gate_rate * pmc.get("fp64_insts_per_gate_step", 0) * p.n / VALU_ISSUE_PEAK,
              "source": pmc.get("source"),
              "pmc_geometry": {"gates_per_launch": pmc.get("gates_per_launch"), "cmux_steps_per_launch": pmc.get("cmux_steps_per_launch")},
              "note": "vector-issue slots: SQ_INSTS_VALU per gate-step (PMC pass, geometry above) x this leg's gate-steps/s over 256 CUs x 4 SIMDs x "
                      "2.4 GHz / 4 cycles; every vector instruction counts as one slot -- not a flop rate"}
        # the per-gate-step count does not depend on the launch size, but say so when the geometries differ
        g = vi["pmc_geometry"]
        vi["geometry_differs_from_pmc_run"] = bool(g["gates_per_launch"] and (abs(g["gates_per_launch"] - gates_per_launch) > 0.5 or
                                                                              abs((g["cmux_steps_per_launch"] or 0) - steps_per_launch) > 0.5))
        cyc = pmc.get("shader_cycles_per_gate_step")
        if cyc and br_avg_ms > 0:
            # the chip lowers its clock under this load (DVFS give-back): shader cycles of a launch from GRBM_GUI_ACTIVE / 8
            # (PMC run, scaled to this leg's launch geometry) over the launch time measured here
            ghz = cyc * gates_per_launch * steps_per_launch / (br_avg_ms * 1e-3) / 1e9
            vi["effective_clock_GHz"] = ghz
            vi["utilisation_at_effective_clock"] = vi["utilisation"] * 2.4 / ghz
        im = pmc.get("issue_model")
        if im and im.get("issue_bound_gates_per_s"):
            # the kernel's own instruction stream priced at measured per-instruction issue times (two waves per SIMD): how close
            # the launch comes to what its instructions cost, as opposed to the nominal 4-cycle slots above
            vi["issue_bound_gates_per_s"] = im["issue_bound_gates_per_s"]
            vi["frac_of_issue_bound"] = br_gate_rate / im["issue_bound_gates_per_s"]
            vi["issue_model"] = {k: im[k] for k in ("ns_per_fp64_slot", "ns_per_other_vector_slot", "source") if k in im}
        out["valu_issue"] = vi
    measured_gbs = (traffic / (br_avg_ms * 1e-3) / 1e9) if (traffic and br_avg_ms > 0) else None
    alg_launch = per_gate * gates_per_launch * steps_per_launch / p.n
    out["hbm_model"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "achieved": gate_rate * per_gate / 1e9, "frac": gate_rate * per_gate / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_gate": per_gate, "algorithmic_bytes_per_launch": alg_launch,
                        "measured_hbm_bytes_per_launch": traffic, "measured_hbm_GBps": measured_gbs,
                        "reuse_factor": (alg_launch / traffic) if traffic else None,
                        "note": "streaming model (every gate streams BK and its KSK rows once), priced end to end on this leg's gate rate: "
                                "above 1 means BK is reused out of L2, not that HBM is saturated; "
                                "measured = FETCH_SIZE x2 + WRITE_SIZE of the blind-rotation launches (PMC, gfx950 correction)"}
    out["per_kernel"] = {
        "blind_rotate": {"algorithmic_bytes_per_gate": bk_b, "model_GBps": br_gate_rate * bk_b / 1e9,
                         "gates_per_s": br_gate_rate, "avg_launch_ms": br_avg_ms},
        "keyswitch": {"algorithmic_bytes_per_gate": ksk_b, "model_GBps": ks_gate_rate * ksk_b / 1e9,
                      "gates_per_s": ks_gate_rate, "avg_launch_ms": ks_avg_ms,
                      "note": "model_GBps = streaming-model bytes x rate; the key switch runs as an int8 MFMA product whose B stream is served from L2 / Infinity Cache (profiles/traffic.json: keyswitch)"}}
    return out


def cpu_baseline(p, keys, seconds=12.0):
    """Times the CPU oracle's gate bootstrap on this host (rank 0, N=1 only).

    kind "port": the reference binary cannot be built (libtfhe absent).  `value` is the oracle's
    FP64-FFT back-end, which follows libtfhe's own algorithm, on ONE thread -- how the reference
    effectively runs (its OpenMP pragmas are inert).  `all_cores` is the same back-end with
    independent gates spread over every CPU the job may use (the host's
    hardware threads, capped by the container's CPU quota); the exact-integer back-end used for
    parity is reported beside them."""
    from oracle import oracle as O
    from ieache_amd import tools
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, keys["bk"], keys["ksk"])
    a = tools.encrypt_bits(p, keys["lwe_key"], np.array([1, 0], dtype=np.uint8), 1)
    b = tools.encrypt_bits(p, keys["lwe_key"], np.array([1, 1], dtype=np.uint8), 2)
    res = {}
    for mode, name, budget in ((O.POLYMUL_FFT, "fft", seconds * 0.45), (O.POLYMUL_NTT, "exact", seconds * 0.2)):
        ck.set_polymul(mode)
        ck.gate("and", a[0], b[0])  # warm-up (also builds the FFT-domain key once)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            ck.gate("xor" if n & 1 else "and", a[n & 1], b[n & 1])
            n += 1
        res[name] = (n, time.perf_counter() - t0)
    n, dt = res["fft"]
    ne, dte = res["exact"]
    # all host cores: batches of independent gates (4 per thread) until ~0.35 x `seconds` have passed
    # (many-core hosts scale far below linearly on this memory-bound loop, so the sample is bounded by time)
    cores = O.max_threads()
    ck.set_polymul(O.POLYMUL_FFT)
    per = 4 * cores
    A = np.ascontiguousarray(np.tile(a, (per // 2, 1)))
    B = np.ascontiguousarray(np.tile(b, (per // 2, 1)))
    ck.gates_batch("and", A[:cores], B[:cores], threads=0)  # thread pool warm-up
    count, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds * 0.35:
        ck.gates_batch("and", A, B, threads=0)
        count += per
    dta = time.perf_counter() - t0
    # BASELINE.md section 4.4's `ldconfig -p | grep tfhe` without a child process (this process holds the GPU):
    # the loader cache lists every library ldconfig knows by name
    probe = ""
    try:
        cache = open("/etc/ld.so.cache", "rb").read()
        probe = ", ".join(sorted({m.decode() for m in re.findall(rb"libtfhe[A-Za-z0-9_.+-]*", cache)}))
    except OSError:
        pass
    return {
        "value": n / dt, "unit": "bootstrapped gate ops/s", "cores": 1, "kind": "port",
        "sample": "%d AND/XOR gates (n=%d,N=%d) in %.1f s with the oracle's FP64-FFT back-end (libtfhe's algorithm), "
                  "1 thread as the reference runs (its OpenMP pragmas are inert); exact-integer back-end: %.2f gates/s"
                  % (n, p.n, p.N, dt, ne / dte),
        "all_cores": {"value": count / dta, "cores": cores,
                      "sample": "%d independent AND gates in %.1f s, OpenMP over gates on the %d CPUs this job may use "
                                "(the host's hardware threads capped by its cgroup CPU quota), same back-end" % (count, dta, cores)},
        "real_libtfhe": ("in the loader cache but not timed: %s" % probe) if probe else "unavailable on this host (no libtfhe* in /etc/ld.so.cache = `ldconfig -p | grep tfhe` empty)",
        "note": "a plain-C port, about half as fast as the library it restates: libtfhe documents ~13 ms per bootstrapped gate on one core "
                "(~77 gates/s; BASELINE.md section 1) with its hand-vectorised FFT, this port measures %.1f ms -- halve any GPU/CPU "
                "ratio read off `value`; the ratio is not a quality claim either way (roofline.frac is)" % (1e3 * dt / max(1, n)),
    }


LINE_LIMIT = 6000       # bytes: the one JSON line must stay far below what the driver reads back (round 4's 21 KB line was not parsed)
DETAILS_FILE = "bench_details.json"


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is a summary; bench_details.json keeps every digit)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if k in d and d[k] is not None}


def compact_roofline(r):
    out = _pick(r, ("bound", "unit", "achieved", "peak", "frac"))
    out["traffic"] = _r(r.get("traffic"))          # HBM bytes per launch (PMC), null when no counters exist for this kernel
    out.update(_pick(r, ("kernel", "avg_launch_ms", "gates_per_launch", "cmux_steps_per_launch", "algorithmic_flops_per_launch",
                         "rocprof_avg_launch_ms", "blind_rotate_share")))
    if r.get("rocprof_stats"):
        out["rocprof_stats"] = r["rocprof_stats"].split(" ")[0]      # the file; the command that made it is in bench_details.json
    hm = r.get("hbm_model") or {}
    if hm:
        out["hbm_frac"] = _r(hm.get("frac"))                        # SURVEY 8(d) streaming model, end to end (> 1: BK reused out of L2)
        out["measured_hbm_GBps"] = _r(hm.get("measured_hbm_GBps"))
    vi = r.get("valu_issue") or {}
    if vi:
        out["valu_issue_utilisation"] = _r(vi.get("utilisation"))
    if r.get("mode"):
        out["mode"] = r["mode"].split(":")[0]
        out["one_stream_gate_ops_per_s"] = _r(r.get("one_stream_gate_ops_per_s"))
    tr = r.get("timed_region")
    if tr:
        out["timed_region"] = _pick(tr, ("streams", "frac_end_to_end", "frac_blind_rotation_in_flight"))
    return out


def compact_line(full, details_path):
    """The ONE line rank 0 prints: the contract's keys, the roofline of the dominant kernel, the metric's own leg, the exact leg, one
    short object per further leg and the CPU baseline -- numbers only.  Every note, the per-kernel and counter breakdowns and the
    audit record stay in `details` (bench_details.json, written next to this script)."""
    out = {k: _r(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data") if k in full}
    out["config"] = _pick(full["config"], ("workload", "value_workload", "metric_leg_workload", "circuit", "batch_per_gpu",
                                           "bootstraps_per_expr", "levels", "params", "parallelism", "kernel", "overlap_streams",
                                           "key_broadcast_s", "rccl_ranks", "collective_backend", "per_rank_gate_ops_per_s"))
    out["roofline"] = compact_roofline(full["roofline"])
    if "metric_leg" in full:
        out["metric_leg"] = _pick(full["metric_leg"], ("workload", "batch_per_gpu", "gate_ops_per_s", "mul32_per_s", "passes",
                                                       "per_pass_gate_ops_per_s", "spread", "ms_per_pass", "roofline_frac"))
    if "mul32_per_s" in full:
        out["mul32_per_s"] = _r(full["mul32_per_s"])
    if "exact" in full:
        e = full["exact"]
        out["exact"] = _pick(e, ("gate_ops_per_s", "kernel", "bit_identical_to_primary_leg", "vs_primary", "passes", "ms_per_pass"))
        out["exact"]["roofline_frac"] = _r(e["roofline"]["frac"])
        out["exact"]["roofline_kernel"] = e["roofline"].get("kernel")
    for key in ("mul32", "muladd64", "mul128"):
        if key in full:
            l = full[key]
            out[key] = _pick(l, ("batch_per_gpu", "gate_ops_per_s", "expressions_per_s", "passes", "ms_per_pass", "sub_batch_of"))
            out[key]["roofline_frac"] = _r(l["roofline"]["frac"])
            for extra in ("folded", "carry_save"):
                if extra in l:
                    out[key][extra] = _pick(l[extra], ("mul32_per_s", "executed_gate_ops_per_s", "speedup_vs_reference_circuit"))
    if "fft_guard" in full:
        out["fft_guard"] = _pick(full["fft_guard"], ("max_rounding_deviation", "reruns_on_two_limb_kernel", "limit"))
        a = full["fft_guard"].get("audit")
        if isinstance(a, dict):
            out["fft_guard"]["audit"] = _pick(a, ("audits", "gates_compared", "mismatches"))
    if full.get("skipped_legs"):
        out["skipped_legs"] = [s["leg"] for s in full["skipped_legs"]]
    if "cpu_baseline" in full:
        c = full["cpu_baseline"]
        out["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind"))
        out["cpu_baseline"]["sample"] = c["sample"].split(" with ")[0]
        out["cpu_baseline"]["all_cores"] = _pick(c["all_cores"], ("value", "cores"))
        out["cpu_baseline"]["real_libtfhe"] = c["real_libtfhe"].split(" (")[0]
    out["details"] = details_path
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:     # never print a line the driver cannot read: shed the optional objects, largest first
        for k in ("mul128", "muladd64", "mul32", "fft_guard", "exact"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    return line


def write_details(full, path=None):
    """The full record the line summarises: `path` (--details), else bench_details.json next to the script (or in $TMPDIR when the
    tree is read-only).  Returns the name the line carries in `details`."""
    import tempfile
    for cand in ([path] if path else [os.path.join(ROOT, DETAILS_FILE), os.path.join(tempfile.gettempdir(), DETAILS_FILE)]):
        try:
            with open(cand, "w") as f:
                json.dump(full, f, indent=1)
            return DETAILS_FILE if cand == os.path.join(ROOT, DETAILS_FILE) else cand
        except OSError:
            continue
    return None


def make_inputs(ia, tools, torch, ctx, p, lwe_key, kind, bits, batch, rank, dev, seed_base):
    """Synthetic inputs: fresh encryptions of uniform random operands (seeded per rank), edge operands in slots 0-3."""
    info = ia.circuit_info(kind, bits)
    rng = np.random.default_rng(seed_base + rank)
    inb = rng.integers(0, 2, size=(batch, info.n_inputs), dtype=np.uint8)
    inb[:, 2 * bits:2 * bits + 32] = 0  # the carry word is always 0 (alice.c:147-149)
    if batch >= 4:  # 0, 1, 2^w-1, process.c's 2^(w-2)
        for slot, v in enumerate((0, 1, (1 << bits) - 1, 1 << (bits - 2))):
            inb[slot, :bits] = tools.int_to_bits(v, bits)
            inb[slot, bits:2 * bits] = tools.int_to_bits(v, bits)
    stride = ctx.lwe_stride
    d_in = torch.zeros((batch, info.n_inputs, stride), dtype=torch.int32, device=dev)
    rows_per = max(1, (1 << 26) // (info.n_inputs * (p.n + 1) * 4))
    for s in range(0, batch, rows_per):  # stream the encryption through host memory in <=64 MiB pieces
        e = min(batch, s + rows_per)
        ct = tools.encrypt_bits(p, lwe_key, inb[s:e], 7777 + seed_base + 131 * rank + s)
        d_in[s:e, :, :p.n + 1] = torch.from_numpy(ct).to(dev)
        del ct
    d_out = torch.zeros((batch, info.n_outputs, stride), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    return info, inb, d_in, d_out


def check_outputs(tools, p, lwe_key, kind, bits, inb, d_out, rank):
    """Every expression of the batch must decrypt to the integer result (P1 at full size)."""
    batch = inb.shape[0]
    for s0 in range(0, batch, 512):
        e0 = min(batch, s0 + 512)
        dec = tools.decrypt_bits(p, lwe_key, d_out[s0:e0, :, :p.n + 1].cpu().numpy())
        for e in range(s0, e0):
            a = tools.bits_to_int(inb[e, :bits])
            b = tools.bits_to_int(inb[e, bits:2 * bits])
            exp = {1: (a + b) % (1 << bits), 2: (a - b) % (1 << bits), 3: (b - a) % (1 << bits), 4: a * b}.get(kind)
            if kind == 5:
                exp = (a * b + tools.bits_to_int(inb[e, 2 * bits + 32:])) % (1 << (2 * bits))
            assert tools.bits_to_int(dec[e - s0]) == exp, "rank %d: expression %d decrypts wrong" % (rank, e)


def timed(torch, dist, world, dev, backend, fn):
    """barrier + synchronize on both sides, MAX over ranks"""
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        elapsed = max(per_rank)
    return elapsed, per_rank


T_START = time.perf_counter()


def stream_counters(ctx):
    """(circuit evaluations run as pipelines, levels issued as halves) so far: a leg used two streams when either moved during it"""
    return ctx.get_option("pipelined_evals"), ctx.get_option("overlapped_levels")


def elapsed_all_ranks(torch, dist, dev, backend):
    """seconds since the start of the run, MAX over ranks: what time-box decisions are taken on, so that every rank takes the same one"""
    so_far = time.perf_counter() - T_START
    if dist is not None:
        t = torch.tensor([so_far], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        so_far = float(t.item())
    return so_far


def progress(rank, msg):
    """One line on stderr per finished leg (stdout carries only the JSON line): a run that is silent for
    minutes looks hung to whoever is watching it."""
    if rank == 0:
        print("[bench %6.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def count_gpus_without_runtime(base="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs of this node as the kernel driver lists them (KFD topology: nodes with SIMDs), narrowed by HIP_/ROCR_VISIBLE_DEVICES;
    None when sysfs does not say (the ranks then report a shortage themselves)."""
    try:
        n = 0
        for node in os.listdir(base):
            props = dict(l.split()[:2] for l in open(os.path.join(base, node, "properties")) if len(l.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([d for d in v.split(",") if d.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves, as a child process tree
    (never exec: this must also work where a GPU-initialised process may not be replaced), relay rank 0's JSON line on
    stdout and the ranks' progress on stderr, return the launcher's exit code."""
    import socket
    import subprocess
    if args.backend == "nccl":
        have = count_gpus_without_runtime()  # no HIP / torch call in the launcher: it must never hold a GPU context
        if have is not None and have < args.gpus:
            print("[bench] --gpus %d but this node shows %d GPU(s); one rank per GPU is needed for RCCL "
                  "(--backend gloo rehearses the flow with ranks sharing a GPU)" % (args.gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("[bench] --gpus %d without a torch.distributed.run environment: starting %d ranks (rendezvous 127.0.0.1:%d)"
          % (args.gpus, args.gpus, port), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:  # rank 0's one JSON line (and nothing else) goes to our stdout
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


# the legs that follow the primary one in the default invocation: (key, workload, per-GPU batch, full per-GPU batch of the config)
DEFAULT_LEGS = (("mul32", "mul32", 1024, 1024), ("muladd64", "muladd64", 128, 128), ("mul128", "mul128", 128, 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="add16", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="expressions per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--legs", default="auto",
                    help="comma-separated extra legs after the primary one, each ONE full timed pass: mul32, muladd64, mul128 "
                         "(auto: all three for the default workload at its default batch; none: primary leg only)")
    ap.add_argument("--mul32-leg", default="auto", choices=["auto", "on", "off"], help="(kept for older command lines) on = --legs mul32")
    ap.add_argument("--mul32-batch", type=int, default=1024)
    ap.add_argument("--muladd64-batch", type=int, default=128)
    ap.add_argument("--mul128-batch", type=int, default=128)
    ap.add_argument("--time-box", type=float, default=330.0,
                    help="a leg is skipped (and named in `skipped_legs`) when the run has already taken this many seconds minus the leg's estimate")
    ap.add_argument("--metric-passes", type=int, default=2,
                    help="timed passes of the mul32 leg (the config BASELINE.json quotes its metric on); the line reports each and their spread")
    ap.add_argument("--details", default=None, help="where the full record goes (default: bench_details.json next to this script)")
    ap.add_argument("--roofline-steps", type=int, default=3,
                    help="steps of the primary batch run with every launch on one stream (overlap = 0) after the timed region, for the per-kernel roofline figures")
    ap.add_argument("--exact-leg", default="on", choices=["on", "off"],
                    help="after the primary leg: the same batch once more on the provably exact two-limb kernels (`exact` in the line)")
    ap.add_argument("--extras", action="store_true",
                    help="also time the two opt-in, decrypt-identical-only 32-bit multipliers (constant-folded, carry-save) in the mul32 leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N>1 flow with CPU collectives (ranks may then share one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # before anything touches the GPU in this process

    import torch
    import ieache_amd as ia
    from ieache_amd import tools

    from ieache_amd import parallel
    rank, world, local_rank, dist = parallel.init_distributed(args.backend)
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())  # rehearsal: ranks wrap around the GPUs present
    args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kind, bits, def_batch, config_name = WORKLOADS[args.workload]
    batch = args.batch or def_batch
    p = ia.default_params()  # n=630 N=1024 l=3 Bgbit=7 t=8 basebit=2

    # ---- keys: generated once on rank 0 (keygen.c seeds), broadcast over RCCL/xGMI ----
    keys = tools.keygen_raw(p, (314, 1592, 657)) if rank == 0 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bdev = dev if args.backend == "nccl" else torch.device("cpu")
    d_bk, d_ksk, d_key = (t.to(dev) for t in parallel.broadcast_cloud_key(p, keys, bdev, dist))
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0 if dist is not None else 0.0
    lwe_key = d_key.cpu().numpy()
    ctx = ia.Context.from_device_pointers(p, d_bk.data_ptr(), d_ksk.data_ptr(), device=local_rank)
    del d_bk, d_ksk
    if args.chunk:
        ctx.set_chunk(args.chunk)
    pmc = pmc_counters(ctx.kernel_variant)
    progress(rank, "cloud key resident on %d GPU(s), kernel %s" % (world, ctx.kernel_variant))

    # ---- primary leg ----
    info, inb, d_in, d_out = make_inputs(ia, tools, torch, ctx, p, lwe_key, kind, bits, batch, rank, dev, 1000)

    def step(stats=None):
        ctx.eval_batch_device(kind, bits, batch, d_in.data_ptr(), d_out.data_ptr(), stats)

    for _ in range(max(1, args.warmup)):  # at least one untimed pass: its outputs are what gets checked
        step()
    check_outputs(tools, p, lwe_key, kind, bits, inb, d_out, rank)

    stats = ia.Stats()
    progress(rank, "%s x%d: warm-up done, every expression decrypts; timing %d steps" % (args.workload, batch, args.steps))
    elapsed, per_rank = timed(torch, dist, world, dev, args.backend, lambda: [step(stats) for _ in range(args.steps)])
    primary_rate = info.bootstraps * batch * args.steps * world / elapsed
    progress(rank, "%s x%d: %.0f gate ops/s" % (args.workload, batch, primary_rate))

    # ---- per-kernel figures: the evaluator issues wide levels as two halves on two streams (option "overlap"), whose kernels share
    # the chip, so a launch's own duration is taken with every launch on ONE stream: a few more steps of the same batch with
    # overlap = 0, HIP events around each launch, and the committed rocprofv3 / PMC summaries are collected in that mode too ----
    overlap_on = bool(ctx.get_option("overlap")) and (ctx.get_option("pipelined_evals") > 0 or ctx.get_option("overlapped_levels") > 0)
    kstats, k_rate = stats, primary_rate / world
    if overlap_on and args.roofline_steps > 0:
        ctx.set_option("overlap", 0)
        step()                                            # untimed: one-stream scratch (a whole level per launch) in place
        kstats = ia.Stats()
        k_elapsed, _ = timed(torch, dist, world, dev, args.backend, lambda: [step(kstats) for _ in range(args.roofline_steps)])
        ctx.set_option("overlap", 1)
        k_rate = info.bootstraps * batch * args.roofline_steps / k_elapsed
        progress(rank, "%s x%d on one stream (per-kernel timings): %.0f gate ops/s" % (args.workload, batch, k_rate * world))

    # ---- the same batch once more on the PROVABLY EXACT product (two-limb transform, exact_fft = 1): one untimed pass, one timed
    # pass, and its outputs compared word for word with the guarded one-limb pass above ----
    exact_out, skipped = None, []
    run_exact = args.exact_leg != "off" and "onelimb" in ctx.kernel_variant
    if run_exact:
        # two passes (one untimed) at about 0.7 x the primary rate: budgeted against the time box like every other leg
        estimate = 2 * info.bootstraps * batch / max(1.0, 0.7 * primary_rate / world)
        so_far = elapsed_all_ranks(torch, dist, dev, args.backend)
        if so_far + estimate > args.time_box:
            skipped.append({"leg": "exact", "reason": "%.0f s elapsed + %.0f s estimated > --time-box %.0f s" % (so_far, estimate, args.time_box)})
            progress(rank, "exact leg skipped: time box")
            run_exact = False
    if run_exact:
        primary_out = d_out
        d_out = torch.zeros_like(primary_out)
        ctx.set_option("exact_fft", 1)
        exact_kernel_variant = ctx.kernel_variant
        step()                                            # untimed: first launches of the two-limb kernels, scratch in place
        est = ia.Stats()
        e_elapsed, e_per_rank = timed(torch, dist, world, dev, args.backend, lambda: step(est))
        ctx.set_option("exact_fft", 0)
        same = bool(torch.equal(primary_out, d_out))
        if dist is not None:
            t = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            same = bool(t.item() > 0.5)
        assert same, "the guarded one-limb pass and the exact two-limb pass differ"
        e_rate = info.bootstraps * batch * world / e_elapsed
        progress(rank, "%s x%d on the exact (two-limb) kernels: %.0f gate ops/s, outputs identical to the one-limb pass" % (args.workload, batch, e_rate))
        exact_out = {"what": "the primary leg's batch on the provably exact product (exact_fft = 1: BK in two balanced 16-bit limbs, every rounded "
                             "sum < 2^35 in a 53-bit mantissa, so rounding recovers the integer whatever the transform's schedule)",
                     "workload": config_name, "batch_per_gpu": batch, "passes": 1, "warmup_passes": 1, "ms_per_pass": e_elapsed * 1e3,
                     "gate_ops_per_s": e_rate, "per_rank_gate_ops_per_s": [info.bootstraps * batch / t for t in e_per_rank],
                     "kernel": exact_kernel_variant,
                     "bit_identical_to_primary_leg": same,
                     "checked": "all %d x %d output samples equal the one-limb pass's, word for word (torch.equal on the device buffers)" % (batch, info.n_outputs),
                     "vs_primary": e_rate / primary_rate,
                     "streams": 2 if overlap_on else 1,
                     "roofline": roofline(p, est, e_rate / world, pmc_counters(exact_kernel_variant),
                                          ctx_kernel_exact(ctx, round((2 if overlap_on else 1) * est.bootstraps / max(1, est.chunks))), limbs=2)}
        del primary_out
    del d_in, d_out

    # ---- the other configs, one full timed pass each on the same resident key ----
    if args.legs == "auto":
        legs = [l[0] for l in DEFAULT_LEGS] if (args.workload == "add16" and not args.batch and args.mul32_leg != "off") else []
        if args.mul32_leg == "on" and "mul32" not in legs:
            legs = ["mul32"]
    else:
        legs = [l for l in args.legs.split(",") if l and l != "none"]
    leg_batch = {"mul32": args.mul32_batch, "muladd64": args.muladd64_batch, "mul128": args.mul128_batch}
    leg_out = {}
    for key, wl, _, full_batch in DEFAULT_LEGS:
        if key not in legs:
            continue
        lkind, lbits, _, lname = WORKLOADS[wl]
        lb = leg_batch[key]
        linfo = ia.circuit_info(lkind, lbits)
        # one decision for all ranks (the pass contains barriers): skip a leg the time box has no room for
        estimate = int(linfo.bootstraps) * lb / max(1.0, primary_rate / world)
        so_far = elapsed_all_ranks(torch, dist, dev, args.backend)
        if so_far + estimate > args.time_box:
            skipped.append({"leg": key, "reason": "%.0f s elapsed + %.0f s estimated > --time-box %.0f s" % (so_far, estimate, args.time_box)})
            progress(rank, "%s x%d skipped: time box" % (key, lb))
            continue
        linfo, linb, ld_in, ld_out = make_inputs(ia, tools, torch, ctx, p, lwe_key, lkind, lbits, lb, rank, dev, 5000 + 1000 * len(leg_out))
        ctx.prepare(lkind, lbits, lb)  # untimed: circuit built and levelised, wire store and the widest level's scratch allocated
        sc0 = stream_counters(ctx)
        lst = ia.Stats()
        # the leg BASELINE.json quotes its metric on is timed more than once (each pass bracketed on its own), so the line carries
        # a spread; a further pass is dropped, by all ranks together, when the time box has no room for it
        want = max(1, args.metric_passes) if key == "mul32" else 1
        pass_s, l_per_rank = [], None
        for pi in range(want):
            if pi and elapsed_all_ranks(torch, dist, dev, args.backend) + pass_s[-1] > args.time_box:
                break
            t_pass, l_per_rank = timed(torch, dist, world, dev, args.backend,
                                       lambda: ctx.eval_batch_device(lkind, lbits, lb, ld_in.data_ptr(), ld_out.data_ptr(), lst))
            pass_s.append(t_pass)
            check_outputs(tools, p, lwe_key, lkind, lbits, linb, ld_out, rank)  # every expression, after each timed pass
            progress(rank, "%s x%d pass %d: %.0f gate ops/s, %.2f expressions/s"
                     % (key, lb, pi + 1, int(linfo.bootstraps) * lb * world / t_pass, lb * world / t_pass))
        l_elapsed = sum(pass_s) / len(pass_s)                       # mean pass
        l_rate = int(linfo.bootstraps) * lb * world / l_elapsed
        per_pass = [int(linfo.bootstraps) * lb * world / t for t in pass_s]
        rec = {"workload": lname, "circuit": "%s%d" % (wl.rstrip("0123456789"), lbits), "batch_per_gpu": lb,
               "bootstraps_per_expr": int(linfo.bootstraps), "levels": int(linfo.depth), "passes": len(pass_s), "ms_per_pass": l_elapsed * 1e3,
               "gate_ops_per_s": l_rate, "expressions_per_s": lb * world / l_elapsed,
               "per_pass_gate_ops_per_s": per_pass, "spread": (max(per_pass) - min(per_pass)) / l_rate,
               "per_rank_gate_ops_per_s": [int(linfo.bootstraps) * lb / t for t in l_per_rank],
               "checked": "all %d expressions decrypt to the integer result after each timed pass" % lb,
               "warmup": "no warm-up pass (one is %.0f s); an untimed prepare call built the circuit and allocated the wire store and the "
                         "widest level's scratch beforehand, so the timed pass makes no allocation; every kernel it launches has run in "
                         "the primary leg" % l_elapsed,
               # two streams: a launch shares the chip with the other stream's launch of the same level; the evaluator picks kernels
               # by the gate instances in flight on both, and blind_rotate_ms is the time with a rotation in flight on either
               "streams": 2 if stream_counters(ctx) != sc0 else 1,
               "roofline": roofline(p, lst, l_rate / world, pmc,
                                    ctx.kernel_for_launch(round((2 if stream_counters(ctx) != sc0 else 1) * lst.bootstraps / max(1, lst.chunks))))}
        if lb != full_batch:
            rec["sub_batch_of"] = full_batch
            rec["full_share_estimate_s"] = l_elapsed * full_batch / lb
            rec["note"] = ("time-boxed sub-batch: %d of the config's %d expressions per GPU; the circuit's levels are whole rounds of resident gates "
                           "from batch 128 on, so the full share takes %d/%d x this pass" % (lb, full_batch, full_batch, lb))
        if key == "mul32":
            rec["mul32_per_s"] = lb * world / l_elapsed
            if args.extras:
                rec.update(mul32_extras(ia, tools, torch, dist, world, dev, args, ctx, p, lwe_key, lb, linb, ld_in, ld_out, rank, l_elapsed))
        leg_out[key] = rec
        del ld_in, ld_out

    if rank == 0:
        gates_total = info.bootstraps * batch * args.steps * world
        value = gates_total / elapsed
        out = {
            "metric": "bootstrapped gate ops/sec; encrypted 32-bit MUL/sec",
            "value": value,
            "unit": "gate ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": config_name, "circuit": "%s%d" % (args.workload.rstrip("0123456789"), bits),
                       "batch_per_gpu": batch, "bootstraps_per_expr": int(info.bootstraps), "levels": int(info.depth),
                       "params": "n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2",
                       "arithmetic": "Torus32 = int32 with wraparound; the negacyclic products inside the external product run as one f64 transform "
                                     "of the 32-bit coefficients with a rounding guard and a sampled two-limb audit (bit-identical to the two-limb "
                                     "exact transform, which exact_fft=1 selects; fft_guard below)",
                       "parallelism": "batch-sharded x%d" % world, "kernel": ctx.kernel_variant,
                       "key_broadcast_s": round(t_bcast, 4),
                       "rccl_ranks": world if (dist is not None and args.backend == "nccl") else 0,
                       "collective_backend": args.backend if dist is not None else None,
                       "per_rank_gate_ops_per_s": [info.bootstraps * batch * args.steps / t for t in per_rank]},
            "roofline": roofline(p, kstats, k_rate, pmc, ctx.kernel_for_launch(round(kstats.bootstraps / max(1, kstats.chunks)))),
            "fft_guard": fft_guard_record(ctx),
        }
        # what the roofline object was measured on, and the timed region priced the same way (launches of both streams merged on
        # the device timeline: EvalStats::blind_rotate_ms is the time with a blind rotation in flight)
        alg = algorithmic_flops_per_gate(p)
        out["roofline"]["mode"] = ("one stream (overlap = 0): %d steps of the same batch after the timed region" % args.roofline_steps) if kstats is not stats else "timed region"
        out["roofline"]["one_stream_gate_ops_per_s"] = k_rate * world
        out["roofline"]["timed_region"] = {
            "streams": 2 if overlap_on else 1,
            "frac_end_to_end": value / world * alg * 1e-12 / FP64_VALU_PEAK_TFLOPS,
            "frac_blind_rotation_in_flight": stats.bootstraps / max(1e-9, stats.blind_rotate_ms * 1e-3) * alg * 1e-12 / FP64_VALU_PEAK_TFLOPS,
            # (ieache_stats describes the LAST call it was passed to: one step)
            "blind_rotate_ms_last_step": stats.blind_rotate_ms, "launches_last_step": stats.blind_rotate_launches}
        out["config"]["overlap_streams"] = 2 if overlap_on else 1
        out.update(leg_out)
        if exact_out:
            out["exact"] = exact_out
        if "mul32" in leg_out:
            # BASELINE.json quotes its metric and target on batched 32-bit MUL: that leg, by name, at the top level.  `value` stays
            # on the K timed steps of configs[1] because one pass of this leg is ~57 s and the driver asks for 20 steps
            m = leg_out["mul32"]
            out["metric_leg"] = {"workload": m["workload"], "batch_per_gpu": m["batch_per_gpu"], "gate_ops_per_s": m["gate_ops_per_s"],
                                 "mul32_per_s": m["mul32_per_s"], "passes": m["passes"], "per_pass_gate_ops_per_s": m["per_pass_gate_ops_per_s"],
                                 "spread": m["spread"], "ms_per_pass": m["ms_per_pass"],
                                 "roofline_frac": m["roofline"]["frac"], "details": "mul32"}
            # `workload` names both configs the line reports on; the two keys after it are the machine-readable halves
            out["config"]["workload"] = "%s + %s" % (config_name, m["workload"])
            out["config"]["value_workload"] = config_name
            out["config"]["metric_leg_workload"] = m["workload"]
        if skipped:
            out["skipped_legs"] = skipped
        if "mul32" in leg_out:
            out["mul32_per_s"] = leg_out["mul32"]["mul32_per_s"]
        elif kind == 4 and bits == 32:
            out["mul32_per_s"] = batch * args.steps * world / elapsed
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, keys, args.cpu_seconds)
        print(compact_line(out, write_details(out, args.details)), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def ctx_kernel_exact(ctx, gates):
    """Context.kernel_for_launch under exact_fft = 1"""
    ctx.set_option("exact_fft", 1)
    try:
        return ctx.kernel_for_launch(gates)
    finally:
        ctx.set_option("exact_fft", 0)


def fft_guard_record(ctx):
    dev_max, reruns = ctx.fft_guard()
    rec = {"max_rounding_deviation": dev_max, "reruns_on_two_limb_kernel": reruns, "limit": 1.0 / 16, "wrong_bit_at": 0.5}
    audit = getattr(ctx, "fft_audit", None)
    if audit is not None:
        rec["audit"] = audit()
    return rec


def mul32_extras(ia, tools, torch, dist, world, dev, args, ctx, p, lwe_key, mb, minb, md_in, md_out, rank, m_elapsed):
    """--extras: the two opt-in 32-bit multipliers (decrypt-identical, NOT the reference's ciphertext bits; never the default)."""
    finfo = ia.circuit_info(4, 32, fold=True)
    ctx.set_option("fold_constants", 1)
    fst = ia.Stats()
    f_elapsed, _ = timed(torch, dist, world, dev, args.backend,
                         lambda: ctx.eval_batch_device(4, 32, mb, md_in.data_ptr(), md_out.data_ptr(), fst))
    ctx.set_option("fold_constants", 0)
    check_outputs(tools, p, lwe_key, 4, 32, minb, md_out, rank)
    progress(rank, "mul32 x%d, constants folded: %.2f MUL/s" % (mb, mb * world / f_elapsed))
    folded = {"flag": "fold_constants=1 (IEACHE_FOLD=1): constant operands folded, repeated gates shared; decrypt-identical, "
                      "NOT the reference's ciphertext bits; never the default",
              "executed_bootstraps_per_expr": int(finfo.bootstraps), "reference_bootstraps_per_expr": int(finfo.reference_bootstraps),
              "levels": int(finfo.depth), "ms_per_pass": f_elapsed * 1e3, "mul32_per_s": mb * world / f_elapsed,
              "executed_gate_ops_per_s": int(finfo.bootstraps) * mb * world / f_elapsed,
              "reference_equivalent_gate_ops_per_s": int(finfo.reference_bootstraps) * mb * world / f_elapsed,
              "speedup_vs_reference_circuit": m_elapsed / f_elapsed, "checked": "all %d products decrypt to a*b" % mb}
    winfo = ia.circuit_info(ia.CIRC_MUL_WALLACE, 32)
    wst = ia.Stats()
    w_elapsed, _ = timed(torch, dist, world, dev, args.backend,
                         lambda: ctx.eval_batch_device(ia.CIRC_MUL_WALLACE, 32, mb, md_in.data_ptr(), md_out.data_ptr(), wst))
    check_outputs(tools, p, lwe_key, 4, 32, minb, md_out, rank)
    progress(rank, "mul32 x%d, carry-save multiplier: %.2f MUL/s" % (mb, mb * world / w_elapsed))
    carry_save = {"flag": "IEACHE_CIRC_MUL_WALLACE (IEACHE_MULTIPLIER=wallace): Dadda carry-save tree + Kogge-Stone add; "
                          "decrypt-identical, NOT the reference's gate sequence; never the default",
                  "executed_bootstraps_per_expr": int(winfo.bootstraps), "reference_bootstraps_per_expr": int(winfo.reference_bootstraps),
                  "levels": int(winfo.depth), "ms_per_pass": w_elapsed * 1e3, "mul32_per_s": mb * world / w_elapsed,
                  "executed_gate_ops_per_s": int(winfo.bootstraps) * mb * world / w_elapsed,
                  "speedup_vs_reference_circuit": m_elapsed / w_elapsed, "checked": "all %d products decrypt to a*b" % mb}
    return {"folded": folded, "carry_save": carry_save}


if __name__ == "__main__":
    main()
