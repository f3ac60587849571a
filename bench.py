#!/usr/bin/env python3
"""Benchmark of the hot path: bootstrapped gate ops/s on level-batched circuits.

  python bench.py --gpus N --steps K --warmup W [--workload add16|mul32|...] [--batch B]

One "step" = one pass of a whole circuit (every level, every gate) over one
batch of expressions whose ciphertexts are already resident in HBM.  Default
workload = BASELINE.json configs[1]: 16-bit ADD, batch 4096 per GPU.  For N>1
launch through torch.distributed.run (one rank per GPU); expressions shard
across ranks with no data-path collective (weak scaling: the per-GPU batch is
fixed), after a one-time RCCL broadcast of the bootstrapping / key-switch key.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# SURVEY.md section 8(d) / BASELINE.md section 3: algorithmic bytes per bootstrapped gate
# = BK n(k+1)l(k+1)N*4 + expected KSK rows N*t*(1-2^-basebit)*(n+1)*4 + LWE I/O 3(n+1)*4
HBM_PEAK_GBS = 8000.0
FP64_VALU_PEAK_TFLOPS = 78.6                 # MI355X_MICROARCH.md: 256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP64_OPS_PER_GATE = 630 * 3900 * 64          # lane-instructions of the blind rotation per gate at n = 630

WORKLOADS = {
    # name: (circuit kind, bits, default per-GPU batch, BASELINE.json config)
    "add16": (1, 16, 4096, "16-bit ADD, batch 4096 ciphertext pairs, 1xMI355X (level-batched bootstraps)"),
    "add32": (1, 32, 1024, "32-bit ADD"),
    "mul32": (4, 32, 1024, "32-bit shift-add MUL, batch 1024, 1xMI355X"),
    "muladd64": (5, 64, 128, "64-bit 3-operand a*b+c, batch 1024, sharded across 8xMI355X"),
    "mul128": (4, 128, 1024, "128-bit multi-precision MUL, batch 8192, 8xMI355X"),
}


def algorithmic_bytes_per_gate(p):
    bk = p.n * (p.k + 1) * p.l * (p.k + 1) * p.N * 4
    ksk = p.k * p.N * p.ks_t * (1.0 - 2.0 ** -p.ks_basebit) * (p.n + 1) * 4
    io = 3 * (p.n + 1) * 4
    return bk + ksk + io


def cpu_baseline(p, keys, seconds=12.0):
    """Times the CPU oracle's gate bootstrap on this host (rank 0, N=1 only).

    kind "port": the reference binary cannot be built (libtfhe absent).  The
    number quoted is the oracle's FP64-FFT back-end, which follows libtfhe's own
    algorithm (the exact-integer back-end used for parity is ~10x slower and is
    reported beside it)."""
    from oracle import oracle as O
    from ieache_amd import tools
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, keys["bk"], keys["ksk"])
    a = tools.encrypt_bits(p, keys["lwe_key"], np.array([1, 0], dtype=np.uint8), 1)
    b = tools.encrypt_bits(p, keys["lwe_key"], np.array([1, 1], dtype=np.uint8), 2)
    res = {}
    for mode, name, budget in ((O.POLYMUL_FFT, "fft", seconds * 0.7), (O.POLYMUL_NTT, "exact", seconds * 0.3)):
        ck.set_polymul(mode)
        ck.gate("and", a[0], b[0])  # warm-up (also builds the FFT-domain key once)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            ck.gate("xor" if n & 1 else "and", a[n & 1], b[n & 1])
            n += 1
        res[name] = (n, time.perf_counter() - t0)
    n, dt = res["fft"]
    ne, dte = res["exact"]
    return {
        "value": n / dt, "unit": "bootstrapped gate ops/s", "cores": 1, "kind": "port",
        "sample": "%d AND/XOR gates (n=%d,N=%d) in %.1f s with the oracle's FP64-FFT back-end (libtfhe's algorithm), "
                  "1 thread as the reference runs (its OpenMP pragmas are inert); exact-integer back-end: %.2f gates/s"
                  % (n, p.n, p.N, dt, ne / dte),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="add16", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="expressions per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N>1 flow with CPU collectives (ranks may then share one GPU)")
    args = ap.parse_args()

    import torch
    import ieache_amd as ia
    from ieache_amd import tools

    from ieache_amd import parallel
    rank, world, local_rank, dist = parallel.init_distributed(args.backend)
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())  # rehearsal: ranks wrap around the GPUs present
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kind, bits, def_batch, config_name = WORKLOADS[args.workload]
    batch = args.batch or def_batch
    p = ia.default_params()  # n=630 N=1024 l=3 Bgbit=7 t=8 basebit=2
    info = ia.circuit_info(kind, bits)

    # ---- keys: generated once on rank 0 (keygen.c seeds), broadcast over RCCL/xGMI ----
    keys = tools.keygen_raw(p, (314, 1592, 657)) if rank == 0 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bdev = dev if args.backend == "nccl" else torch.device("cpu")
    d_bk, d_ksk, d_key = (t.to(dev) for t in parallel.broadcast_cloud_key(p, keys, bdev, dist))
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0 if world > 1 else 0.0
    lwe_key = d_key.cpu().numpy()
    ctx = ia.Context.from_device_pointers(p, d_bk.data_ptr(), d_ksk.data_ptr(), device=local_rank)
    del d_bk, d_ksk
    if args.chunk:
        ctx.set_chunk(args.chunk)

    # ---- synthetic inputs: fresh encryptions of uniform random operands (seeded per rank) ----
    rng = np.random.default_rng(1000 + rank)
    inb = rng.integers(0, 2, size=(batch, info.n_inputs), dtype=np.uint8)
    inb[:, 2 * bits:2 * bits + 32] = 0  # the carry word is always 0 (alice.c:147-149)
    if batch >= 4:  # edge operands in slots 0-3: 0, 1, 2^w-1, process.c's 2^(w-2)
        for slot, v in enumerate((0, 1, (1 << bits) - 1, 1 << (bits - 2))):
            inb[slot, :bits] = tools.int_to_bits(v, bits)
            inb[slot, bits:2 * bits] = tools.int_to_bits(v, bits)
    stride = ctx.lwe_stride
    d_in = torch.zeros((batch, info.n_inputs, stride), dtype=torch.int32, device=dev)
    rows_per = max(1, (1 << 26) // (info.n_inputs * (p.n + 1) * 4))
    for s in range(0, batch, rows_per):  # stream the encryption through host memory in <=64 MiB pieces
        e = min(batch, s + rows_per)
        ct = tools.encrypt_bits(p, lwe_key, inb[s:e], 7777 + 131 * rank + s)
        d_in[s:e, :, :p.n + 1] = torch.from_numpy(ct).to(dev)
        del ct
    d_out = torch.zeros((batch, info.n_outputs, stride), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step(stats=None):
        ctx.eval_batch_device(kind, bits, batch, d_in.data_ptr(), d_out.data_ptr(), stats)

    for _ in range(args.warmup):
        step()
    # correctness of what is being timed: decrypt a few expressions
    if args.warmup == 0:
        step()
    # every expression of the batch must decrypt to the integer result (P1 at full size)
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

    stats = ia.Stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(stats)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        gates_total = info.bootstraps * batch * args.steps * world
        value = gates_total / elapsed
        per_gate = algorithmic_bytes_per_gate(p)
        # dominant kernel: blind rotation (k_blind_rotate_*).  One launch advances `gates_per_launch`
        # gates by `steps_per_launch` of their n CMux steps, i.e. processes that fraction of each gate's
        # algorithmic bytes; achieved = algorithmic bytes per launch / average launch duration (HIP
        # events on the evaluator's stream bracket the launches of each chunk).
        br_avg_ms = stats.blind_rotate_ms / max(1, stats.blind_rotate_launches)
        gates_per_launch = stats.bootstraps / max(1, stats.chunks)
        launches_per_gate = stats.blind_rotate_launches / max(1, stats.chunks)
        bytes_per_launch = per_gate * gates_per_launch / max(1.0, launches_per_gate)
        achieved = bytes_per_launch / (br_avg_ms * 1e-3) / 1e9 if br_avg_ms > 0 else 0.0
        # HBM traffic of the dominant kernel from PMC counters: collected offline (rocprofv3 cannot run
        # inside this process) by scripts/pmc_passes.sh and committed in profiles/traffic.json
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(ctx.kernel_variant)
            if tj:
                traffic = tj["hbm_bytes_per_gate_step"] * gates_per_launch * (p.n / max(1.0, launches_per_gate))
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "bootstrapped gate ops/sec",
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
                       "arithmetic": "Torus32 = int32 with wraparound; the negacyclic products inside the external product run as an exact two-limb f64 transform", "parallelism": "batch-sharded x%d" % world,
                       "kernel": ctx.kernel_variant, "key_broadcast_s": round(t_bcast, 4)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_blind_rotate", "avg_launch_ms": br_avg_ms, "gates_per_launch": gates_per_launch,
                         "cmux_steps_per_launch": p.n / max(1.0, launches_per_gate),
                         "algorithmic_bytes_per_gate": per_gate, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "blind_rotate_share": stats.blind_rotate_ms / max(1e-9, stats.total_ms),
                         "keyswitch_share": stats.keyswitch_ms / max(1e-9, stats.total_ms),
                         # secondary line: BK is shared by all gates in flight, so the kernel's real bound is
                         # the FP64 vector pipe: ~3 900 FP64-rate instructions per lane and CMux step
                         "secondary": {"bound": "fp64_valu", "unit": "TFLOP/s-equivalent (1 instr = 2 flop)",
                                       "achieved": FP64_OPS_PER_GATE * 2e-12 * (achieved * 1e9 / per_gate),
                                       "peak": FP64_VALU_PEAK_TFLOPS,
                                       "frac": FP64_OPS_PER_GATE * 2e-12 * (achieved * 1e9 / per_gate) / FP64_VALU_PEAK_TFLOPS}},
        }
        if kind == 4 and bits == 32:
            out["mul32_per_s"] = batch * args.steps * world / elapsed
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, keys, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
