"""Same-box A/B of the evaluator's stream modes: whole passes of a circuit over a resident batch, the modes interleaved, wall time per
pass bracketed by device synchronisation; outputs compared word for word between the modes.
  one      : every launch on one stream (overlap = 0)
  halves   : each level cut in two, the halves on two streams, join before the next level (pipe_min = never)
  pipes    : the batch cut into two halves of expressions, each through all levels on its own stream (the default for wide batches)
usage: overlap_ab.py [workload:batch:passes ...]   default add16:4096:5 mul32:1024:1     MODES=one,halves,pipes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ieache_amd as ia
from ieache_amd import tools
import bench as B

specs = [s.split(":") for s in (sys.argv[1:] or ["add16:4096:5", "mul32:1024:1"])]
modes = os.environ.get("MODES", "one,halves,pipes").split(",")
p = ia.default_params()
k = tools.keygen_raw(p, (314, 1592, 657))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
dev = torch.device("cuda", 0)
exact = int(os.environ.get("EXACT_FFT", "0"))
ctx.set_option("exact_fft", exact)
NEVER = 1 << 50


def set_mode(m):   # pipes / pipes3 / pipes4: two, three or four pipelines
    ctx.set_option("overlap", 0 if m == "one" else 1)
    ctx.set_option("pipe_min", 0 if m.startswith("pipes") else NEVER)
    ctx.set_option("pipe_lanes", int(m[5:] or 2) if m.startswith("pipes") else 2)


for wl, batch, passes in specs:
    batch, passes = int(batch), int(passes)
    kind, bits, _, name = B.WORKLOADS[wl]
    info, inb, d_in, d_out = B.make_inputs(ia, tools, torch, ctx, p, k["lwe_key"], kind, bits, batch, 0, dev, 1000)
    outs, times = {}, {m: [] for m in modes}
    ctx.prepare(kind, bits, batch)
    for m in modes:  # untimed: scratch of every mode in place
        set_mode(m)
        ctx.eval_batch_device(kind, bits, batch, d_in.data_ptr(), d_out.data_ptr())
    for r in range(passes):
        for m in modes:
            set_mode(m)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.eval_batch_device(kind, bits, batch, d_in.data_ptr(), d_out.data_ptr())
            torch.cuda.synchronize()
            times[m].append(time.perf_counter() - t0)
            print("[%s x %d pass %d mode %s: %.2f s]" % (wl, batch, r, m, times[m][-1]), file=sys.stderr, flush=True)  # a long run must not look hung
            if r == 0:
                outs[m] = d_out.clone()
    same = all(bool(torch.equal(outs[modes[0]], outs[m])) for m in modes)
    B.check_outputs(tools, p, k["lwe_key"], kind, bits, inb, outs[modes[-1]], 0)
    g = int(info.bootstraps) * batch
    base = g / min(times[modes[0]])
    txt = "; ".join("%s %.0f (mean %.0f, %+.2f %%)" % (m, g / min(times[m]), g / (sum(times[m]) / passes), 100 * (g / min(times[m]) / base - 1)) for m in modes)
    print("%s x %d%s, gate ops/s best of %d interleaved passes: %s; outputs identical %s, every expression decrypts"
          % (wl, batch, " exact_fft" if exact else "", passes, txt, same), flush=True)
    del d_in, d_out, outs
