"""One circuit over a batch in a given stream mode, a few passes (for rocprofv3 --kernel-trace).  usage: pipe_probe.py workload batch one|halves|pipes|default passes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ieache_amd as ia
from ieache_amd import tools
import bench as B
wl, batch, mode, passes = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
p = ia.default_params()
k = tools.keygen_raw(p, (314, 1592, 657))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
ctx.set_option("fft_audit", int(os.environ.get("AUDIT", "0")))  # AUDIT=K: every K-th one-limb launch audited (the product's default is 64)
dev = torch.device("cuda", 0)
kind, bits, _, name = B.WORKLOADS[wl]
info, inb, d_in, d_out = B.make_inputs(ia, tools, torch, ctx, p, k["lwe_key"], kind, bits, batch, 0, dev, 1000)
if mode != "default":  # default: the product's own choice of stream mode
    ctx.set_option("overlap", 0 if mode == "one" else 1)
    ctx.set_option("pipe_min", (1 << 50) if mode in ("one", "halves") else 0)
ctx.eval_batch_device(kind, bits, batch, d_in.data_ptr(), d_out.data_ptr())
torch.cuda.synchronize()
ts = []
for _ in range(passes):
    t0 = time.perf_counter()
    ctx.eval_batch_device(kind, bits, batch, d_in.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = min(ts[-2:])  # the last passes: past the trial evaluations of "pipe_auto"
print("%s x %d mode %s: %.0f gate ops/s (%.1f ms per pass, best of the last two of %d; pipelined evaluations %d, trials %d)"
      % (wl, batch, mode, int(info.bootstraps) * batch / dt, dt * 1e3, passes, ctx.get_option("pipelined_evals"), ctx.get_option("tuned_evals")))
