"""What overlapping the launches of two independent halves of a level would be worth (probe, not a product path): two contexts
(each its own stream and its own copy of the key) evaluate `count` gates each, one after the other against both at once from
two host threads.  A level's launches end with a tail (the last workgroups of every 16-step slice) and start with a ramp;
when another stream has work queued the freed CUs pick it up.  usage: overlap_probe.py [count ...]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ieache_amd as ia
from ieache_amd import tools
counts = [int(c) for c in sys.argv[1:]] or [2048, 4096, 8192]
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctxs = [ia.Context.from_arrays(p, k["bk"], k["ksk"]) for _ in range(2)]
rng = np.random.default_rng(5)
cmax = max(counts)
bits = rng.integers(0, 2, size=(2, cmax)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
stride = ctxs[0].lwe_stride
def dev(x):
    t = torch.zeros((x.shape[0], stride), dtype=torch.int32, device="cuda")
    t[:, : x.shape[1]] = torch.from_numpy(x).cuda()
    return t
da, db = dev(a), dev(b)
outs = [torch.zeros_like(da) for _ in range(2)]
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "5"))
def run(i, c):
    ctxs[i].gates_device(ia.GATE_XOR, c, da.data_ptr(), db.data_ptr(), outs[i].data_ptr())
for c in counts:
    for i in range(2):
        run(i, c)  # warm
    seq, par, one = [], [], []
    for r in range(reps):
        t0 = time.perf_counter(); run(0, c); t1 = time.perf_counter(); run(1, c); t2 = time.perf_counter()
        one.append(t1 - t0); seq.append(t2 - t0)
        th = [threading.Thread(target=run, args=(i, c)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        par.append(time.perf_counter() - t0)
    same = bool(torch.equal(outs[0][:c], outs[1][:c]))
    print("2 x %d gates: one call %.2f ms; one after the other %.2f ms; both at once %.2f ms (%.1f %% of sequential); outputs equal %s"
          % (c, min(one) * 1e3, min(seq) * 1e3, min(par) * 1e3, 100.0 * min(par) / min(seq), same), flush=True)
