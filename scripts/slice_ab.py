"""Slice length (CMux steps per blind-rotation launch) A/B for the mid-size kernels, interleaved in ONE process so that
box-to-box and run-to-run drift cancels (development aid).  usage: slice_ab.py variant:count[,count...] ...   (SLICES=16,64,630 ROUNDS=7)"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
specs = [(int(s.split(":")[0]), [int(c) for c in s.split(":")[1].split(",")]) for s in sys.argv[1:]]
slices = [int(x) for x in os.environ.get("SLICES", "16,64,630").split(",")]
rounds = int(os.environ.get("ROUNDS", "7"))
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(5)
count = max(max(c) for _, c in specs)
bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
ctx.set_option("exact_fft", 1)
ref = ctx.gates(ia.GATE_XOR, a, b)
ctx.set_option("exact_fft", 0)
for variant, counts in specs:
    ctx.set_option("br_variant", variant)
    for c in counts:
        t = {s: [] for s in slices}
        same = True
        for r in range(rounds):
            for s in slices:
                ctx.set_option("br_slice", s)
                st = ia.Stats()
                out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
                same = same and np.array_equal(ref[:c], out)
                t[s].append(st.blind_rotate_ms)
        print("variant", variant, "count", c, "identical", same, " ".join(
            "slice %d: min %.3f median %.3f ms" % (s, min(t[s]), statistics.median(t[s])) for s in slices), flush=True)
