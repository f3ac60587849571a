"""Blind-rotation kernel variants against the two-limb (exact) kernel at full size: bit identity, time, guard record
(development aid).  usage: variant_check.py 13:4099,8192 31:4099,8192 ...   (variant:counts)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
specs = [(int(s.split(":")[0]), [int(c) for c in s.split(":")[1].split(",")]) for s in sys.argv[1:]]
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
for opt in ("br_slice",):
    if os.environ.get(opt.upper()):
        ctx.set_option(opt, int(os.environ[opt.upper()]))
for variant, counts in specs:
    ctx.set_option("br_variant", variant)
    for c in counts:
        best = None
        for rep in range(int(os.environ.get("REPS", "3"))):
            st = ia.Stats()
            out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
            best = st.blind_rotate_ms if best is None else min(best, st.blind_rotate_ms)
        print("variant", variant, "count", c, "identical", np.array_equal(ref[:c], out), "BR ms %.3f" % best,
              "(%.0f gates/s)" % (c / best * 1e3), "guard", ctx.fft_guard(), flush=True)
