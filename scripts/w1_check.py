"""One-limb kernels vs the two-limb ones: bit identity and the guard record, at full size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(5)
count = 8192
bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
ctx.set_option("exact_fft", 1)
ref = ctx.gates(ia.GATE_XOR, a, b)
ctx.set_option("exact_fft", 0)
for variant, counts in ((13, (4099, 8192)), (20, (600, 1024)), (24, (1, 37, 256)), (7, (1, 256))):
    ctx.set_option("br_variant", variant)
    for c in counts:
        best = None
        for rep in range(3):
            st = ia.Stats()
            out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
            best = st.blind_rotate_ms if best is None else min(best, st.blind_rotate_ms)
        print("variant", variant, "count", c, "identical", np.array_equal(ref[:c], out), "BR ms %.3f" % best, "guard", ctx.fft_guard(), flush=True)
