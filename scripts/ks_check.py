"""Sliced key switch vs the other key-switch kernels at full size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 + 37
u = rng.integers(-2**31, 2**31, size=(count, p.N + 1), dtype=np.int64).astype(np.int32)
ctx.set_option("ks_sliced_min", 1 << 40)
ctx.set_option("ks_batch_min", 1 << 40)
ref = ctx.debug_keyswitch(u)
ctx.set_option("ks_batch_min", 1)
assert np.array_equal(ctx.debug_keyswitch(u), ref), "batch kernel differs"
ctx.set_option("ks_sliced_min", 1)
for sl, g in ((0, 0), (7, 8), (64, 16), (1024, 32), (100, 8), (0, 16), (333, 32), (0, 4), (9, 4)):
    ctx.set_option("ks_slice", sl)
    ctx.set_option("ks_gates", g)
    out = ctx.debug_keyswitch(u)
    ok = np.array_equal(out, ref)
    print("sliced key switch, slice", sl, "gates/wg", g, "count", count, "bit-exact", ok, flush=True)
    if not ok:
        bad = np.argwhere(out != ref)
        print("first mismatches (row, col):", bad[:8].tolist(), "of", len(bad))
        sys.exit(1)
