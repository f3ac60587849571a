"""One-limb vs two-limb blind rotation: bit identity and the guard record, at full size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(5)
for count in (4099, 1030, 8192):
    bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
    a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
    b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
    ctx.set_option("exact_fft", 1)
    ref = ctx.gates(ia.GATE_XOR, a, b)
    ctx.set_option("exact_fft", 0)
    ctx.set_option("one_limb_min", 0)
    ctx.set_option("br_wide_max", 0)
    st = ia.Stats()
    out = ctx.gates(ia.GATE_XOR, a, b, st)
    print("count", count, "identical", np.array_equal(ref, out), "decrypt ok",
          np.array_equal(tools.decrypt_bits(p, k["lwe_key"], out), bits[0] ^ bits[1]), "BR ms %.2f" % st.blind_rotate_ms,
          "guard", ctx.fft_guard(), flush=True)
