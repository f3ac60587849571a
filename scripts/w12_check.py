"""k_blind_rotate_wide12 (variants 40 / 41: 4L waves per gate, half-size transforms) against k_blind_rotate_wide4 (38) and the
two-limb reference at full size: bit identity and time per launch size (development aid).  usage: w12_check.py [count ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
counts = [int(c) for c in sys.argv[1:]] or [1, 7, 64, 128, 256]
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(5)
count = max(counts)
bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
ctx.set_option("exact_fft", 1)
ref = ctx.gates(ia.GATE_XOR, a, b)
assert np.array_equal(tools.decrypt_bits(p, k["lwe_key"], ref), bits[0] ^ bits[1])
ctx.set_option("exact_fft", 0)
reps = int(os.environ.get("REPS", "5"))
for c in counts:
    for name, opts in (("wide4 (38)", {"br_variant": 38}), ("wide4 spread (45)", {"br_variant": 45}), ("wide12 (40)", {"br_variant": 40}),
                       ("wide12 no prefetch (46)", {"br_variant": 46}), ("wide12 guard all (41)", {"br_variant": 41}),
                       ("wide4 (38) again", {"br_variant": 38})):
        for o, v in opts.items():
            ctx.set_option(o, v)
        ctx.set_option("br_slice", 4096)
        best = None
        for rep in range(reps):
            st = ia.Stats()
            out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
            best = st.blind_rotate_ms if best is None else min(best, st.blind_rotate_ms)
        same = np.array_equal(ref[:c], out)
        extra = ""
        if not same:
            bad = np.nonzero((ref[:c] != out).any(axis=1))[0]
            extra = " rows differing %d/%d, decrypts %s, max |diff| %d" % (
                len(bad), c, np.array_equal(tools.decrypt_bits(p, k["lwe_key"], out), bits[0][:c] ^ bits[1][:c]),
                int(np.abs((ref[:c].astype(np.int64) - out.astype(np.int64) + 2**31) % 2**32 - 2**31).max()))
        print("%-22s" % name, "count", c, "identical", same, "BR ms %.3f" % best, "guard", ctx.fft_guard(), extra, flush=True)
if os.environ.get("DIAG"):
    ctx.set_option("br_variant", 42)   # phase stamps on stderr
    ctx.set_option("br_slice", 4096)
    ctx.gates(ia.GATE_XOR, a[:64], b[:64])
    ctx.gates(ia.GATE_XOR, a[:64], b[:64])
