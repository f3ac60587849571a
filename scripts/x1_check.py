"""k_blind_rotate_x1 (two limbs, one wave per gate) against k_blind_rotate_w2 (two limbs, two waves per gate) at full size:
bit identity and time per launch size and slice length (development aid).  usage: x1_check.py [count ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
counts = [int(c) for c in sys.argv[1:]] or [1025, 2048, 4099, 8192, 16384]
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(5)
count = max(counts)
bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
ctx.set_option("exact_fft", 1)
ctx.set_option("exact_one_wave_min", 1 << 40)
ref = ctx.gates(ia.GATE_XOR, a, b)
assert np.array_equal(tools.decrypt_bits(p, k["lwe_key"], ref), bits[0] ^ bits[1])
reps = int(os.environ.get("REPS", "3"))
for c in counts:
    for name, opts in (("w2", {"br_variant": 0, "exact_one_wave_min": 1 << 40, "br_slice": 0}),
                       ("x1 (9)", {"br_variant": 9, "br_slice": 16}),
                       ("x1 (9) again", {"br_variant": 9, "br_slice": 16})):
        for o, v in opts.items():
            ctx.set_option(o, v)
        best = None
        for rep in range(reps):
            st = ia.Stats()
            out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
            best = st.blind_rotate_ms if best is None else min(best, st.blind_rotate_ms)
        print("%-16s" % name, ctx.kernel_for_launch(c).split("<")[0], "count", c, "identical", np.array_equal(ref[:c], out),
              "BR ms %.3f" % best, "(%.0f gates/s)" % (c / best * 1e3), flush=True)
ctx.set_option("exact_fft", 0)
ctx.set_option("br_slice", 0)
ctx.set_option("br_variant", 0)
for c in counts:
    best = None
    for rep in range(reps):
        st = ia.Stats()
        out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
        best = st.blind_rotate_ms if best is None else min(best, st.blind_rotate_ms)
    print("%-16s" % "one limb", ctx.kernel_for_launch(c).split("<")[0], "count", c, "identical", np.array_equal(ref[:c], out),
          "BR ms %.3f" % best, "(%.0f gates/s)" % (c / best * 1e3), "guard", ctx.fft_guard(), flush=True)
